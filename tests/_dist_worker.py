"""Worker of tests/test_dist_gloo.py: one rank of a world-size-N gloo job that rehearses, on CPU, the
row-sharded algorithm libsla_hip runs over RCCL: slab generation, padded-shard all-gather of the SpMV
input, rank-ordered sums of per-rank partials, and one BiCGSTAB step assembled from sharded pieces.
The oracle is used here as the per-rank checker arithmetic (this is a test, not the product)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sparse-linear-algebra_amd")):
    sys.path.insert(0, p)

from oracle import oracle as orc  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    # import the partition / workload modules without loading the HIP library
    import importlib.util
    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "sparse-linear-algebra_amd", "sla_amd", name + ".py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
    part, wl = load("partition"), load("workloads")

    dims, (rp, ci, va) = wl.laplace3d(8, 7, 9)
    n = dims[0]
    full = orc.Csr(n, n, rp, ci, va)
    b, e = part.row_block(n, rank, P)
    S = part.shard_size(n, P)
    _, (rpl, cil, val) = wl.laplace3d(8, 7, 9, b, e)                 # this rank builds ONLY its slab
    assert np.array_equal(rpl, rp[b:e + 1] - rp[b])
    loc = orc.Csr(e - b, n, rpl, cil, val)                           # global column ids

    def allgather(x_local):
        send = torch.from_numpy(part.pad_shard(x_local, S))
        recv = [torch.zeros(S, dtype=torch.float64) for _ in range(P)]
        dist.all_gather(recv, send)
        return torch.cat(recv).numpy()                               # global index g lives at g

    def gdot(u, v):                                                  # per-rank fold, rank-ordered sum
        mine = torch.tensor([orc.dot(u, v)], dtype=torch.float64)
        parts = [torch.zeros(1, dtype=torch.float64) for _ in range(P)]
        dist.all_gather(parts, mine)
        acc = 0.0
        for t in parts:
            acc += float(t.item())
        return acc

    def spmv(x_local):
        return orc.spmv(loc, allgather(x_local)[:n])

    # ---- window (halo) exchange: the library's own plan, executed over gloo send/recv -----------------
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "sparse-linear-algebra_amd", "lib", "libsla_hip.so"))   # loads without a GPU
    mine = torch.tensor([int(cil.min()), int(cil.max())], dtype=torch.int64)
    wins = [torch.zeros(2, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(wins, mine)
    windows = np.ascontiguousarray(torch.cat(wins).numpy())
    plans = []
    for q in range(P):
        o = [np.zeros(P, dtype=np.int64) for _ in range(4)]
        use = C.c_int()
        rc = L.sla_plan_window_exchange(C.c_int(P), C.c_int(q), C.c_int64(n), C.c_void_p(windows.ctypes.data),
                                        *[C.c_void_p(a.ctypes.data) for a in o], C.byref(use))
        assert rc == 0
        plans.append((o, use.value))
    for p_ in range(P):                                              # pairwise consistency: no hang by construction
        for q in range(P):
            if p_ != q:
                assert plans[p_][0][0][q] == plans[q][0][2][p_] and plans[p_][0][1][q] == plans[q][0][3][p_]
    assert all(u == plans[0][1] for _, u in plans)                   # every rank takes the same mode
    (sb, sl, rb_, rl), use_window = plans[rank]
    assert use_window == 1 or P == 1, "a slab-partitioned stencil must pick the window exchange"

    def window_exchange(x_local):
        xfull = np.full(S * P, np.nan)                               # untouched entries must never be read
        xfull[b:e] = x_local
        ops, bufs = [], {}
        for q in range(P):
            if q == rank:
                continue
            if rl[q] > 0:
                bufs[q] = torch.zeros(int(rl[q]), dtype=torch.float64)
                ops.append(dist.P2POp(dist.irecv, bufs[q], q))
            if sl[q] > 0:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(x_local[sb[q] - b: sb[q] - b + sl[q]].copy()), q))
        for w_ in (dist.batch_isend_irecv(ops) if ops else []):
            w_.wait()
        for q, t in bufs.items():
            xfull[rb_[q]: rb_[q] + rl[q]] = t.numpy()
        return xfull

    rng = np.random.default_rng(5)
    xg = rng.standard_normal(n)
    yw = orc.spmv(loc, window_exchange(xg[b:e])[:n])
    assert np.array_equal(yw, orc.spmv(full, xg)[b:e]), "window-exchange SpMV != whole-matrix SpMV"
    recv_total = int(rl.sum())
    assert recv_total <= 2 * 8 * 7, "a 7-pt slab needs at most one plane from each neighbour"
    y_full = orc.spmv(full, xg)
    y_loc = spmv(xg[b:e])
    assert np.array_equal(y_loc, y_full[b:e]), "sharded SpMV != whole-matrix SpMV"
    d = gdot(xg[b:e], y_loc)
    assert abs(d - orc.dot(xg, y_full)) <= 1e-12 * abs(d)

    # one bicgstabStep from sharded pieces vs the oracle's whole-matrix step
    bg = orc.spmv(full, np.ones(n))
    x0 = np.full(n, 0.1)
    so = orc.BicgstabState(full, bg, x0)
    r0hat_g = bg - orc.spmv(full, x0)
    so.step(r0hat_g, 1)
    x, r = x0[b:e].copy(), (bg[b:e] - spmv(x0[b:e]))
    p, r0hat = r.copy(), r.copy()
    rho = gdot(r, r0hat)
    ap = spmv(p)
    alpha = rho / gdot(ap, r0hat)
    s = r - alpha * ap
    as_ = spmv(s)
    omega = gdot(as_, s) / gdot(as_, as_)
    x = (x + alpha * p) + omega * s
    r = s - omega * as_
    beta = gdot(r, r0hat) / rho * alpha / omega
    p = r + beta * (p - omega * ap)
    assert np.allclose(x, so.x[b:e], rtol=1e-12, atol=1e-13)
    assert np.allclose(p, so.p[b:e], rtol=1e-10, atol=1e-12)
    # ---- ghost-row BiCGSTAB (enqueue_bicgstab_ghost): 3 exchanges per step instead of 5, same bits --------------------
    # r, p, Ap and s are kept valid on the ghost rows (the neighbours' rows this slab reads); halo(Ap) travels with the
    # alpha partials and halo(r') with the rho partials, each as ONE batch of point-to-point transfers -- per-rank sums
    # included, like the library's pure send/recv group -- and no exchange precedes either SpMV.
    ext = np.zeros(S * P, dtype=bool)
    ext[b:e] = True
    for q in range(P):
        if q != rank and rl[q] > 0:
            ext[rb_[q]: rb_[q] + rl[q]] = True

    def sums_and_halo(mine, vfull):
        """all-gather of this rank's partial sums + halo exchange of vfull (in place), as one p2p batch"""
        k = len(mine)
        got = {rank: np.array(mine, dtype=np.float64)}
        ops, sb_, hb = [], {}, {}
        for q in range(P):
            if q == rank:
                continue
            sb_[q] = torch.zeros(k, dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, sb_[q], q))
            ops.append(dist.P2POp(dist.isend, torch.tensor(mine, dtype=torch.float64), q))
        for q in range(P):
            if q == rank:
                continue
            if rl[q] > 0:
                hb[q] = torch.zeros(int(rl[q]), dtype=torch.float64)
                ops.append(dist.P2POp(dist.irecv, hb[q], q))
            if sl[q] > 0:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(vfull[sb[q]: sb[q] + sl[q]].copy()), q))
        for w_ in (dist.batch_isend_irecv(ops) if ops else []):
            w_.wait()
        for q, t in sb_.items():
            got[q] = t.numpy()
        for q, t in hb.items():
            vfull[rb_[q]: rb_[q] + rl[q]] = t.numpy()
        tot = np.zeros(k)
        for q in range(P):                                           # rank-ordered sum, like gdot
            tot = tot + got[q]
        return tot

    def plain_steps(k):
        x, r = x0[b:e].copy(), (bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n]))
        p, r0h = r.copy(), r.copy()
        rho = gdot(r, r0h)
        for _ in range(k):
            ap = orc.spmv(loc, window_exchange(p)[:n])               # exchange 1
            alpha = rho / gdot(ap, r0h)                              # exchange 2
            s = r - alpha * ap
            as_ = orc.spmv(loc, window_exchange(s)[:n])              # exchange 3
            omega = gdot(as_, s) / gdot(as_, as_)                    # exchange 4 (one all-gather of two sums in the library)
            x = (x + alpha * p) + omega * s
            r = s - omega * as_
            rho1 = gdot(r, r0h)                                      # exchange 5
            beta = rho1 / rho * alpha / omega
            p = r + beta * (p - omega * ap)
            rho = rho1
        return x, r, p

    def ghost_steps(k):
        nanv = lambda: np.full(S * P, np.nan)                        # noqa: E731  (rows outside own + ghost stay NaN: never read)
        x = x0[b:e].copy()
        R, Pv, AP, Sv = nanv(), nanv(), nanv(), nanv()
        R[b:e] = bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n])
        r0h = R[b:e].copy()
        R[:] = window_exchange(R[b:e])                               # invariant at step start: halo(r), halo(p) valid
        Pv[:] = R
        rho = gdot(R[b:e], r0h)
        for _ in range(k):
            AP[b:e] = orc.spmv(loc, Pv[:n])
            alpha = rho / sums_and_halo([orc.dot(AP[b:e], r0h)], AP)[0]          # exchange 1: alpha partials + halo(Ap)
            Sv[ext] = R[ext] - alpha * AP[ext]
            as_ = orc.spmv(loc, Sv[:n])
            omega = gdot(as_, Sv[b:e]) / gdot(as_, as_)                          # exchange 2
            x = (x + alpha * Pv[b:e]) + omega * Sv[b:e]
            R[b:e] = Sv[b:e] - omega * as_
            rho1 = sums_and_halo([orc.dot(R[b:e], r0h)], R)[0]                   # exchange 3: rho partials + halo(r')
            beta = rho1 / rho * alpha / omega
            Pv[ext] = R[ext] + beta * (Pv[ext] - omega * AP[ext])
            rho = rho1
        return x, R[b:e].copy(), Pv[b:e].copy()

    for a_, g_ in zip(plain_steps(3), ghost_steps(3)):
        assert np.array_equal(a_, g_), "ghost-row flow must reproduce the plain sharded flow bit for bit"

    # ---- the fused K4+K5 sweep on row-sharded contexts (round 3; the library's default, SLA_BICG_FUSE45) ----------------------
    # rho' = s . r0hat - omega (As . r0hat) by linearity (bicg_k45_kernel), so K3's four sums -- As . s, As . As, As . r0hat,
    # s . r0hat -- travel as ONE exchange and the rho exchange is gone: 4 exchanges per step on the plain flow; on the ghost-row
    # flow halo(As) rides with the four sums and the sweep runs on own + ghost rows: TWO exchanges per step.
    def fused_plain_steps(k):
        x, r = x0[b:e].copy(), (bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n]))
        p, r0h = r.copy(), r.copy()
        rho = gdot(r, r0h)
        for _ in range(k):
            ap = orc.spmv(loc, window_exchange(p)[:n])               # exchange 1
            alpha = rho / gdot(ap, r0h)                              # exchange 2
            s = r - alpha * ap
            as_ = orc.spmv(loc, window_exchange(s)[:n])              # exchange 3
            q = sums_only([orc.dot(as_, s), orc.dot(as_, as_), orc.dot(as_, r0h), orc.dot(s, r0h)])   # exchange 4: ONE all-gather of four sums
            omega = q[0] / q[1]
            rho1 = q[3] - omega * q[2]
            beta = rho1 / rho * alpha / omega
            x = (x + alpha * p) + omega * s
            r = s - omega * as_
            p = r + beta * (p - omega * ap)
            rho = rho1
        return x, r, p

    def sums_only(mine):
        parts = [torch.zeros(len(mine), dtype=torch.float64) for _ in range(P)]
        dist.all_gather(parts, torch.tensor(mine, dtype=torch.float64))
        tot = np.zeros(len(mine))
        for q in range(P):
            tot = tot + parts[q].numpy()
        return tot

    def fused_ghost_steps(k):
        nanv = lambda: np.full(S * P, np.nan)                        # noqa: E731
        X, R, Pv, AP, Sv, AS = nanv(), nanv(), nanv(), nanv(), nanv(), nanv()
        X[b:e] = x0[b:e]
        X[ext & np.isnan(X)] = 0.0                                   # (x rides along on the ghost rows: whatever is there)
        R[b:e] = bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n])
        r0h = R[b:e].copy()
        R[:] = window_exchange(R[b:e])
        Pv[:] = R
        rho = gdot(R[b:e], r0h)
        for _ in range(k):
            AP[b:e] = orc.spmv(loc, Pv[:n])
            alpha = rho / sums_and_halo([orc.dot(AP[b:e], r0h)], AP)[0]          # exchange 1: alpha partials + halo(Ap)
            Sv[ext] = R[ext] - alpha * AP[ext]
            AS[b:e] = orc.spmv(loc, Sv[:n])
            q = sums_and_halo([orc.dot(AS[b:e], Sv[b:e]), orc.dot(AS[b:e], AS[b:e]), orc.dot(AS[b:e], r0h), orc.dot(Sv[b:e], r0h)], AS)   # exchange 2: four sums + halo(As)
            omega = q[0] / q[1]
            rho1 = q[3] - omega * q[2]
            beta = rho1 / rho * alpha / omega
            X[ext] = (X[ext] + alpha * Pv[ext]) + omega * Sv[ext]
            R[ext] = Sv[ext] - omega * AS[ext]
            Pv[ext] = R[ext] + beta * (Pv[ext] - omega * AP[ext])
            rho = rho1
        return X[b:e].copy(), R[b:e].copy(), Pv[b:e].copy()

    fp, fg = fused_plain_steps(3), fused_ghost_steps(3)
    for a_, g_ in zip(fp, fg):
        assert np.array_equal(a_, g_), "the fused ghost-row flow must reproduce the fused plain sharded flow bit for bit"
    for a_, g_ in zip(plain_steps(3), fp):                           # rho' through the identity: a regrouping-level difference
        assert np.allclose(a_, g_, rtol=1e-9, atol=1e-11)

    # ---- ghost-row CGS (enqueue_cgs_ghost): 2 exchanges per step instead of 4 ------------------------------------------
    def cgs_plain(k):
        x, r = x0[b:e].copy(), (bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n]))
        p, u, rh = r.copy(), r.copy(), r.copy()
        rho = gdot(r, rh)
        for _ in range(k):
            aap = orc.spmv(loc, window_exchange(p)[:n])              # exchange 1
            alpha = rho / gdot(aap, rh)                              # exchange 2
            q = u - alpha * aap
            uq = u + q
            x = x + alpha * uq
            r = r - alpha * orc.spmv(loc, window_exchange(uq)[:n])   # exchange 3
            rho1 = gdot(r, rh)                                       # exchange 4
            beta = rho1 / rho
            u = r + beta * q
            p = u + beta * (q + beta * p)
            rho = rho1
        return x, r, p, u

    def cgs_ghost(k):
        nanv = lambda: np.full(S * P, np.nan)                        # noqa: E731
        x = x0[b:e].copy()
        R, Pv, U, AAP, Q, UQ = nanv(), nanv(), nanv(), nanv(), nanv(), nanv()
        R[b:e] = bg[b:e] - orc.spmv(loc, window_exchange(x0[b:e])[:n])
        rh = R[b:e].copy()
        Pv[:] = window_exchange(R[b:e])                              # invariant at step start: halo(p), halo(u) valid
        U[:] = Pv
        rho = gdot(R[b:e], rh)
        for _ in range(k):
            AAP[b:e] = orc.spmv(loc, Pv[:n])
            alpha = rho / sums_and_halo([orc.dot(AAP[b:e], rh)], AAP)[0]         # exchange 1: alpha partials + halo(aap)
            Q[ext] = U[ext] - alpha * AAP[ext]
            UQ[ext] = U[ext] + Q[ext]
            x = x + alpha * UQ[b:e]
            R[b:e] = R[b:e] - alpha * orc.spmv(loc, UQ[:n])
            rho1 = sums_and_halo([orc.dot(R[b:e], rh)], R)[0]                    # exchange 2: rho partials + halo(r')
            beta = rho1 / rho
            U[ext] = R[ext] + beta * Q[ext]
            Pv[ext] = U[ext] + beta * (Q[ext] + beta * Pv[ext])
            rho = rho1
        return x, R[b:e].copy(), Pv[b:e].copy(), U[b:e].copy()

    for a_, g_ in zip(cgs_plain(3), cgs_ghost(3)):
        assert np.array_equal(a_, g_), "ghost-row CGS must reproduce the plain sharded flow bit for bit"

    # sharded transpose SpMV (CGNE, <#): local transposed block -> full-length partial -> sum over ranks -> own shard
    wv = rng.standard_normal(n)
    tl = orc.transpose(loc)                                        # n rows, columns = local row ids
    partial = torch.from_numpy(orc.spmv(tl, wv[b:e]))
    dist.all_reduce(partial)                                       # (the library reduce-scatters; the shard is the same)
    ref_t = orc.spmv(orc.transpose(full), wv)
    assert np.allclose(partial.numpy()[b:e], ref_t[b:e], rtol=1e-13, atol=1e-13)
    # ---- overlapped all-gather of x for all-gather-mode tile matrices (round 4) ------------------------------
    # The library's own plan (sla_plan_allgather_groups / _passes, pure host), its message pattern over gloo -- ONE batch_isend_irecv
    # per exchange group, every source sending its pieces to every peer -- and the pass-ordered fold: a pass may only read columns
    # that are the rank's own or have arrived with the groups it waits for (everything else is NaN until then).
    rdims, (rrp, rci, rva) = wl.random_spd(6000, 6, 17)
    rn = rdims[0]
    rfull = orc.Csr(rn, rn, rrp, rci, rva)
    rb0, re0 = part.row_block(rn, rank, P)
    rloc = orc.Csr(re0 - rb0, rn, *part.local_rows_of(rrp, rci, rva, rb0, re0))
    xr = rng.standard_normal(rn)
    shift = 9
    npan = (rn + (1 << shift) - 1) >> shift
    for order, groups in ((0, 4), (0, 2), (1, 4)):
        cap = 64 * P
        buf = np.zeros(4 * cap, dtype=np.int64)
        cnt = C.c_int()
        assert L.sla_plan_allgather_groups(C.c_int(P), C.c_int64(rn), C.c_int(shift), C.c_int(groups), C.c_int(order), C.c_void_p(buf.ctypes.data),
                                           C.c_int(cap), C.byref(cnt)) == 0
        pieces = buf[:4 * cnt.value].reshape(-1, 4).tolist()
        visit, pptr, pneed = np.zeros(npan, np.int32), np.zeros(npan + 1, np.int32), np.zeros(npan, np.int32)
        npass, ng = C.c_int(), C.c_int()
        assert L.sla_plan_allgather_passes(C.c_int(P), C.c_int(rank), C.c_int64(rn), C.c_int(shift), C.c_int(groups), C.c_int(order),
                                           C.c_void_p(visit.ctypes.data), C.c_void_p(pptr.ctypes.data), C.c_void_p(pneed.ctypes.data),
                                           C.byref(npass), C.byref(ng)) == 0
        xfull = np.full(rn, np.nan)
        xfull[rb0:re0] = xr[rb0:re0]
        yrun = np.zeros(re0 - rb0)
        arrived = 0

        def exchange_group(g):
            ops, land = [], []
            for gg, src, pb, pe in pieces:                           # every rank walks the same list: matching order on both sides
                if gg != g or pe <= pb:
                    continue
                if src == rank:
                    for q in range(P):
                        if q != rank:
                            ops.append(dist.P2POp(dist.isend, torch.from_numpy(xr[pb:pe].copy()), q))
                else:
                    t = torch.zeros(pe - pb, dtype=torch.float64)
                    land.append((pb, pe, t))
                    ops.append(dist.P2POp(dist.irecv, t, src))
            for w_ in (dist.batch_isend_irecv(ops) if ops else []):
                w_.wait()
            for pb, pe, t in land:
                xfull[pb:pe] = t.numpy()

        for p_ in range(npass.value):
            while arrived < pneed[p_]:
                exchange_group(arrived)
                arrived += 1
            for j in visit[pptr[p_]:pptr[p_ + 1]]:                   # one panel of the pass: continue every row's running sum
                lo, hi = int(j) << shift, min(rn, (int(j) + 1) << shift)
                assert not np.isnan(xfull[lo:hi]).any(), "a pass read columns that have not arrived"
                for i in range(re0 - rb0):
                    acc = yrun[i]
                    for k in range(rloc.rowptr[i], rloc.rowptr[i + 1]):
                        if lo <= rloc.colidx[k] < hi:
                            acc = acc + rloc.val[k] * xfull[rloc.colidx[k]]
                    yrun[i] = acc
        while arrived < ng.value:                                    # (groups nobody on this rank waited for still have to be received)
            exchange_group(arrived)
            arrived += 1
        assert not np.isnan(xfull).any()
        assert np.array_equal(yrun, orc.spmv_panel_order(rloc, xr, shift, visit)), "pass-ordered fold != the oracle's restatement of it"
        yref = orc.spmv(rfull, xr)[rb0:re0]
        if order == 1:
            assert np.array_equal(yrun, yref), "ascending order must be the reference's left fold bit for bit"
        else:
            assert np.abs(yrun - yref).max() <= 64 * np.finfo(np.float64).eps * np.abs(rva).max() * np.abs(xr).max()
    dist.barrier()
    if rank == 0:
        print("DIST_OK", P)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
