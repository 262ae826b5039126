"""Seeded fuzzer of the fused Gram-Schmidt step (csrc/sla_arnoldi_orth.hip): random row counts around the block edges (one workgroup, a short last
block, odd counts, exactly full blocks), random numbers of Krylov columns (1 .. 31), banded / stencil / random matrices -- arnoldi with arn_orth = 1
against arn_orth = 0 (H 1e-10, Q 1e-9 -- the two group the inner products differently) and against the oracle where it finishes in seconds.
usage: python tools/fuzz_arnoldi.py [cases] [seed]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sparse-linear-algebra_amd"))
import numpy as np
import sla_amd as sla
from sla_amd import workloads as wl
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 616)
fused = compared = 0
for case in range(cases):
    kind = case % 3
    if kind == 0:
        n = int(rng.choice([1024, 1025, 2047, 2048, 8192, 8193, 16384, 65537, 262144, 262145 + 2 * int(rng.integers(0, 500)), 700001]))
        dims, (rp, ci, va) = wl.banded_nonsym(n, seed=int(rng.integers(1, 1000)))
    elif kind == 1:
        g = int(rng.integers(11, 64))
        dims, (rp, ci, va) = wl.laplace3d(g, g + int(rng.integers(0, 5)), g + int(rng.integers(0, 3)))
    else:
        dims, (rp, ci, va) = wl.random_spd(int(rng.integers(1100, 60000)), int(rng.integers(2, 6)), int(rng.integers(1, 1000)))
    n = dims[0]
    kn = int(rng.integers(1, 32))
    b = rng.standard_normal(n)
    res = []
    for orth in (1, 0):
        ctx = sla.Context(0).set_options(arn_orth=orth)
        A = sla.fromCSR(dims, rp, ci, va, ctx)
        Q, H = sla.arnoldi(A, sla.fromVector(b, ctx), kn)
        res.append((Q, H, int(ctx.get_option("arn_orth_launches"))))
        del A
        ctx.close()
    (Q1, H1, l1), (Q0, H0, l0) = res
    assert l0 == 0 and (l1 > 0) == (n >= 1024), (case, n, kn, l1, l0)
    fused += l1 > 0
    assert H1.shape == H0.shape and Q1.shape == Q0.shape, (case, n, kn, H1.shape, H0.shape)
    hs = max(np.abs(H0).max(), 1e-300)
    # Classical Gram-Schmidt loses orthogonality like kappa^2 eps: once the Krylov vectors are nearly dependent (random SPD matrices, many columns) ANY
    # regrouping of the inner products moves H and Q by far more than rounding -- the two flows are then compared through what must hold regardless,
    # the Arnoldi relation aa #> q_j = Q h_j column by column (checkArnoldi, LibSpec.hs:642-653), and entry by entry only while the launch flow's own
    # basis is orthonormal to 1e-10.
    Ao = orc.Csr(n, n, rp, ci, va)
    k = H1.shape[1]
    for Q, H, who in ((Q1, H1, "fused"), (Q0, H0, "launch flow")):
        for j in sorted(set((0, k // 2, k - 1))):
            lhs = orc.spmv(Ao, Q[:, j])
            assert np.abs(lhs - Q @ H[:, j]).max() <= 1e-10 * max(np.abs(lhs).max(), hs), (case, n, kn, who, "Arnoldi relation, column", j)
    loss = np.abs(Q0.T @ Q0 - np.eye(Q0.shape[1])).max()
    if loss <= 1e-10:
        compared += 1
        assert np.abs(H1 - H0).max() <= 1e-7 * hs and np.abs(Q1 - Q0).max() <= 1e-6, (case, n, kn, loss, np.abs(H1 - H0).max() / hs, np.abs(Q1 - Q0).max())
        assert np.abs(Q1.T @ Q1 - np.eye(Q1.shape[1])).max() <= 1e-8, (case, n, kn, "orthonormality of the fused basis")
print(f"arnoldi fuzz ok: {cases} cases, fused step taken in {fused}, compared entry by entry (well-conditioned bases) in {compared}; Arnoldi relation checked in all")
