// sla_solvers.cpp -- host drivers of the on-device solver loops (C ABI rows A5..A10).
//
// The host only ENQUEUES kernels; every scalar (alpha, omega, beta, rho, residual norm, the
// convergence decision) lives in device memory (sla::SolverScalars).  linSolve0's per-iteration
// "recompute the true residual and test it" (Sparse.hs:1043-1052) runs on the device: once the test
// passes, every later kernel of the batch returns immediately, so the iterate handed back is exactly
// the reference's x' of the first converged step, while the host polls only every `check_every` steps.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "sla_internal.hpp"

using namespace sla;

namespace {

// (P_ASS .. P_SR0 are consecutive: the four sums of the fused K4+K5 flow are folded and exchanged as one block, publish4)
enum Slot { P_APR = 0, P_ASS = 1, P_ASAS = 2, P_TR0 = 3, P_SR0 = 4, P_RHO = 5, P_RES = 6, P_TMP = 7, P_SLOTS = 8 };

double *slot(sla_solver *S, int s) { return S->d_parts + (size_t)s * kMaxParts; }

// Make the partials a producer kernel just wrote (np per array) consumable: on one GPU they are used
// as they are; when sharded, each rank folds its partials to one value per array and the per-rank
// values are all-gathered, to be summed in rank order by every consumer.
int publish(sla_solver *S, int s1, int s2, int np, Parts *o1, Parts *o2) {
    sla_ctx *c = S->ctx;
    if (!c->collectives) {
        *o1 = Parts{slot(S, s1), np, 1};
        if (o2) *o2 = Parts{s2 >= 0 ? slot(S, s2) : nullptr, np, 1};
        return SLA_OK;
    }
    double *loc = S->d_gath + (size_t)P_SLOTS * 2 * c->nranks;  // 2 doubles of staging
    ProfScope prof(c, SLA_KERNEL_SUMS);
    SLA_TRY(launch_finalize(c, slot(S, s1), s2 >= 0 ? slot(S, s2) : nullptr, np, loc));   // loc[1] = 0 without a second array
    double *g = S->d_gath + (size_t)s1 * 2 * c->nranks;
    SLA_TRY(dist_allgather_f64(c, loc, g, 2));
    *o1 = Parts{g, c->nranks, 2};
    if (o2) *o2 = Parts{g + 1, c->nranks, 2};
    return SLA_OK;
}

// ghost-row flow: the per-rank sums AND the neighbours' rows of `halo` (in place, around its own rows) in one grouped launch
static int publish_with_halo(sla_solver *S, int s1, int s2, int np, Parts *o1, Parts *o2, sla_vec *halo) {
    sla_ctx *c = S->ctx;
    double *g = S->d_gath + (size_t)s1 * 2 * c->nranks;
    double *loc = g + 2 * c->rank;   // in place: this rank's sums are written straight into their slot of the gathered table
    ProfScope prof(c, SLA_KERNEL_SUMS);
    SLA_TRY(launch_finalize(c, slot(S, s1), s2 >= 0 ? slot(S, s2) : nullptr, np, loc));
    // (the per-rank sums travel as point-to-point transfers too, so that the group is a pure send/recv group)
    SLA_TRY(dist_group_begin(c));
    int rc = dist_allgather_p2p_f64(c, loc, g, 2);
    if (rc == SLA_OK) rc = dist_exchange_window(c, *S->A->xplan, halo->d, halo->begin, halo->n_local, halo->d - halo->begin);
    const int rc_end = dist_group_end(c);
    if (rc != SLA_OK) return rc;
    SLA_TRY(rc_end);
    *o1 = Parts{g, c->nranks, 2};
    if (o2) *o2 = Parts{g + 1, c->nranks, 2};
    return SLA_OK;
}

// The four sums K3 leaves for the fused K4+K5 sweep (As . s, As . As, As . r0hat, s . r0hat; slots P_ASS .. P_SR0) made consumable.
// One GPU: as they are.  Sharded: ONE fold launch (four columns), the per-rank quadruples all-gathered -- and, in the ghost-row
// flow, the neighbours' rows of `halo` (As) in the same grouped launch, so that the sweep can run on own + ghost rows.
static int publish4(sla_solver *S, int np, Parts out[4], sla_vec *halo) {
    sla_ctx *c = S->ctx;
    if (!c->collectives) {
        for (int j = 0; j < 4; ++j) out[j] = Parts{slot(S, P_ASS + j), np, 1};
        return SLA_OK;
    }
    ProfScope prof(c, SLA_KERNEL_SUMS);
    double *g = S->d_gath + (size_t)P_ASS * 2 * c->nranks;   // 4 doubles per rank: the tables of P_ASS and P_ASAS are adjacent
    double *loc = g + 4 * c->rank;                            // in place: this rank's sums go straight into their slot
    SLA_TRY(launch_finalize_cols(c, slot(S, P_ASS), np, kMaxParts, 1, 4, loc));
    if (halo) {
        SLA_TRY(dist_group_begin(c));
        int rc = dist_allgather_p2p_f64(c, loc, g, 4);
        if (rc == SLA_OK) rc = dist_exchange_window(c, *S->A->xplan, halo->d, halo->begin, halo->n_local, halo->d - halo->begin);
        const int rc_end = dist_group_end(c);
        if (rc != SLA_OK) return rc;
        SLA_TRY(rc_end);
    } else {
        SLA_TRY(dist_allgather_f64(c, loc, g, 4));
    }
    for (int j = 0; j < 4; ++j) out[j] = Parts{g + j, c->nranks, 4};
    return SLA_OK;
}

int solver_alloc(sla_csr *A, int method, sla_solver **out) {
    sla_ctx *c = A->ctx;
    sla_solver *S = new sla_solver();
    S->ctx = c;
    S->A = A;
    S->method = method;
    const int64_t nx = A->n, nr = A->m;
    int rc = SLA_OK;
    auto mk = [&](int64_t n, sla_vec **v) { if (rc == SLA_OK) rc = vec_alloc(c, n, v); };
    mk(nx, &S->x);
    mk(nr, &S->r);
    mk(nx, &S->p);
    mk(nr, &S->r0hat);
    mk(nr, &S->b);
    if (method == SLA_CGS_ || method == SLA_BCG_) mk(nr, &S->u);   // (BCG: u holds phat)
    if (method != SLA_CGNE_) { mk(nr, &S->t1); mk(nr, &S->t2); mk(nr, &S->t3); }
    else if (c->collectives) mk(nx, &S->t1);  // CGNE, row-sharded: landing buffer of the reduce-scattered A^T r
    hipError_t e = hipSuccess;
    if (rc == SLA_OK) e = dev_malloc(c, (void **)&S->d_parts, sizeof(double) * P_SLOTS * kMaxParts);
    if (rc == SLA_OK && e == hipSuccess) e = dev_malloc(c, (void **)&S->d_gath, sizeof(double) * ((size_t)P_SLOTS * 2 * c->nranks + 8));
    if (rc == SLA_OK && e == hipSuccess) e = dev_malloc(c, (void **)&S->d_sc, sizeof(SolverScalars));
    if (rc == SLA_OK && e == hipSuccess) e = hipHostMalloc((void **)&S->h_sc, sizeof(SolverScalars), hipHostMallocDefault);
    if (rc == SLA_OK && e != hipSuccess) rc = fail(SLA_ERR_ALLOC, std::string("solver allocation: ") + hipGetErrorString(e));
    if (rc != SLA_OK) {
        sla_solver_destroy(S);
        return rc;
    }
    *out = S;
    return SLA_OK;
}

struct StepCtl {
    int step_index = 0;  // parity source
    Parts res{nullptr, 0, 1};
    Parts pp{nullptr, 0, 1};  // CGNE: p . p partials of the current p
};

// true residual of the current x: partials of ||A x - b||^2   (trueResidualNorm, Sparse.hs:1041)
int enqueue_residual(sla_solver *S, Parts *res) {
    SpmvLaunch l;
    l.epi = EPI_RES;
    l.w = S->b->d;
    l.p1 = slot(S, P_RES);
    l.sc = S->d_sc;
    l.kernel_id = SLA_KERNEL_SPMV_RES;
    int gk = 0;
    SLA_TRY(spmv_exchanged(S->A, S->x, l, &gk));
    return publish(S, P_RES, -1, gk, res, nullptr);
}

// bicgstabStep (Sparse.hs:972-981)
// bicgstabStep on a row slab with GHOST rows.  The plain sharded flow below exchanges the halo of p before K1 and of s
// before K3 and all-gathers three groups of partial sums: five dependent collectives per step, which is what bounds
// strong scaling (the slab kernels take ~10 us each).  Here r, p, Ap and s are kept valid on the ghost rows as well:
//   halo(Ap) travels with the alpha partials (Ap is complete after K1), K2 then forms s = r - alpha Ap on own + ghost rows;
//   halo(r') travels with the rho partials (r' is complete after K4), K5 then forms p' = r' + beta (p - omega Ap) on own +
//   ghost rows from halo(r'), halo(p) (kept from the previous step) and halo(Ap) (still in place).
// Three grouped launches per step, no exchange in front of either SpMV; every ghost value is computed by the same
// kernel from the same bits as on its owner, so the iterates are bit-identical to the plain flow.
// With the fused K4+K5 sweep (SLA_BICG_FUSE45, default) it is TWO: halo(As) travels with K3's four sums and the sweep runs
// on own + ghost rows (see below), which makes r' and p' valid there without a third exchange.
static int enqueue_bicgstab_ghost(sla_solver *S, int par, const Parts *check) {
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A;
    const int64_t n = S->x->n_local, b = S->x->begin;
    const int64_t gl = S->ghl + (S->ghl & 1);        // (even, so that the extended kernels keep their 16-byte pairs aligned;
    const int64_t next = n + gl + S->ghr;             //  the extra element lies in the allocation's slack and is never read)
    const int g = spmv_grid(A);
    Parts apr, ass, asas, rhon;
    {
        SpmvLaunch l;  // K1: aap = aa #> p ; aap <.> r0hat        (halo(p) is valid: no exchange)
        l.epi = EPI_DOT;
        l.x = S->p->d - b;
        l.y = S->t1->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_APR);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = 1 | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        SLA_TRY(launch_spmv(A, l));
        SLA_TRY(publish_with_halo(S, P_APR, -1, g, &apr, nullptr, S->t1));
    }
    // (K2 folded into K3 where the whole-slab launch runs the plane-march or the gather kernel -- see enqueue_bicgstab; a rank's own decision, no
    // collective depends on it: r and Ap are valid on the ghost
    // rows, so the staged windows and the sweep's rebuilt s hold there what K2 would have written)
    const bool fuse23 = c->bicg_fuse23 != 0 && c->bicg_fuse45 != 0 && spmv_fuse_s_ok(A, true);
    if (!fuse23) SLA_TRY(launch_bicg_k2(c, next, S->d_sc, apr, par, Parts{nullptr, 0, 1}, 0, S->r->d - gl, S->t1->d - gl, S->t2->d - gl));
    {
        SpmvLaunch l;  // K3: aasj = aa #> sj ; aasj <.> sj ; aasj <.> aasj      (halo(s) was computed by K2)
        l.epi = EPI_DOT2;
        l.x = S->t2->d - b;
        l.y = S->t3->d;
        l.w = S->t2->d;
        l.p1 = slot(S, P_ASS);
        l.p2 = slot(S, P_ASAS);
        l.sc = S->d_sc;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        if (c->bicg_fuse45) {
            // Fused K4+K5 on own + ghost rows: K3 also sums As . r0hat and s . r0hat (rho' by linearity, bicg_k45_kernel), halo(As)
            // travels WITH the four sums, and the sweep then forms x', r' and p' on the ghost rows too from s (K2 left it there),
            // As and Ap (their halos are in place) and the p it kept -- so NO third exchange and no K5: two grouped exchanges and
            // four kernels per step.  (x rides along over the ghost range: its slack holds nothing anybody reads.)
            l.epi = EPI_DOT4;
            l.z = S->r0hat->d;
            l.p3 = slot(S, P_TR0);
            l.p4 = slot(S, P_SR0);
            if (fuse23) {
                l.x = S->r->d - b;
                l.fs_ap = S->t1->d - b;
                l.w = nullptr;
                l.pa = apr.p;
                l.npa = apr.n;
                l.pa_stride = apr.stride;
                l.step_begin = par << 1;
            }
            SLA_TRY(launch_spmv(A, l));
            Parts q[4];
            SLA_TRY(publish4(S, g, q, S->t3));
            return launch_bicg_k45(c, next, S->d_sc, q[0], q[1], q[2], q[3], par, fuse23 ? nullptr : S->t2->d - gl, S->t3->d - gl, S->t1->d - gl, S->x->d - gl,
                                   S->r->d - gl, S->p->d - gl);
        }
        SLA_TRY(launch_spmv(A, l));
        SLA_TRY(publish(S, P_ASS, P_ASAS, g, &ass, &asas));
    }
    SLA_TRY(launch_bicg_k4(c, n, S->d_sc, ass, asas, S->p->d, S->t2->d, S->t3->d, S->r0hat->d, S->x->d, S->r->d, slot(S, P_RHO)));
    SLA_TRY(publish_with_halo(S, P_RHO, -1, vec_grid(n), &rhon, nullptr, S->r));
    SLA_TRY(launch_bicg_k5(c, next, S->d_sc, rhon, par, S->r->d - gl, S->t1->d - gl, S->p->d - gl));
    return SLA_OK;
}

int enqueue_bicgstab(sla_solver *S, int par, const Parts *check, bool dual_prev) {
    if (S->ghost) return enqueue_bicgstab_ghost(S, par, check);
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A;
    const int64_t n = S->x->n_local;
    const int g = spmv_grid(A);
    Parts apr, ass, asas, rhon;
    {
        SpmvLaunch l;  // K1: aap = aa #> p ; aap <.> r0hat
        l.epi = EPI_DOT;
        l.y = S->t1->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_APR);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = (dual_prev ? 0 : 1) | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        if (dual_prev) {  // the same matrix sweep also evaluates ||A x - b||^2 of the current x
            l.x2 = S->x->d;
            l.b2 = S->b->d;
            l.p2 = slot(S, P_RES);
            l.kernel_id = SLA_KERNEL_SPMV_DUAL;
        }
        int gk = g;   // (row-sharded: the interior rows run while the halo of p is in flight, spmv_exchanged)
        SLA_TRY(spmv_exchanged(A, S->p, l, &gk));
        SLA_TRY(publish(S, P_APR, -1, gk, &apr, nullptr));
    }
    // Round 5: where K3 runs on the plane-march kernel, or on the gather kernel of the wave-sliced forms up to 4 M rows (spmv_fuse_s_ok), K2 is
    // folded into it -- s = r - alpha Ap is built while the x windows are staged (gather kernel: from the gathered row pairs of r and Ap) and never stored; the fused K4+K5 sweep rebuilds it from r and Ap, which it reads anyway.  Three
    // launches and 121 n bytes per step instead of four and 138 n; the same s and As bit for bit (bicg_k2_kernel's alpha and multiply-add -- the
    // iterates differ in the last bits only where the folded instantiation's occupancy regroups K3's fused sums: INTEGRATION.md).
    const bool fuse23 = c->bicg_fuse23 != 0 && c->bicg_fuse45 != 0 && !dual_prev && spmv_fuse_s_ok(A);
    if (!fuse23)
        SLA_TRY(launch_bicg_k2(c, n, S->d_sc, apr, par, dual_prev ? Parts{slot(S, P_RES), g, 1} : Parts{nullptr, 0, 1},
                               dual_prev ? 1 : 0, S->r->d, S->t1->d, S->t2->d));
    {
        SpmvLaunch l;  // K3: aasj = aa #> sj ; aasj <.> sj ; aasj <.> aasj
        l.epi = EPI_DOT2;
        l.y = S->t3->d;
        l.w = S->t2->d;
        l.p1 = slot(S, P_ASS);
        l.p2 = slot(S, P_ASAS);
        l.sc = S->d_sc;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        // the sweep also sums As . r0hat and s . r0hat, which give rho_{j+1} before r_{j+1} exists, so that K4 and K5 become one
        // sweep (bicg_k45_kernel) -- on one GPU and, since round 3, on row-sharded contexts too (publish4)
        // (the variable-coefficient wave-sliced kernel is register-bound at 5 workgroups per CU and spills with the third operand,
        // so its EPI_DOT4 instantiation is compiled for 4: 2 M-row banded problem, same box, three interleaved runs each: split
        // 11 501-11 581 it/s, fused 11 963-12 055)
        const bool fuse = c->bicg_fuse45 != 0;
        if (fuse) {
            l.epi = EPI_DOT4;
            l.z = S->r0hat->d;
            l.p3 = slot(S, P_TR0);
            l.p4 = slot(S, P_SR0);
        }
        int gk = g;
        if (fuse23) {
            l.x = S->r->d;          // (one rank: the gather base is the vector itself)
            l.fs_ap = S->t1->d;
            l.w = nullptr;          // (As . s takes s from the staged window)
            l.pa = apr.p;
            l.npa = apr.n;
            l.pa_stride = apr.stride;
            l.step_begin = par << 1;
            l.kernel_id = SLA_KERNEL_SPMV_DOT2;
            SLA_TRY(launch_spmv(A, l));
        } else {
            SLA_TRY(spmv_exchanged(A, S->t2, l, &gk));
        }
        if (fuse) {   // (row-sharded: the four sums travel as ONE all-gather; the rho exchange and K5 are gone)
            Parts q[4];
            SLA_TRY(publish4(S, gk, q, nullptr));
            return launch_bicg_k45(c, n, S->d_sc, q[0], q[1], q[2], q[3], par, fuse23 ? nullptr : S->t2->d, S->t3->d, S->t1->d, S->x->d, S->r->d, S->p->d);
        }
        SLA_TRY(publish(S, P_ASS, P_ASAS, gk, &ass, &asas));
    }
    SLA_TRY(launch_bicg_k4(c, n, S->d_sc, ass, asas, S->p->d, S->t2->d, S->t3->d, S->r0hat->d, S->x->d, S->r->d, slot(S, P_RHO)));
    SLA_TRY(publish(S, P_RHO, -1, vec_grid(n), &rhon, nullptr));
    SLA_TRY(launch_bicg_k5(c, n, S->d_sc, rhon, par, S->r->d, S->t1->d, S->p->d));
    return SLA_OK;
}

// cgsStep on a row slab with ghost rows (see enqueue_bicgstab_ghost): u, p, aap, q and u + q are kept valid on the ghost
// rows.  halo(aap) travels with the alpha partials, C2 then forms q and u + q on own + ghost rows; halo(r') travels with
// the rho partials, C4 then forms u' and p' on own + ghost rows.  Two grouped exchanges per step instead of four, same bits.
static int enqueue_cgs_ghost(sla_solver *S, int par, const Parts *check) {
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A;
    const int64_t n = S->x->n_local, b = S->x->begin;
    const int64_t gl = S->ghl + (S->ghl & 1), next = n + gl + S->ghr;
    const int g = spmv_grid(A);
    Parts apr, rhon;
    {
        SpmvLaunch l;  // C1: aap = aa #> p ; aap <.> rhat          (halo(p) is valid: no exchange)
        l.epi = EPI_DOT;
        l.x = S->p->d - b;
        l.y = S->t1->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_APR);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = 1 | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        SLA_TRY(launch_spmv(A, l));
        SLA_TRY(publish_with_halo(S, P_APR, -1, g, &apr, nullptr, S->t1));
    }
    // (x rides along over the ghost range: its slack holds nothing anybody reads)
    SLA_TRY(launch_cgs_c2(c, next, S->d_sc, apr, par, Parts{nullptr, 0, 1}, 0, S->u->d - gl, S->t1->d - gl, S->t2->d - gl,
                          S->t3->d - gl, S->x->d - gl));
    {
        SpmvLaunch l;  // C3: rj1 = r ^-^ alphaj .* (aa #> (u ^+^ q)) ; rj1 <.> rhat      (halo(u + q) was computed by C2)
        l.epi = EPI_AXPY_DOT;
        l.x = S->t3->d - b;
        l.z = S->r->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_RHO);
        l.sc = S->d_sc;
        l.step_begin = par << 1;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        SLA_TRY(launch_spmv(A, l));
        SLA_TRY(publish_with_halo(S, P_RHO, -1, g, &rhon, nullptr, S->r));
    }
    SLA_TRY(launch_cgs_c4(c, next, S->d_sc, rhon, par, S->r->d - gl, S->t2->d - gl, S->u->d - gl, S->p->d - gl));
    return SLA_OK;
}

// cgsStep (Sparse.hs:928-939)
int enqueue_cgs(sla_solver *S, int par, const Parts *check, bool dual_prev) {
    if (S->ghost) return enqueue_cgs_ghost(S, par, check);
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A;
    const int64_t n = S->x->n_local;
    const int g = spmv_grid(A);
    Parts apr, rhon;
    {
        SpmvLaunch l;  // C1: aap = aa #> p ; aap <.> rhat
        l.epi = EPI_DOT;
        l.y = S->t1->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_APR);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = (dual_prev ? 0 : 1) | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        if (dual_prev) {
            l.x2 = S->x->d;
            l.b2 = S->b->d;
            l.p2 = slot(S, P_RES);
            l.kernel_id = SLA_KERNEL_SPMV_DUAL;
        }
        int gk = g;
        SLA_TRY(spmv_exchanged(A, S->p, l, &gk));
        SLA_TRY(publish(S, P_APR, -1, gk, &apr, nullptr));
    }
    // Round 5, plane-march form on one rank (see enqueue_bicgstab): C2 is folded away -- C3 builds u + q in its staged windows from u and
    // A p (q = u - alpha A p), and ONE sweep after it does C2's x update and C4's u, p (cgs_c24_kernel): three launches, 129 n bytes per step
    // instead of four and 146 n, the same bits
    if (c->bicg_fuse23 != 0 && !dual_prev && spmv_fuse_s_ok(A) && wd_march_on(A)) {
        SpmvLaunch l;
        l.epi = EPI_AXPY_DOT;
        l.x = S->u->d;
        l.fs_ap = S->t1->d;
        l.pa = apr.p;
        l.npa = apr.n;
        l.pa_stride = apr.stride;
        l.z = S->r->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_RHO);
        l.sc = S->d_sc;
        l.step_begin = par << 1;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        SLA_TRY(launch_spmv(A, l));
        SLA_TRY(publish(S, P_RHO, -1, g, &rhon, nullptr));
        return launch_cgs_c24(c, n, S->d_sc, rhon, par, S->r->d, S->t1->d, S->u->d, S->p->d, S->x->d);
    }
    SLA_TRY(launch_cgs_c2(c, n, S->d_sc, apr, par, dual_prev ? Parts{slot(S, P_RES), g, 1} : Parts{nullptr, 0, 1},
                          dual_prev ? 1 : 0, S->u->d, S->t1->d, S->t2->d, S->t3->d, S->x->d));
    {
        SpmvLaunch l;  // C3: rj1 = r ^-^ alphaj .* (aa #> (u ^+^ q)) ; rj1 <.> rhat
        l.epi = EPI_AXPY_DOT;
        l.z = S->r->d;
        l.w = S->r0hat->d;
        l.p1 = slot(S, P_RHO);
        l.sc = S->d_sc;
        l.step_begin = par << 1;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        int gk = g;
        SLA_TRY(spmv_exchanged(A, S->t3, l, &gk));
        SLA_TRY(publish(S, P_RHO, -1, gk, &rhon, nullptr));
    }
    SLA_TRY(launch_cgs_c4(c, n, S->d_sc, rhon, par, S->r->d, S->t2->d, S->u->d, S->p->d));
    return SLA_OK;
}

// cgneStep (Sparse.hs:870-878); ctl->pp carries p . p of the current p
int enqueue_cgne(sla_solver *S, int par, const Parts *check, StepCtl *ctl) {
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A, *T = nullptr;
    SLA_TRY(csr_transposed(A, &T));
    Parts rr1;
    const bool sharded = c->collectives;
    {
        SpmvLaunch l;  // N1: alphai = (r.r)/(p.p) ; r1 = r ^-^ alphai .* (aa #> p) ; r1 . r1
        l.epi = EPI_AXPY_DOT;
        l.z = S->r->d;
        l.w = nullptr;
        l.pa = ctl->pp.p;
        l.npa = ctl->pp.n;
        l.pa_stride = ctl->pp.stride;
        l.p1 = slot(S, P_RHO);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = 1 | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        int gk = 0;
        SLA_TRY(spmv_exchanged(A, S->p, l, &gk));
        SLA_TRY(publish(S, P_RHO, -1, gk, &rr1, nullptr));
    }
    SLA_TRY(launch_cgne_n2(c, S->x->n_local, S->d_sc, S->p->d, S->x->d));  // x1 = x ^+^ alphai .* p
    if (sharded) {
        // N3 unfused: t = transpose aa #> r1 (local partial + reduce-scatter), then p1 = t ^+^ beta .* p ; p1 . p1
        SLA_TRY(spmv_transposed(A, S->r->d, S->t1->d, S->t1->shard));
        SLA_TRY(launch_cgne_n3b(c, S->p->n_local, S->d_sc, rr1, par, S->t1->d, S->p->d, slot(S, P_ASS)));
        SLA_TRY(publish(S, P_ASS, -1, vec_grid(S->p->n_local), &ctl->pp, nullptr));
    } else {
        SpmvLaunch l;  // N3: beta = (r1.r1)/(r.r) ; p1 = transpose aa #> r1 ^+^ beta .* p ; p1 . p1
        l.epi = EPI_XPBY_NRM;
        SLA_TRY(gather_x(S->A, S->r, &l.x));
        l.z = S->p->d;
        l.pa = rr1.p;
        l.npa = rr1.n;
        l.pa_stride = rr1.stride;
        l.p1 = slot(S, P_ASS);
        l.sc = S->d_sc;
        l.step_begin = par << 1;
        l.kernel_id = SLA_KERNEL_SPMV_DOT2;
        SLA_TRY(launch_spmv(T, l));
        SLA_TRY(publish(S, P_ASS, -1, spmv_grid(T), &ctl->pp, nullptr));
    }
    return SLA_OK;
}

// bcgStep -- an EXTENSION: the reference keeps it commented out (Sparse.hs:899-909) and `linSolve0 BCG_` throws (:1031; sla_linsolve0 keeps
// doing that).  State record BCG x r rhat p phat (:886-887): rhat lives in S->r0hat, phat in S->u.  One (#>), one (<#) and two sweeps:
//   B1  aap = aa #> p ; aap <.> phat                              (SpMV, EPI_DOT)
//   B2  transpose aa #> phat                                      (SpMV on the transposed copy; row-sharded: partial + reduce-scatter)
//   B3  alpha ; x1, r1, rhat1 ; r1 <.> rhat1                      (bcg_b3_kernel)
//   B4  beta ; p1 = r1 ^+^ beta .* p ; phat1 = rhat1 ^+^ beta .* phat   (bcg_b4_kernel)
// rho = r <.> rhat is carried from step to step (rho2[par], as in the other methods): the same sum the reference would recompute.
int enqueue_bcg(sla_solver *S, int par, const Parts *check) {
    sla_ctx *c = S->ctx;
    sla_csr *A = S->A;
    const int64_t n = S->x->n_local;
    Parts app, rr1;
    {
        SpmvLaunch l;
        l.epi = EPI_DOT;
        l.y = S->t1->d;
        l.w = S->u->d;
        l.p1 = slot(S, P_APR);
        l.sc = S->d_sc;
        if (check && check->p) { l.pres = check->p; l.npres = check->n; l.pres_stride = check->stride; }
        l.step_begin = 1 | (par << 1);
        l.kernel_id = SLA_KERNEL_SPMV_DOT;
        int gk = 0;
        SLA_TRY(spmv_exchanged(A, S->p, l, &gk));
        SLA_TRY(publish(S, P_APR, -1, gk, &app, nullptr));
    }
    SLA_TRY(spmv_transposed(A, S->u->d, S->t2->d, S->t2->shard));
    SLA_TRY(launch_bcg_b3(c, n, S->d_sc, app, par, S->p->d, S->t1->d, S->t2->d, S->x->d, S->r->d, S->r0hat->d, slot(S, P_RHO)));
    SLA_TRY(publish(S, P_RHO, -1, vec_grid(n), &rr1, nullptr));
    SLA_TRY(launch_bcg_b4(c, n, S->d_sc, rr1, par, S->r->d, S->r0hat->d, S->p->d, S->u->d));
    return SLA_OK;
}

StepCtl &ctl_of(sla_solver *S) {
    static_assert(sizeof(StepCtl) <= 128, "StepCtl fits the solver's opaque block");
    return *reinterpret_cast<StepCtl *>(S->ctl_storage);
}

// One solver step.  res_after: follow it with the stand-alone true-residual SpMV (KR).  dual_prev: this
// step's K1 also evaluates the true residual of the CURRENT x (the previous step's x') and K2 tests it --
// the reference's "step; recompute residual; test" order with two matrix sweeps per iteration, not three.
int enqueue_step(sla_solver *S, bool res_after, bool dual_prev) {
    StepCtl &ctl = ctl_of(S);
    const int par = ctl.step_index & 1;
    const Parts *check = S->have_res ? &ctl.res : nullptr;
    if (S->method == SLA_BICGSTAB_) SLA_TRY(enqueue_bicgstab(S, par, check, dual_prev));
    else if (S->method == SLA_CGS_) SLA_TRY(enqueue_cgs(S, par, check, dual_prev));
    else if (S->method == SLA_BCG_) SLA_TRY(enqueue_bcg(S, par, check));
    else SLA_TRY(enqueue_cgne(S, par, check, &ctl));
    ctl.step_index++;
    S->have_res = false;
    if (res_after) {
        SLA_TRY(enqueue_residual(S, &ctl.res));
        S->have_res = true;
    }
    return SLA_OK;
}

// the dual-SpMV flow needs x and p resident on this rank as whole vectors and the stream kernel
bool dual_ok(const sla_solver *S) {
    // (with column panels the residual SpMV is cheaper as its own panel-blocked sweep than fused into K1)
    return !S->ctx->collectives && S->ctx->spmv_algo == 0 && S->ctx->dual_spmv && S->method != SLA_CGNE_ && S->method != SLA_BCG_ &&
           (S->A->panels.empty() || !S->ctx->panels) &&
           // (the wave-sliced form streams ~2 B of matrix per row: fusing the two sweeps saves nothing there)
           !(S->A->use_wdia && wd_on(S->A)) &&
           // (the LDS-panel form has no fused two-vector variant; two of its sweeps beat one L2-gathering dual sweep)
           !(S->A->use_lpanel && S->ctx->lpanel) && !lflat_on(S->A) && !tiles_on(S->A);
}

int read_scalars(sla_solver *S) {
    sla_ctx *c = S->ctx;
    SLA_HIP_TRY(hipMemcpyAsync(S->h_sc, S->d_sc, sizeof(SolverScalars), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    return SLA_OK;
}

// *Init (Sparse.hs:921-924, 962-965, 864-868) + the tolerance of linSolve0 (:1032-1037)
int solver_init_common(int method, sla_csr *A, sla_vec *b, sla_vec *x0, double tol_abs, double tol_rel, sla_solver **out, int hist_cap = 0) {
    if (!A || !b || !x0 || !out) return fail(SLA_ERR_INVALID, "solver init: null argument");
    // (SLA_BCG_ is an extension of the STATE-RECORD interface only -- the reference's bcgStep is commented out, Sparse.hs:886-909;
    // sla_linsolve0 rejects it like the reference's linSolve0 does)
    if (method != SLA_BICGSTAB_ && method != SLA_CGS_ && method != SLA_CGNE_ && method != SLA_BCG_)
        return fail(SLA_ERR_UNSUPPORTED_METHOD, "Only BICGSTAB_, CGS_, and CGNE_ are implemented");
    if (A->m != b->n) return fail(SLA_ERR_DIM_MISMATCH, "linSolve0 : matrix rows and rhs dimension differ");
    if (A->n != x0->n) return fail(SLA_ERR_DIM_MISMATCH, "matVec : mismatched dimensions");
    if (method != SLA_CGNE_ && A->m != A->n) return fail(SLA_ERR_DIM_MISMATCH, "CGS/BiCGSTAB/BCG need a square matrix");
    sla_ctx *c = A->ctx;
    if (!b->kids.empty() || !x0->kids.empty() || b->ctx != c || x0->ctx != c) return mixed_handles("solver init");
    Bind bind(c);
    sla_solver *S = nullptr;
    {   // a rank whose allocation fails must not leave its peers blocked in the first collective below: agree first
        int rc_alloc = solver_alloc(A, method, &S);
        const std::string msg = rc_alloc != SLA_OK ? sla_last_error() : "";
        int bad = rc_alloc != SLA_OK ? 1 : 0;
        if (c->collectives) SLA_TRY(dist_allreduce_max_i32(c, &bad));
        if (rc_alloc != SLA_OK) return fail(rc_alloc, msg);
        if (bad) {
            sla_solver_destroy(S);
            return fail(SLA_ERR_ALLOC, "solver state allocation failed on another rank of the row-sharded job");
        }
    }
    new (&ctl_of(S)) StepCtl();
    int rc = SLA_OK;
    do {
        if (hist_cap > 0) {   // (a local failure here is reported after the collectives below have run their course: no rank is left waiting)
            if (dev_malloc(c, (void **)&S->d_hist, sizeof(double) * (size_t)hist_cap) == hipSuccess &&
                hipMemsetAsync(S->d_hist, 0, sizeof(double) * (size_t)hist_cap, stream_of(c)) == hipSuccess)
                S->hist_cap = hist_cap;
        }
        if ((rc = sla_vec_copy(x0, S->x)) != SLA_OK) break;
        if ((rc = sla_vec_copy(b, S->b)) != SLA_OK) break;
        SpmvLaunch l;  // r0 = b ^-^ (aa #> x0)
        l.epi = EPI_SUB;
        l.y = S->r->d;
        l.w = S->b->d;
        if ((rc = spmv_exchanged(A, S->x, l, nullptr)) != SLA_OK) break;
        if ((rc = sla_vec_copy(S->r, S->r0hat)) != SLA_OK) break;
        if (method == SLA_CGNE_) {
            // p0 = transposeSM aa #> r0
            if ((rc = spmv_transposed(A, S->r->d, S->p->d, S->p->shard)) != SLA_OK) break;
            if ((rc = launch_dot(c, S->p->n_local, S->p->d, S->p->d, slot(S, P_ASS))) != SLA_OK) break;
            if ((rc = publish(S, P_ASS, -1, vec_grid(S->p->n_local), &ctl_of(S).pp, nullptr)) != SLA_OK) break;
        } else {
            if ((rc = sla_vec_copy(S->r, S->p)) != SLA_OK) break;
            if ((method == SLA_CGS_ || method == SLA_BCG_) && (rc = sla_vec_copy(S->r, S->u)) != SLA_OK) break;   // u0 = r0 (CGS) / p0hat = r0hat = r0 (BCG)
            if (method == SLA_BCG_) {   // the transposed copy is built here, not inside the first step (a captured step graph must not allocate)
                sla_csr *T = nullptr;
                if ((rc = csr_transposed(A, &T)) != SLA_OK) break;
            }
        }
        if ((method == SLA_BICGSTAB_ || method == SLA_CGS_) && c->collectives && c->bicg_ghost) {
            // ghost-row flow (enqueue_bicgstab_ghost / enqueue_cgs_ghost): every rank must take the same decision -- the
            // collectives differ
            int64_t gl = 0, gr = 0;
            int not_ok = (A->m == A->n && S->r->shard == S->p->shard && halo_inplace_extents(A, S->p, &gl, &gr)) ? 0 : 1;
            if ((rc = dist_allreduce_max_i32(c, &not_ok)) != SLA_OK) break;
            if (!not_ok) {
                S->ghost = true;
                S->ghl = gl;
                S->ghr = gr;
                if (getenv("SLA_DEBUG_EXCHANGE"))
                    fprintf(stderr, "[sla] rank %d: ghost-row %s, %lld + %lld ghost rows\n", c->rank, method == SLA_CGS_ ? "CGS" : "BiCGSTAB", (long long)gl, (long long)gr);
                // halo(p0) and halo(r0) (BiCGSTAB) / halo(u0) (CGS): the invariant every step starts from
                sla_vec *second = method == SLA_CGS_ ? S->u : S->r;
                if ((rc = dist_exchange_window(c, *A->xplan, S->p->d, S->p->begin, S->p->n_local, S->p->d - S->p->begin)) != SLA_OK) break;
                if ((rc = dist_exchange_window(c, *A->xplan, second->d, second->begin, second->n_local, second->d - second->begin)) != SLA_OK) break;
            }
        }
        // rho = r0 . r0hat = r0 . r0 ; r0norm = sqrt (r0 . r0)
        Parts rho;
        if ((rc = launch_dot(c, S->r->n_local, S->r->d, S->r->d, slot(S, P_TMP))) != SLA_OK) break;
        if ((rc = publish(S, P_TMP, -1, vec_grid(S->r->n_local), &rho, nullptr)) != SLA_OK) break;
        if ((rc = launch_init_scalars(c, S->d_sc, rho, rho, tol_abs, tol_rel, S->hist_cap ? S->d_hist : nullptr, S->hist_cap)) != SLA_OK) break;
    } while (0);
    if (rc != SLA_OK) {
        sla_solver_destroy(S);
        return rc;
    }
    *out = S;
    return SLA_OK;
}

void fill_info(sla_solver *S, sla_solve_info *info, bool hit_max) {
    if (!info) return;
    const SolverScalars &h = *S->h_sc;
    info->iters = h.iters;
    info->flags = h.flags | (h.done ? 0 : (hit_max ? SLA_FLAG_MAX_ITERS : 0)) | (csr_fold_relaxed(S->A) ? SLA_FLAG_RELAXED_ORDER : 0);
    info->resnorm = h.resnorm;
    info->r0norm = h.r0norm;
    info->tol = h.tol;
}

// ---- Arnoldi workspace -------------------------------------------------------------------------------
struct ArnoldiWs {
    sla_ctx *c = nullptr;
    int64_t n = 0, n_local = 0, ld = 0;
    int kn = 0;
    double *Q = nullptr, *w = nullptr, *H = nullptr, *parts = nullptr, *gath = nullptr, *ycoef = nullptr;
    double *Qalloc = nullptr;   // what guard_free gets back (Q = Qalloc + halo when the basis columns carry halo slack)
    int64_t halo = 0;           // row-sharded, window exchange: slack on both sides of every basis column for the neighbours' planes
    int64_t begin = 0, shard = 0;
    SolverScalars *d_sc = nullptr, *h_sc = nullptr;
    unsigned *bar = nullptr;    // arrival counters + epoch word of the fused Gram-Schmidt step (sla_arnoldi_orth.hip); zeroed once
    std::vector<double> Hhost;
    ~ArnoldiWs() {
        if (bar) (void)hipFree(bar);
        if (Qalloc) (void)guard_free(Qalloc);
        if (w) (void)hipFree(w);
        if (H) (void)hipFree(H);
        if (parts) (void)hipFree(parts);
        if (gath) (void)hipFree(gath);
        if (ycoef) (void)hipFree(ycoef);
        if (d_sc) (void)hipFree(d_sc);
        if (h_sc) (void)hipHostFree(h_sc);
    }
};

int arn_alloc(ArnoldiWs &ws, sla_csr *A, sla_vec *like, int kn) {
    sla_ctx *c = A->ctx;
    ws.c = c;
    ws.n = like->n;
    ws.n_local = like->n_local;
    ws.ld = std::max<int64_t>(like->shard + (like->shard & 1), 2);  // even leading dimension: 16-byte aligned columns
    ws.kn = kn;
    ws.begin = like->begin;
    ws.shard = like->shard;
    {   // Row-sharded with the window exchange: give every basis column the slack the neighbours' halo planes need, so that
        // aa #> q_i receives them IN PLACE around the column (and overlaps the exchange with the interior rows, spmv_exchanged)
        // instead of going through the full-length landing buffer with a copy of the own rows.
        int64_t gl = 0, gr = 0;
        if (c->collectives && A->m == A->n && halo_inplace_extents(A, like, &gl, &gr)) {
            ws.halo = std::max(gl, gr) + 8;
            ws.halo += ws.halo & 1;
            ws.ld += 2 * ws.halo;
        }
    }
    const size_t qbytes = sizeof(double) * (size_t)ws.ld * (size_t)(kn + 1);
    SLA_HIP_TRY(guard_malloc(c, (void **)&ws.Qalloc, qbytes));  // SpMV gathers from its columns
    SLA_HIP_TRY(hipMemsetAsync(ws.Qalloc, 0, qbytes, stream_of(c)));
    ws.Q = ws.Qalloc + ws.halo;
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.w, sizeof(double) * (size_t)ws.ld));
    SLA_HIP_TRY(hipMemsetAsync(ws.w, 0, sizeof(double) * (size_t)ws.ld, stream_of(c)));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.H, sizeof(double) * (size_t)(kn + 1) * (size_t)kn));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.parts, sizeof(double) * (size_t)(kMaxKrylov + 2) * kArnGridMax));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.gath, sizeof(double) * ((size_t)(kMaxKrylov + 2) * (size_t)c->nranks + kMaxKrylov + 8)));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.ycoef, sizeof(double) * (kMaxKrylov + 2)));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.d_sc, sizeof(SolverScalars)));
    SLA_HIP_TRY(dev_malloc(c, (void **)&ws.bar, arn_orth_bar_bytes()));
    SLA_HIP_TRY(hipMemsetAsync(ws.bar, 0, arn_orth_bar_bytes(), stream_of(c)));
    SLA_HIP_TRY(hipHostMalloc((void **)&ws.h_sc, sizeof(SolverScalars), hipHostMallocDefault));
    ws.Hhost.assign((size_t)(kn + 1) * (size_t)kn, 0.0);
    return SLA_OK;
}

// arn_alloc + agreement over the ranks: a rank whose allocation fails must not leave its peers blocked in the first collective
int arn_alloc_agreed(ArnoldiWs &ws, sla_csr *A, sla_vec *like, int kn) {
    sla_ctx *c = A->ctx;
    const int rc = arn_alloc(ws, A, like, kn);
    const std::string msg = rc != SLA_OK ? sla_last_error() : "";
    int bad = rc != SLA_OK ? 1 : 0;
    if (c->collectives) SLA_TRY(dist_allreduce_max_i32(c, &bad));
    if (rc != SLA_OK) return fail(rc, msg);
    if (bad) return fail(SLA_ERR_ALLOC, "Arnoldi workspace allocation failed on another rank of the row-sharded job");
    return SLA_OK;
}

// per-column partials written by a producer -> what the consumer kernel should read
struct ColParts { const double *p; int np, cs, stride; };
int arn_publish(ArnoldiWs &ws, const double *parts, int np, int ncols, ColParts *out) {
    sla_ctx *c = ws.c;
    // one column on a single-rank context (the norms in front of arn_normalize_kernel): the consumer folds the partials itself --
    // reduce_parts, the very additions the fold launch would make -- and the launch (4.8 us plus a dependent dispatch) is saved
    if (ncols == 1 && !c->collectives && np <= kMaxParts) {
        *out = ColParts{parts, np, np, 1};
        return SLA_OK;
    }
    // the dots pass's partials on a single-rank context: arn_update_kernel folds its wavefronts' columns itself (sixteen loads per
    // lane, all in flight at once; arn_dots_grid sized the pass for it) -- the fold launch and its dependent dispatch are saved
    if (!c->collectives && (np & 63) == 0 && ((ncols + 3) / 4) * (np / 64) <= 16) {
        *out = ColParts{parts, np, np, 1};
        return SLA_OK;
    }
    // one tiny launch folds the per-workgroup partials to ncols values, so that the producers can use a
    // chip-filling grid without every consumer workgroup re-reducing ncols x grid partials
    double *loc = ws.gath + (size_t)(kMaxKrylov + 2) * c->nranks;
    SLA_TRY(launch_finalize_cols(c, parts, np, np, 1, ncols, loc));
    if (!c->collectives) {
        *out = ColParts{loc, 1, 1, 1};
        return SLA_OK;
    }
    SLA_TRY(dist_allgather_f64(c, loc, ws.gath, ncols));  // layout [rank][ncols]
    *out = ColParts{ws.gath, c->nranks, 1, ncols};
    return SLA_OK;
}

// arnoldi aa src kn (Sparse.hs:630-667) with src given as a raw local device pointer.
// Runs `kn` columns at most; Hhost (ld = ws.kn + 1) and *k_done are valid on return (synchronises).
int arn_run(ArnoldiWs &ws, sla_csr *A, const double *src_local, int kn, int *k_done) {
    sla_ctx *c = ws.c;
    const int64_t n = ws.n_local;
    const int ldh = ws.kn + 1;
    const int g = arn_grid(n);
    const bool orth = ws.halo == 0 && arn_orth_usable(c, n, ws.ld, kn);
    SLA_HIP_TRY(hipMemsetAsync(ws.H, 0, sizeof(double) * (size_t)ldh * (size_t)ws.kn, stream_of(c)));
    SLA_HIP_TRY(hipMemsetAsync(ws.d_sc, 0, sizeof(SolverScalars), stream_of(c)));
    ColParts cp;
    // q0 = normalize2 b
    SLA_TRY(launch_dot(c, n, src_local, src_local, ws.parts));
    SLA_TRY(arn_publish(ws, ws.parts, vec_grid(n), 1, &cp));
    SLA_TRY(launch_arn_normalize(c, n, Parts{cp.p, cp.np, cp.stride}, src_local, ws.Q, nullptr, ws.d_sc, 1));
    for (int i = 0; i < kn; ++i) {
        const double *qi = ws.Q + (size_t)i * ws.ld;
        SpmvLaunch l;  // aqi = aa #> qi
        l.y = ws.w;
        l.sc = ws.d_sc;
        if (ws.halo > 0) {   // the column as a vector view with its own halo slack: in-place exchange, interior rows overlapped
            sla_vec qv;
            qv.ctx = c;
            qv.n = ws.n;
            qv.n_local = ws.n_local;
            qv.shard = ws.shard;
            qv.begin = ws.begin;
            qv.d = const_cast<double *>(qi);
            SLA_TRY(spmv_exchanged(A, &qv, l, nullptr));
        } else {
            SLA_TRY(gather_raw(c, A, qi, (ws.n + c->nranks - 1) / c->nranks, &l.x));
            SLA_TRY(launch_spmv(A, l));
        }
        if (orth) {   // hhcoli, qipnn, qip, h_{i+1,i} and the breakdown test in ONE persistent launch (sla_arnoldi_orth.hip)
            SLA_TRY(launch_arn_orth(c, n, ws.Q, ws.ld, i + 1, ws.w, ws.Q + (size_t)(i + 1) * ws.ld, ws.H + (size_t)i * ldh, ws.H + (size_t)i * ldh + i + 1,
                                    ws.d_sc, ws.parts, ws.bar, i == 0 ? 1 : 0));
            continue;
        }
        // hhcoli = fmap (`dot` aqi) qv
        SLA_TRY(launch_arn_dots(c, n, ws.Q, ws.ld, i + 1, ws.w, ws.parts, ws.d_sc));
        SLA_TRY(arn_publish(ws, ws.parts, arn_dots_grid(n, i + 1), i + 1, &cp));
        // qipnn = aqi ^-^ sum_k h_k q_k ; partial ||qipnn||^2 ; H[0..i, i]
        double *pn = ws.parts + (size_t)kMaxKrylov * kArnGridMax;
        SLA_TRY(launch_arn_update(c, n, ws.Q, ws.ld, i + 1, cp.p, cp.np, cp.cs, cp.stride, ws.w, pn,
                                  ws.H + (size_t)i * ldh, ws.d_sc));
        ColParts cn;
        SLA_TRY(arn_publish(ws, pn, g, 1, &cn));
        // qip = normalize2 qipnn ; H[i+1, i] = norm2' qipnn ; breakdown test (not in arnInit)
        SLA_TRY(launch_arn_normalize(c, n, Parts{cn.p, cn.np, cn.stride}, ws.w, ws.Q + (size_t)(i + 1) * ws.ld,
                                     ws.H + (size_t)i * ldh + i + 1, ws.d_sc, i == 0 ? 1 : 0));
    }
    SLA_HIP_TRY(hipMemcpyAsync(ws.Hhost.data(), ws.H, sizeof(double) * (size_t)ldh * (size_t)ws.kn, hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipMemcpyAsync(ws.h_sc, ws.d_sc, sizeof(SolverScalars), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    if (orth && (ws.h_sc->flags & SLA_FLAG_SYNC_TIMEOUT)) {
        // a fused step gave up waiting for its other workgroups (CUs held by another job): its counters are in an unknown state and the basis is
        // incomplete -- the launch flow from here on for this context, the counters cleared, the whole run repeated (src is untouched)
        c->arn_orth = 0;
        c->arn_orth_fallbacks += 1;
        SLA_HIP_TRY(hipMemsetAsync(ws.bar, 0, arn_orth_bar_bytes(), stream_of(c)));
        return arn_run(ws, A, src_local, kn, k_done);
    }
    *k_done = ws.h_sc->kdone;
    return SLA_OK;
}

// min_y || beta e1 - H y ||_2 for the (k+1) x k Hessenberg H (column-major, ld = ldh): Givens QR +
// back substitution (the reference's commented gmres does qr + triUpperSolve, Sparse.hs:837-848)
void hessenberg_lsq(int k, int ldh, const double *H, double beta, double *y) {
    std::vector<double> R((size_t)(k + 1) * (size_t)k), g((size_t)k + 1, 0.0);
    for (int j = 0; j < k; ++j)
        for (int i = 0; i <= k; ++i) R[(size_t)j * (k + 1) + i] = H[(size_t)j * ldh + i];
    g[0] = beta;
    for (int j = 0; j < k; ++j) {
        const double a = R[(size_t)j * (k + 1) + j], b = R[(size_t)j * (k + 1) + j + 1];
        const double d = hypot(a, b);
        double cs = 1.0, sn = 0.0;
        if (d != 0.0) { cs = a / d; sn = b / d; }
        for (int l = j; l < k; ++l) {
            const double u = R[(size_t)l * (k + 1) + j], v = R[(size_t)l * (k + 1) + j + 1];
            R[(size_t)l * (k + 1) + j] = cs * u + sn * v;
            R[(size_t)l * (k + 1) + j + 1] = -sn * u + cs * v;
        }
        const double gu = g[(size_t)j], gv = g[(size_t)j + 1];
        g[(size_t)j] = cs * gu + sn * gv;
        g[(size_t)j + 1] = -sn * gu + cs * gv;
    }
    for (int i = k - 1; i >= 0; --i) {
        double acc = g[(size_t)i];
        for (int l = i + 1; l < k; ++l) acc -= R[(size_t)l * (k + 1) + i] * y[l];
        y[i] = acc / R[(size_t)i * (k + 1) + i];
    }
}

}  // namespace

extern "C" {

int sla_solver_init(int method, sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out) {
    if (A && !A->kids.empty()) return m_solver_init(method, A, b, x0, out);
    return no_throw("sla_solver_init", [&]() -> int {
        return solver_init_common(method, A, b, x0, 1e-6, 1e-4, out);
    });
}

int sla_solver_step(sla_solver_t S, int k_steps) {
    if (S && !S->kids.empty()) return k_steps >= 0 ? m_solver_step(S, k_steps) : fail(SLA_ERR_INVALID, "sla_solver_step: bad argument");
    if (!S || k_steps < 0) return fail(SLA_ERR_INVALID, "sla_solver_step: bad argument");
    sla_ctx *c = S->ctx;
    Bind bind(c);
    int k = 0;
    // Round 6: constant-coefficient stencil / banded matrices whose whole solver state fits the chip's registers + LDS run the k steps as ONE
    // persistent launch (sla_onchip.hip): same formulas, two counter barriers per step instead of three launches.
    if (k_steps > 0 && c->onchip != 0) {
        if (onchip_usable(S)) {
            SLA_TRY(launch_onchip_steps(S, ctl_of(S).step_index & 1, k_steps));
            ctl_of(S).step_index += k_steps;
            return SLA_OK;
        }
        if (c->onchip == 2 && (S->method == SLA_BICGSTAB_ || S->method == SLA_CGS_))
            return fail(SLA_ERR_INVALID, "sla_solver_step: onchip = 2 but this state record cannot run on chip (" + c->onchip_note + ")");
    }
    // Launch-bound sizes ("Launches and HIP graphs"): below ~2 M rows the five dependent launches of a step
    // cost about as much as its kernels.  Two consecutive steps (both parities of the double-buffered rho) are captured ONCE
    // into a HIP graph and replayed: the same kernels with the same arguments in the same order -- bit-identical iterates --
    // at the dependent-node latency of a graph instead of five stream dispatches per step.
    const bool graph_ok = c->step_graph != 0 && !S->step_graph_failed && !c->collectives && c->prof_kernel == -2 && !S->have_res &&
                          S->method != SLA_CGNE_ && (c->step_graph > 0 || S->A->rows <= c->step_graph_max_rows) && k_steps >= 4;
    if (graph_ok) {
        if (ctl_of(S).step_index & 1) {   // the captured pair starts at even parity
            SLA_TRY(enqueue_step(S, false, false));
            ++k;
        }
        if (S->step_graph && S->step_graph_gen != c->opt_gen) {   // options changed since the capture: the stream launches below would run another flow
            (void)hipGraphExecDestroy(S->step_graph);
            S->step_graph = nullptr;
        }
        if (!S->step_graph) {
            S->step_graph_gen = c->opt_gen;
            // A failed capture or instantiation is not an error of the step: the graph is an optimisation.  Restore the step
            // bookkeeping exactly (enqueue_step may have advanced it by 0, 1 or 2), never try again on this state record and
            // fall through to the plain stream launches below.
            hipGraph_t g = nullptr;
            const int index0 = ctl_of(S).step_index;
            if (S->A->canon_lazy && !spmv_value_indexed(S->A, false)) SLA_TRY(csr_ensure_canon(S->A));   // (an allocation: not inside the capture)
            hipError_t e = hipStreamBeginCapture(stream_of(c), hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rc = enqueue_step(S, false, false);
                if (rc == SLA_OK) rc = enqueue_step(S, false, false);
                e = hipStreamEndCapture(stream_of(c), &g);
                if (rc != SLA_OK && e == hipSuccess) e = hipErrorUnknown;
            }
            ctl_of(S).step_index = index0;   // (captured, not executed)
            S->have_res = false;
            if (e == hipSuccess) e = hipGraphInstantiate(&S->step_graph, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                S->step_graph = nullptr;
                S->step_graph_failed = true;
            }
        }
        for (; S->step_graph && k + 2 <= k_steps; k += 2) {
            SLA_HIP_TRY(hipGraphLaunch(S->step_graph, stream_of(c)));
            ctl_of(S).step_index += 2;
        }
    }
    for (; k < k_steps; ++k) SLA_TRY(enqueue_step(S, false, false));
    return SLA_OK;
}

int sla_solver_get(sla_solver_t S, int field, sla_vec_t out) {
    if (S && !S->kids.empty()) return m_solver_get(S, field, out);
    if (!S || !out) return fail(SLA_ERR_INVALID, "null argument");
    sla_vec *src = nullptr;
    switch (field) {
        case SLA_STATE_X: src = S->x; break;
        case SLA_STATE_R: src = S->r; break;
        case SLA_STATE_P: src = S->p; break;
        case SLA_STATE_U: src = S->method == SLA_BCG_ ? nullptr : S->u; break;
        case SLA_STATE_RHAT: src = S->method == SLA_BCG_ ? S->r0hat : nullptr; break;
        case SLA_STATE_PHAT: src = S->method == SLA_BCG_ ? S->u : nullptr; break;
    }
    if (!src) return fail(SLA_ERR_INVALID, "sla_solver_get: this method has no such state field");
    if (!out->kids.empty() || out->ctx != S->ctx) return mixed_handles("sla_solver_get");
    return sla_vec_copy(src, out);
}

// A deep copy of a state record: the reference's step functions are PURE (`bicgstabStep aa r0hat s` returns a new record and
// leaves s alone, Sparse.hs:972-981), so `iterate (bicgstabStep aa r0hat) s0 !! k` or keeping s_j around while stepping on
// needs states that do not alias.  Everything the step kernels read is copied on the context stream: the state vectors,
// the partial sums the next kernel's prologue reduces, the per-rank tables, the device scalars, the step bookkeeping.
int sla_solver_clone(sla_solver_t S, sla_solver_t *out) {
    if (S && !S->kids.empty()) return out ? m_solver_clone(S, out) : fail(SLA_ERR_INVALID, "null argument");
    return no_throw("sla_solver_clone", [&]() -> int {
        if (!S || !out) return fail(SLA_ERR_INVALID, "sla_solver_clone: null argument");
        sla_ctx *c = S->ctx;
        Bind bind(c);
        sla_solver *T = nullptr;
        SLA_TRY(solver_alloc(S->A, S->method, &T));
        int rc = SLA_OK;
        sla_vec *src[] = {S->x, S->r, S->p, S->u, S->r0hat, S->b, S->t1, S->t2, S->t3};
        sla_vec *dst[] = {T->x, T->r, T->p, T->u, T->r0hat, T->b, T->t1, T->t2, T->t3};
        for (int i = 0; i < 9 && rc == SLA_OK; ++i)
            if (src[i] && dst[i]) {
                // (sla_vec_copy moves the own rows; the ghost-row flows also keep the neighbours' planes valid in the slack around
                // them, so copy the whole guarded allocation when the state runs that flow)
                if (S->ghost) {
                    const size_t g = c->vec_guard, bytes = sizeof(double) * (size_t)std::max<int64_t>(src[i]->shard, 1) + 2 * g;
                    if (hipMemcpyAsync((char *)dst[i]->d - g, (const char *)src[i]->d - g, bytes, hipMemcpyDeviceToDevice, stream_of(c)) != hipSuccess)
                        rc = fail(SLA_ERR_HIP, "sla_solver_clone: device copy failed");
                } else {
                    rc = sla_vec_copy(src[i], dst[i]);
                }
            }
        hipError_t e = hipSuccess;
        if (rc == SLA_OK) e = hipMemcpyAsync(T->d_parts, S->d_parts, sizeof(double) * P_SLOTS * kMaxParts, hipMemcpyDeviceToDevice, stream_of(c));
        if (rc == SLA_OK && e == hipSuccess)
            e = hipMemcpyAsync(T->d_gath, S->d_gath, sizeof(double) * ((size_t)P_SLOTS * 2 * c->nranks + 8), hipMemcpyDeviceToDevice, stream_of(c));
        if (rc == SLA_OK && e == hipSuccess) e = hipMemcpyAsync(T->d_sc, S->d_sc, sizeof(SolverScalars), hipMemcpyDeviceToDevice, stream_of(c));
        if (rc == SLA_OK && e != hipSuccess) rc = fail(SLA_ERR_HIP, std::string("sla_solver_clone: ") + hipGetErrorString(e));
        if (rc != SLA_OK) {
            sla_solver_destroy(T);
            return rc;
        }
        *T->h_sc = *S->h_sc;
        T->have_res = S->have_res;
        T->ghost = S->ghost;
        T->ghl = S->ghl;
        T->ghr = S->ghr;
        // the bookkeeping holds pointers into the source's partial / per-rank tables: rebase them onto the copy's
        new (&ctl_of(T)) StepCtl(ctl_of(S));
        auto rebase = [&](Parts &p) {
            if (!p.p) return;
            if (p.p >= S->d_parts && p.p < S->d_parts + (size_t)P_SLOTS * kMaxParts) p.p = T->d_parts + (p.p - S->d_parts);
            else p.p = T->d_gath + (p.p - S->d_gath);
        };
        rebase(ctl_of(T).res);
        rebase(ctl_of(T).pp);
        *out = T;
        return SLA_OK;
    });
}

// Replace the shadow residual r0hat of a CGS / BiCGSTAB state (the explicit `r0hat` / `rhat` argument of bicgstabStep /
// cgsStep, Sparse.hs:928, :972; sla_solver_init stores r0 = b - A x0 there, the README's choice) and re-evaluate the
// carried rho = r . r0hat with it -- the reference recomputes `r <.> r0hat` at the top of every step.
int sla_solver_set_shadow(sla_solver_t S, sla_vec_t r0hat) {
    if (S && !S->kids.empty()) return m_solver_set_shadow(S, r0hat);
    return no_throw("sla_solver_set_shadow", [&]() -> int {
        if (!S || !r0hat) return fail(SLA_ERR_INVALID, "sla_solver_set_shadow: null argument");
        if (S->method != SLA_BICGSTAB_ && S->method != SLA_CGS_) return fail(SLA_ERR_INVALID, "sla_solver_set_shadow: CGS / BiCGSTAB states only");
        if (r0hat->n != S->r0hat->n) return fail(SLA_ERR_DIM_MISMATCH, "sla_solver_set_shadow: dimension mismatch");
        if (S->ghost) return fail(SLA_ERR_INVALID, "sla_solver_set_shadow: not available on the ghost-row sharded flow");
        sla_ctx *c = S->ctx;
        if (!r0hat->kids.empty() || r0hat->ctx != c) return mixed_handles("sla_solver_set_shadow");
        Bind bind(c);
        SLA_TRY(sla_vec_copy(r0hat, S->r0hat));
        Parts rho;
        SLA_TRY(launch_dot(c, S->r->n_local, S->r->d, S->r0hat->d, slot(S, P_TMP)));
        SLA_TRY(publish(S, P_TMP, -1, vec_grid(S->r->n_local), &rho, nullptr));
        return launch_set_rho(c, S->d_sc, rho, ctl_of(S).step_index & 1);
    });
}

int sla_solver_destroy(sla_solver_t S) {
    if (S && !S->kids.empty()) return m_solver_destroy(S);
    if (!S) return SLA_OK;
    Bind bind(S->ctx);
    if (S->ctx && S->ctx->stream) (void)hipStreamSynchronize(S->ctx->stream);
    sla_vec *vs[] = {S->x, S->r, S->p, S->u, S->r0hat, S->b, S->t1, S->t2, S->t3};
    for (sla_vec *v : vs) sla_vec_destroy(v);
    if (S->step_graph) (void)hipGraphExecDestroy(S->step_graph);
    if (S->d_parts) (void)hipFree(S->d_parts);
    if (S->d_gath) (void)hipFree(S->d_gath);
    if (S->d_sc) (void)hipFree(S->d_sc);
    if (S->d_hist) (void)hipFree(S->d_hist);
    if (S->h_sc) (void)hipHostFree(S->h_sc);
    delete S;
    return SLA_OK;
}

int sla_bicgstab_init(sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out) { return sla_solver_init(SLA_BICGSTAB_, A, b, x0, out); }
int sla_bicgstab_step(sla_solver_t S, int k) {
    if (!S || S->method != SLA_BICGSTAB_) return fail(SLA_ERR_INVALID, "not a BiCGSTAB state");
    return sla_solver_step(S, k);
}
int sla_cgs_init(sla_csr_t A, sla_vec_t b, sla_vec_t x0, sla_solver_t *out) { return sla_solver_init(SLA_CGS_, A, b, x0, out); }
int sla_cgs_step(sla_solver_t S, int k) {
    if (!S || S->method != SLA_CGS_) return fail(SLA_ERR_INVALID, "not a CGS state");
    return sla_solver_step(S, k);
}

int sla_linsolve0(int method, sla_csr_t A, sla_vec_t b, sla_vec_t x0, const sla_solve_opts *opts, sla_vec_t x_out,
                  sla_solve_info *user_info) {
    if (A && !A->kids.empty()) return m_linsolve0(method, A, b, x0, opts, x_out, user_info);
    return no_throw("sla_linsolve0", [&]() -> int {
        if (!A || !b || !x0 || !x_out) return fail(SLA_ERR_INVALID, "sla_linsolve0: null argument");
        sla_solve_opts o;
        SLA_TRY(read_solve_opts(opts, &o, "sla_linsolve0"));
        if (o.max_iters <= 0) o.max_iters = 200;
        if (o.check_every <= 0) o.check_every = 16;
        sla_solve_info local, *info = &local;   // (committed to the caller's struct -- the members it has -- on every exit path)
        SLA_TRY(info_begin(user_info, &local, "sla_linsolve0"));
        struct Commit { sla_solve_info *u; const sla_solve_info &l; ~Commit() { info_commit(u, l); } } commit{user_info, local};
        const int hist_cap = (o.history && o.history_cap > 0 && o.true_residual) ? std::min(o.history_cap, o.max_iters) : 0;
        // | m /= nb = throwM (MatVecSizeMismatchException "linSolve0" dm nb)      (Sparse.hs:1022)
        if (A->m != b->n) return fail(SLA_ERR_DIM_MISMATCH, "linSolve0 : matrix rows and rhs dimension differ");
        if (x_out->n != A->n) return fail(SLA_ERR_DIM_MISMATCH, "linSolve0 : output vector has the wrong dimension");
        sla_ctx *c = A->ctx;
        if (!b->kids.empty() || !x0->kids.empty() || !x_out->kids.empty() || b->ctx != c || x0->ctx != c || x_out->ctx != c) return mixed_handles("sla_linsolve0");
        Bind bind(c);
        // solve aa' b' | isDiagonalSM aa' = return $ reciprocal aa' #> b'           (Sparse.hs:1024-1025)
        if (A->is_diagonal) {
            SLA_TRY(csr_ensure_canon(A));
            SLA_TRY(launch_diag_solve(c, b->n_local, A->d_val, b->d, x_out->d));
            SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
            if (info) info->flags = SLA_FLAG_DIAGONAL;
            return SLA_OK;
        }
        if (method != SLA_BICGSTAB_ && method != SLA_CGS_ && method != SLA_CGNE_)
            return fail(SLA_ERR_UNSUPPORTED_METHOD, "linSolve0 : Only BICGSTAB_, CGS_, and CGNE_ are implemented");  // :1031
        sla_solver *S = nullptr;
        SLA_TRY(solver_init_common(method, A, b, x0, o.tol_abs, o.tol_rel, &S, hist_cap));
        int rc = SLA_OK, total = 0;
        // Round 6: on matrices with an on-chip plan the whole loop -- step, true residual, test (runIter, :1043-1052) -- is ONE persistent
        // launch that stops at the first iterate with resnorm <= tol or after max_iters steps (sla_onchip.hip, RES instantiations); the
        // device keeps iters, resnorm, the flags and the residual trace exactly as the launch flow's check does.
        if (o.true_residual && o.max_iters > 0 && c->onchip != 0 && onchip_usable(S, true)) {
            rc = launch_onchip_steps(S, ctl_of(S).step_index & 1, o.max_iters, true);
            if (rc == SLA_OK) rc = read_scalars(S);
            if (rc == SLA_OK && (S->h_sc->flags & SLA_FLAG_SYNC_TIMEOUT)) {
                // The launch lost its co-residency (a workgroup never arrived: another job on the device holding a CU): its state is not to be
                // trusted.  linSolve0 owns this record -- build it again from b and x0 and let the launch flow below do the solve.
                c->onchip_fallbacks += 1;
                sla_solver_destroy(S);
                S = nullptr;
                SLA_TRY(solver_init_common(method, A, b, x0, o.tol_abs, o.tol_rel, &S, hist_cap));
            } else if (rc == SLA_OK) {
                ctl_of(S).step_index += S->h_sc->iters;
                total = o.max_iters;   // (done, or max_iters steps taken: either way the loop below has nothing left to do)
            }
        }
        while (rc == SLA_OK && total < o.max_iters) {  // runIter n state | n >= nits = return x        (:1045)
            const int k = std::min(o.check_every, o.max_iters - total);
            const bool dual = o.true_residual != 0 && dual_ok(S);
            for (int j = 0; j < k && rc == SLA_OK; ++j) {
                if (!o.true_residual) rc = enqueue_step(S, false, false);
                else if (!dual) rc = enqueue_step(S, true, false);
                else rc = enqueue_step(S, /*res_after=*/j == k - 1, /*dual_prev=*/j > 0);
            }
            if (rc != SLA_OK) break;
            total += k;
            if (o.true_residual) {
                if ((rc = launch_check(c, S->d_sc, ctl_of(S).res)) != SLA_OK) break;
                if ((rc = read_scalars(S)) != SLA_OK) break;
                if (S->h_sc->done) break;
            }
        }
        if (rc == SLA_OK && !o.true_residual) {  // extension mode: report the final true residual once
            Parts res;
            if ((rc = enqueue_residual(S, &res)) == SLA_OK && (rc = launch_check(c, S->d_sc, res)) == SLA_OK) rc = read_scalars(S);
        }
        if (rc == SLA_OK) rc = sla_vec_copy(S->x, x_out);
        if (rc == SLA_OK) {
            hipError_t e = hipStreamSynchronize(stream_of(c));
            if (e != hipSuccess) rc = fail(SLA_ERR_HIP, hipGetErrorString(e));
        }
        if (rc == SLA_OK) fill_info(S, info, true);
        if (rc == SLA_OK && S->hist_cap > 0) {   // the trace: one true residual norm per iteration taken (cgsStepDebug's output, Sparse.hs:942-948)
            const int len = std::min<int>(S->h_sc->iters, S->hist_cap);
            if (len > 0 && hipMemcpy(o.history, S->d_hist, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(SLA_ERR_HIP, "sla_linsolve0: residual trace download failed");
            else if (info) info->history_len = len;
        }
        sla_solver_destroy(S);
        return rc;
    });
}

int sla_arnoldi(sla_csr_t A, sla_vec_t b, int kn, double *Q_colmajor, double *H_colmajor, int *k_done) {
    if (A && !A->kids.empty()) return (H_colmajor && k_done) ? m_arnoldi(A, b, kn, Q_colmajor, H_colmajor, k_done) : fail(SLA_ERR_INVALID, "sla_arnoldi: null argument");
    return no_throw("sla_arnoldi", [&]() -> int {
        if (!A || !b || !H_colmajor || !k_done) return fail(SLA_ERR_INVALID, "sla_arnoldi: null argument");
        // | otherwise = throwM (MatVecSizeMismatchException "arnoldi" (m,n) nb)      (Sparse.hs:637)
        if (A->n != b->n) return fail(SLA_ERR_DIM_MISMATCH, "arnoldi : matrix columns and vector dimension differ");
        if (A->m != A->n) return fail(SLA_ERR_DIM_MISMATCH, "arnoldi : matrix must be square");
        if (kn < 1 || kn + 1 > kMaxKrylov) return fail(SLA_ERR_INVALID, "sla_arnoldi: kn must be in [1, 63]");
        sla_ctx *c = A->ctx;
        if (!b->kids.empty() || b->ctx != c) return mixed_handles("sla_arnoldi");
        Bind bind(c);
        ArnoldiWs ws;
        SLA_TRY(arn_alloc_agreed(ws, A, b, kn));
        int k = 0;
        SLA_TRY(arn_run(ws, A, b->d, kn, &k));
        memcpy(H_colmajor, ws.Hhost.data(), sizeof(double) * (size_t)(kn + 1) * (size_t)kn);
        *k_done = k;
        if (Q_colmajor && b->n_local > 0) {
            // this rank's rows of the k+1 basis vectors, leading dimension n_local
            SLA_HIP_TRY(hipMemcpy2D(Q_colmajor, sizeof(double) * (size_t)b->n_local, ws.Q, sizeof(double) * (size_t)ws.ld,
                                    sizeof(double) * (size_t)b->n_local, (size_t)(k + 1), hipMemcpyDeviceToHost));
        }
        return SLA_OK;
    });
}

int sla_gmres(sla_csr_t A, sla_vec_t b, sla_vec_t x0, int restart, const sla_solve_opts *opts, sla_vec_t x_out,
              sla_solve_info *user_info) {
    if (A && !A->kids.empty()) return m_gmres(A, b, x0, restart, opts, x_out, user_info);
    return no_throw("sla_gmres", [&]() -> int {
        if (!A || !b || !x0 || !x_out) return fail(SLA_ERR_INVALID, "sla_gmres: null argument");
        sla_solve_opts o;
        SLA_TRY(read_solve_opts(opts, &o, "sla_gmres"));
        if (o.max_iters <= 0) o.max_iters = 200;
        sla_solve_info local, *info = &local;
        SLA_TRY(info_begin(user_info, &local, "sla_gmres"));
        struct Commit { sla_solve_info *u; const sla_solve_info &l; ~Commit() { info_commit(u, l); } } commit{user_info, local};
        if (A->m != b->n) return fail(SLA_ERR_DIM_MISMATCH, "gmres : matrix rows and rhs dimension differ");
        if (A->m != A->n || A->n != x0->n || x_out->n != A->n) return fail(SLA_ERR_DIM_MISMATCH, "gmres : mismatched dimensions");
        if (restart < 1) return fail(SLA_ERR_INVALID, "sla_gmres: restart must be >= 1");
        restart = std::min<int64_t>({(int64_t)restart, (int64_t)kMaxKrylov - 1, std::max<int64_t>(A->n, 1)});
        sla_ctx *c = A->ctx;
        if (!b->kids.empty() || !x0->kids.empty() || !x_out->kids.empty() || b->ctx != c || x0->ctx != c || x_out->ctx != c) return mixed_handles("sla_gmres");
        Bind bind(c);
        ArnoldiWs ws;
        sla_vec *x = nullptr, *r = nullptr;
        int rc;
        {   // all local allocations first, then one agreement (a rank failing here must not leave its peers in a collective)
            rc = arn_alloc(ws, A, b, restart);
            if (rc == SLA_OK) rc = vec_alloc(c, A->n, &x);
            if (rc == SLA_OK) rc = vec_alloc(c, A->n, &r);
            const std::string msg = rc != SLA_OK ? sla_last_error() : "";
            int bad = rc != SLA_OK ? 1 : 0;
            if (c->collectives) {
                const int rca = dist_allreduce_max_i32(c, &bad);
                if (rc == SLA_OK) rc = rca;
            }
            if (rc == SLA_OK && bad) rc = fail(SLA_ERR_ALLOC, "GMRES workspace allocation failed on another rank of the row-sharded job");
            else if (rc != SLA_OK && !msg.empty()) set_error(msg);
            if (rc != SLA_OK) {
                sla_vec_destroy(x);
                sla_vec_destroy(r);
                return rc;
            }
        }
        double tol = 0.0, beta = NAN, r0norm = NAN;
        int total = 0, flags = 0;
        bool first = true;
        std::vector<double> y((size_t)restart + 1);
        if (rc == SLA_OK) rc = sla_vec_copy(x0, x);
        while (rc == SLA_OK) {
            SpmvLaunch l;  // r = b ^-^ (aa #> x)
            l.epi = EPI_SUB;
            l.y = r->d;
            l.w = b->d;
            if ((rc = spmv_exchanged(A, x, l, nullptr)) != SLA_OK) break;
            double ss = 0.0;
            if ((rc = launch_dot(c, r->n_local, r->d, r->d, c->d_parts)) != SLA_OK) break;
            if ((rc = reduce_to_host(c, c->d_parts, nullptr, vec_grid(r->n_local), &ss)) != SLA_OK) break;
            beta = sqrt(ss);
            if (first) { r0norm = beta; tol = fmax(o.tol_abs, o.tol_rel * beta); first = false; }
            if (beta <= tol) { flags |= SLA_FLAG_CONVERGED; break; }
            if (!(beta == beta) || isinf(beta)) { flags |= SLA_FLAG_NONFINITE; break; }
            if (total >= o.max_iters) { flags |= SLA_FLAG_MAX_ITERS; break; }
            const int mc = std::min(restart, o.max_iters - total);
            int k = 0;
            if ((rc = arn_run(ws, A, r->d, mc, &k)) != SLA_OK) break;
            if (ws.h_sc->flags & SLA_FLAG_BREAKDOWN) flags |= SLA_FLAG_BREAKDOWN;
            hessenberg_lsq(k, ws.kn + 1, ws.Hhost.data(), beta, y.data());
            hipError_t e = hipMemcpyAsync(ws.ycoef, y.data(), sizeof(double) * (size_t)k, hipMemcpyHostToDevice, stream_of(c));
            if (e != hipSuccess) { rc = fail(SLA_ERR_HIP, hipGetErrorString(e)); break; }
            if ((rc = launch_gemv_accum(c, x->n_local, ws.Q, ws.ld, k, ws.ycoef, x->d)) != SLA_OK) break;
            e = hipStreamSynchronize(stream_of(c));  // y is host memory reused next cycle
            if (e != hipSuccess) { rc = fail(SLA_ERR_HIP, hipGetErrorString(e)); break; }
            total += k;
        }
        if (rc == SLA_OK) rc = sla_vec_copy(x, x_out);
        if (rc == SLA_OK) {
            hipError_t e = hipStreamSynchronize(stream_of(c));
            if (e != hipSuccess) rc = fail(SLA_ERR_HIP, hipGetErrorString(e));
        }
        if (rc == SLA_OK && info) {
            info->iters = total;
            info->flags = flags | (csr_fold_relaxed(A) ? SLA_FLAG_RELAXED_ORDER : 0);
            info->resnorm = beta;
            info->r0norm = r0norm;
            info->tol = tol;
        }
        sla_vec_destroy(x);
        sla_vec_destroy(r);
        return rc;
    });
}

// instance LinearSystem (SpVector Double): aa <\> b = linSolve0 GMRES_ aa b (mkSpVR n $ replicate n 0.1)
// (dead code in the reference, Sparse.hs:1080-1084)
int sla_linsolve(sla_csr_t A, sla_vec_t b, sla_vec_t x_out, sla_solve_info *info) {
    if (A && !A->kids.empty()) return m_linsolve(A, b, x_out, info);
    return no_throw("sla_linsolve", [&]() -> int {
        if (!A || !b || !x_out) return fail(SLA_ERR_INVALID, "sla_linsolve: null argument");
        Bind bind(A->ctx);
        sla_vec *x0 = nullptr;
        SLA_TRY(vec_alloc(A->ctx, A->n, &x0));
        int rc = launch_fill(A->ctx, x0->n_local, 0.1, x0->d);
        if (rc == SLA_OK) rc = sla_gmres(A, b, x0, 30, nullptr, x_out, info);
        sla_vec_destroy(x0);
        return rc;
    });
}

}  // extern "C"
