// sla_vec_kernels.hip -- the streaming BLAS-1 kernels and the vector kernels of the solver steps (16 bytes per lane, grid-stride):
//   (<.>) / norm2 (Data/Sparse/SpVector.hs:116-129), (^+^) (^-^) (.*) (:107-114): dot, axpby, scal, fill, the partial-sum folds;
//   bicgstabStep (Numeric/LinearAlgebra/Sparse.hs:972-981): K2, K4, K5 and the fused K4+K5 sweep (K1 / K3 are SpMV epilogues);
//   cgsStep (:928-939): C2, C4;  cgneStep (:870-878): N2, N3b;  linSolve0's residual test and scalar set-up (:1032-1052).
// Every reduction is two-stage and deterministic: producers write one partial per workgroup, the first consumer re-reduces them in
// a fixed order in its prologue and workgroup 0 publishes the scalar (SolverScalars).  No atomics, no host round trip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

// per-stream cache policy of the two BiCGSTAB sweeps at sizes that overflow the memory-side cache (ctx option vec_policy; bit set = the
// stream goes past the caches).  K2: 0 r, 1 Ap, 2 s (store).  K4+K5: 3 s, 4 As, 5 Ap, 6 p, 7 x, 8 x (store), 9 r (store), 10 p (store).
// CGS stores: C2 11 q, 12 uq; C4 13 u, 14 p (their loads and C2's x store always go past the caches).
__device__ __forceinline__ double2 ldpol(const double *p, int64_t i2, int pol, int bit) { return (pol >> bit) & 1 ? ld2_nt(p, i2) : ld2_t(p, i2); }
__device__ __forceinline__ void stpol(double *p, int64_t i2, double2 v, int pol, int bit) {
    if ((pol >> bit) & 1) st2_nt(p, i2, v);
    else st2_t(p, i2, v);
}

// ---------------------------------------------------------------------------------------------
// streaming BLAS-1 kernels: 16 bytes per lane (double2), grid-stride, <= kVecGridMax workgroups
// ---------------------------------------------------------------------------------------------
int vec_grid(int64_t n_local) {
    int64_t g = (n_local / 2 + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kVecGridMax) g = kVecGridMax;
    return (int)g;
}

__global__ void __launch_bounds__(kBlock) dot_kernel(int64_t n, const double *x, const double *y, double *p1) {
    __shared__ double s_red[4];
    double acc = 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 a = ld2(x, i2), b = ld2(y, i2);
        acc += a.x * b.x;
        acc += a.y * b.y;
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) acc += x[n - 1] * y[n - 1];
    const double s = block_sum(acc, s_red);
    if (threadIdx.x == 0) p1[blockIdx.x] = s;
}

// out[j] = sum_i parts[j * cs + i * stride], j < ncols; one workgroup per column (grid-stride over columns)
__global__ void __launch_bounds__(kBlock) finalize_kernel(const double *parts, int np, int cs, int stride, int ncols,
                                                           double *out) {
    __shared__ double s_red[4];
    for (int j = blockIdx.x; j < ncols; j += gridDim.x) {
        const double s = reduce_parts(parts + (int64_t)j * cs, np, stride, s_red);
        if (threadIdx.x == 0) out[j] = s;
    }
}

__global__ void __launch_bounds__(kBlock) axpby_kernel(int64_t n, double a, const double *x, double b, double *y) {
    SLA_VEC_LOOP_BEGIN(n)
        const double2 u = ld2(x, i2);
        double2 v = ld2(y, i2);
        v.x = a * u.x + b * v.x;
        v.y = a * u.y + b * v.y;
        st2(y, i2, v);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) y[n - 1] = a * x[n - 1] + b * y[n - 1];
}

__global__ void __launch_bounds__(kBlock) scal_kernel(int64_t n, double a, double *x) {
    SLA_VEC_LOOP_BEGIN(n)
        double2 v = ld2(x, i2);
        v.x *= a;
        v.y *= a;
        st2(x, i2, v);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) x[n - 1] *= a;
}

__global__ void __launch_bounds__(kBlock) fill_kernel(int64_t n, double a, double *x) {
    SLA_VEC_LOOP_BEGIN(n)
        st2(x, i2, make_double2(a, a));
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) x[n - 1] = a;
}

int launch_dot(sla_ctx *c, int64_t n, const double *x, const double *y, double *p1) {
    hipLaunchKernelGGL(dot_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, x, y, p1);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
// out[0] = sum p1, out[1] = sum p2 (0 when p2 is null): two workgroups, one launch
__global__ void __launch_bounds__(kBlock) finalize2_kernel(const double *p1, const double *p2, int np, double *out) {
    __shared__ double s_red[4];
    const double *p = blockIdx.x == 0 ? p1 : p2;
    const double s = p ? reduce_parts(p, np, 1, s_red) : 0.0;
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
int launch_finalize(sla_ctx *c, const double *p1, const double *p2, int np, double *out) {
    hipLaunchKernelGGL(finalize2_kernel, dim3(2), dim3(kBlock), 0, stream_of(c), p1, p2, np, out);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_finalize_cols(sla_ctx *c, const double *parts, int np, int cs, int stride, int ncols, double *out) {
    hipLaunchKernelGGL(finalize_kernel, dim3(ncols > 0 ? ncols : 1), dim3(kBlock), 0, stream_of(c), parts, np, cs, stride, ncols, out);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_axpby(sla_ctx *c, int64_t n, double a, const double *x, double b, double *y) {
    hipLaunchKernelGGL(axpby_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, a, x, b, y);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_scal(sla_ctx *c, int64_t n, double a, double *x) {
    hipLaunchKernelGGL(scal_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, a, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_fill(sla_ctx *c, int64_t n, double a, double *x) {
    hipLaunchKernelGGL(fill_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, a, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// BiCGSTAB (Sparse.hs:972-981): K2 / K4 / K5 (K1, K3 are SpMV epilogues)
// ---------------------------------------------------------------------------------------------
// K2: alphaj = (r <.> r0hat) / (aap <.> r0hat) ; sj = r ^-^ (alphaj .* aap)
template <bool NT>
__global__ void __launch_bounds__(kBlock) bicg_k2_kernel(int64_t n, SolverScalars *sc, Parts apr, int par,
                                                          Parts res, int count_iter, const double *r,
                                                          const double *ap, double *s, int pol) {
    __shared__ double s_red[4];
    // everything the head of the kernel needs is issued before any of it is waited for: the scalars, the partials of
    // Ap . r0hat and the first element pair of the sweep (one round trip instead of three in a row, with HBM already streaming)
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kBlock, i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int done = sc->done;
    const double rho = sc->rho2[par];
    double pv[8];
    parts_issue(apr.p, apr.n, apr.stride, pv);
    const int64_t i0c = n2 > 0 ? min(i0, n2 - 1) : 0;   // (clamped: the loads are unconditional)
    double2 a = make_double2(0.0, 0.0), b = a;
    if (n2 > 0) { a = ldpol(r, i0c, pol, 0); b = ldpol(ap, i0c, pol, 1); }
    if (done) return;
    // dual-SpMV flow: K1 of THIS step also evaluated the previous step's true residual; test it here
    if (res.p && residual_converged(sc, res.p, res.n, res.stride, s_red)) return;
    if (count_iter && blockIdx.x == 0 && threadIdx.x == 0) sc->iters += 1;
    const double alpha = rho / block_sum(parts_fold(pv, apr.n), s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
    for (int64_t i2 = i0; i2 < n2; i2 += gs) {
        if (i2 != i0) { a = ldpol(r, i2, pol, 0); b = ldpol(ap, i2, pol, 1); }
        stpol(s, i2, make_double2(a.x - alpha * b.x, a.y - alpha * b.y), pol, 2);
    }
    if (SLA_HAS_TAIL(n)) s[n - 1] = r[n - 1] - alpha * ap[n - 1];
}

// K4: omegaj = (aasj <.> sj) / (aasj <.> aasj) ; xj1 = x ^+^ alphaj .* p ^+^ omegaj .* sj ;
//     rj1 = sj ^-^ omegaj .* aasj ; partial rj1 <.> r0hat
template <bool NT>
__global__ void __launch_bounds__(kBlock) bicg_k4_kernel(int64_t n, SolverScalars *sc, Parts ass, Parts asas,
                                                          const double *p, const double *s, const double *as,
                                                          const double *r0hat, double *x, double *r, double *prho) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double num = reduce_parts(ass.p, ass.n, ass.stride, s_red);
    const double den = reduce_parts(asas.p, asas.n, asas.stride, s_red);
    const double omega = num / den, alpha = sc->alpha;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->omega = omega;
    double acc = 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 pv = ld2s<NT>(p, i2), sv = ld2s<NT>(s, i2), av = ld2s<NT>(as, i2), hv = ld2s<NT>(r0hat, i2);
        double2 xv = ld2s<NT>(x, i2);
        xv.x = (xv.x + alpha * pv.x) + omega * sv.x;
        xv.y = (xv.y + alpha * pv.y) + omega * sv.y;
        if (NT) st2_nt(x, i2, xv);  // nobody reads x before the next K4: do not let it push the live vectors out
        else st2(x, i2, xv);
        const double2 rv = make_double2(sv.x - omega * av.x, sv.y - omega * av.y);
        st2(r, i2, rv);
        acc += rv.x * hv.x;
        acc += rv.y * hv.y;
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        x[i] = (x[i] + alpha * p[i]) + omega * s[i];
        const double rv = s[i] - omega * as[i];
        r[i] = rv;
        acc += rv * r0hat[i];
    }
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) prho[blockIdx.x] = t;
}

// K4 + K5 in one sweep (single-rank contexts, SLA_BICG_FUSE45).  K5 needs beta = rho_{j+1} / rho_j * alpha / omega with
// rho_{j+1} = r_{j+1} . r0hat, a sum over ALL rows of the r_{j+1} that K4 is only just writing -- which is why the reference's step
// splits there.  By linearity r_{j+1} . r0hat = (s - omega As) . r0hat = s . r0hat - omega (As . r0hat), and both of those sums
// are available BEFORE the sweep when K3 (which streams s and As anyway) also reads r0hat: EPI_DOT4.  The update
// formulas of x, r and p are the reference's, term by term; only rho is evaluated through the identity (its rounding error is
// eps (|s| + |omega| |As|) . |r0hat| either way: the elementwise r_{j+1} = s - omega As carries the same cancellation).  Eight
// vector passes (p, s, As, x, Ap in; x, r, p out) instead of seven + four, and the r0hat pass moves into K3: 16 instead of 19
// passes per step.
// FS (round 5, K2 folded into K3 -- spmv_wdia_march_kernel<.., SF>): s was never stored.  It is rebuilt here as r - alpha Ap from the old r
// (read in place of s: the same five input streams) and Ap, with bicg_k2_kernel's fused multiply-add: the bits the SpMV gathered.
template <bool NT, bool FS>
__global__ void __launch_bounds__(kBlock) bicg_k45_kernel(int64_t n, SolverScalars *sc, Parts ass, Parts asas, Parts tr0, Parts sr0,
                                                           int par, const double *s, const double *as, const double *ap, double *x,
                                                           double *r, double *p, int pol) {
    if constexpr (FS) s = r;
    __shared__ double s_red[16];
    // the four sums, the scalars and the first element pairs of the five input vectors are issued together (see bicg_k2_kernel)
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kBlock, i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int done = sc->done;
    const double alpha = sc->alpha, rho = sc->rho2[par];
    double q0[8], q1[8], q2[8], q3[8];
    parts_issue(ass.p, ass.n, ass.stride, q0);
    parts_issue(asas.p, asas.n, asas.stride, q1);
    parts_issue(tr0.p, tr0.n, tr0.stride, q2);
    parts_issue(sr0.p, sr0.n, sr0.stride, q3);
    const int64_t i0c = n2 > 0 ? min(i0, n2 - 1) : 0;
    double2 sv = make_double2(0.0, 0.0), av = sv, vv = sv, pv = sv, xv = sv;
    if (n2 > 0) { sv = ldpol(s, i0c, pol, 3); av = ldpol(as, i0c, pol, 4); vv = ldpol(ap, i0c, pol, 5); pv = ldpol(p, i0c, pol, 6); xv = ldpol(x, i0c, pol, 7); }
    if (done) return;
    double sums[4] = {parts_fold(q0, ass.n), parts_fold(q1, asas.n), parts_fold(q2, tr0.n), parts_fold(q3, sr0.n)};
    block_sum_multi<4>(sums, s_red);
    const double num = sums[0], den = sums[1], t0 = sums[2], s0 = sums[3];
    const double omega = num / den;
    const double rn = s0 - omega * t0;                       // = r_{j+1} . r0hat
    const double beta = rn / rho * alpha / omega;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->omega = omega;
        sc->beta = beta;
        sc->rho2[par ^ 1] = rn;
    }
    for (int64_t i2 = i0; i2 < n2; i2 += gs) {
        if (i2 != i0) { sv = ldpol(s, i2, pol, 3); av = ldpol(as, i2, pol, 4); vv = ldpol(ap, i2, pol, 5); pv = ldpol(p, i2, pol, 6); xv = ldpol(x, i2, pol, 7); }
        if constexpr (FS) {
            sv.x = __builtin_fma(-alpha, vv.x, sv.x);
            sv.y = __builtin_fma(-alpha, vv.y, sv.y);
        }
        xv.x = (xv.x + alpha * pv.x) + omega * sv.x;
        xv.y = (xv.y + alpha * pv.y) + omega * sv.y;
        stpol(x, i2, xv, pol, 8);   // (default past the caches: nobody reads x before the next step's sweep)
        const double2 rv = make_double2(sv.x - omega * av.x, sv.y - omega * av.y);
        stpol(r, i2, rv, pol, 9);   // (default past the caches: r is next read by K2, after K1 has streamed 250 MB; only p, K1's x, should stay)
        pv.x = rv.x + beta * (pv.x - omega * vv.x);
        pv.y = rv.y + beta * (pv.y - omega * vv.y);
        stpol(p, i2, pv, pol, 10);
    }
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        const double si = FS ? __builtin_fma(-alpha, ap[i], r[i]) : s[i];
        x[i] = (x[i] + alpha * p[i]) + omega * si;
        const double rv = si - omega * as[i];
        r[i] = rv;
        p[i] = rv + beta * (p[i] - omega * ap[i]);
    }
}

// K5: betaj = (rj1 <.> r0hat)/(r <.> r0hat) * alphaj / omegaj ; pj1 = rj1 ^+^ betaj .* (p ^-^ omegaj .* aap)
template <bool NT>
__global__ void __launch_bounds__(kBlock) bicg_k5_kernel(int64_t n, SolverScalars *sc, Parts rhonew, int par,
                                                          const double *r, const double *ap, double *p) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double rn = reduce_parts(rhonew.p, rhonew.n, rhonew.stride, s_red);
    const double omega = sc->omega;
    const double beta = rn / sc->rho2[par] * sc->alpha / omega;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rn; }
    SLA_VEC_LOOP_BEGIN(n)
        const double2 rv = ld2s<NT>(r, i2), av = ld2s<NT>(ap, i2);
        double2 pv = ld2s<NT>(p, i2);
        pv.x = rv.x + beta * (pv.x - omega * av.x);
        pv.y = rv.y + beta * (pv.y - omega * av.y);
        st2(p, i2, pv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) p[n - 1] = r[n - 1] + beta * (p[n - 1] - omega * ap[n - 1]);
}

int launch_bicg_k2(sla_ctx *c, int64_t n, SolverScalars *sc, Parts apr, int par, Parts res, int count_iter,
                   const double *r, const double *ap, double *s) {
    ProfScope prof(c, SLA_KERNEL_BICG_K2, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, bicg_k2_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, r, ap, s, c->vec_policy);
    else
        SLA_KLAUNCH(c, bicg_k2_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, r, ap, s, 0);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, const double *p, const double *s,
                   const double *as, const double *r0hat, double *x, double *r, double *prho) {
    ProfScope prof(c, SLA_KERNEL_BICG_K4, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, bicg_k4_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, p, s, as, r0hat, x, r, prho);
    else
        SLA_KLAUNCH(c, bicg_k4_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, p, s, as, r0hat, x, r, prho);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k5(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *ap, double *p) {
    ProfScope prof(c, SLA_KERNEL_BICG_K5, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, bicg_k5_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, ap, p);
    else
        SLA_KLAUNCH(c, bicg_k5_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, ap, p);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k45(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, Parts tr0, Parts sr0, int par, const double *s,
                    const double *as, const double *ap, double *x, double *r, double *p) {
    ProfScope prof(c, SLA_KERNEL_BICG_K45, true);
    if (!s) {   // (K2 folded into K3: s is rebuilt from r and ap)
        if (vec_stream_nt(c, n))
            SLA_KLAUNCH(c, (bicg_k45_kernel<true, true>), dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p, c->vec_policy);
        else
            SLA_KLAUNCH(c, (bicg_k45_kernel<false, true>), dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p, 0);
    } else if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, (bicg_k45_kernel<true, false>), dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p, c->vec_policy);
    else
        SLA_KLAUNCH(c, (bicg_k45_kernel<false, false>), dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p, 0);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// CGS (Sparse.hs:928-939): C2 / C4 (C1 = SpMV+dot, C3 = SpMV + r update + dot)
// ---------------------------------------------------------------------------------------------
// C2: alphaj ; q = u ^-^ alphaj .* aap ; uq = u ^+^ q ; xj1 = x ^+^ alphaj .* uq
template <bool NT>
__global__ void __launch_bounds__(kBlock) cgs_c2_kernel(int64_t n, SolverScalars *sc, Parts apr, int par,
                                                         Parts res, int count_iter, const double *u,
                                                         const double *aap, double *q, double *uq, double *x, int pol) {
    __shared__ double s_red[4];
    // (scalars, partials and the first element pairs issued together: see bicg_k2_kernel)
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kBlock, i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int done = sc->done;
    const double rho = sc->rho2[par];
    double pq[8];
    parts_issue(apr.p, apr.n, apr.stride, pq);
    const int64_t i0c = n2 > 0 ? min(i0, n2 - 1) : 0;
    double2 uv = make_double2(0.0, 0.0), av = uv, xv = uv;
    if (n2 > 0) { uv = ld2s<NT>(u, i0c); av = ld2s<NT>(aap, i0c); xv = ld2s<NT>(x, i0c); }
    if (done) return;
    if (res.p && residual_converged(sc, res.p, res.n, res.stride, s_red)) return;
    if (count_iter && blockIdx.x == 0 && threadIdx.x == 0) sc->iters += 1;
    const double alpha = rho / block_sum(parts_fold(pq, apr.n), s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
    for (int64_t i2 = i0; i2 < n2; i2 += gs) {
        if (i2 != i0) { uv = ld2s<NT>(u, i2); av = ld2s<NT>(aap, i2); xv = ld2s<NT>(x, i2); }
        const double2 qv = make_double2(uv.x - alpha * av.x, uv.y - alpha * av.y);
        const double2 sv = make_double2(uv.x + qv.x, uv.y + qv.y);
        xv.x += alpha * sv.x;
        xv.y += alpha * sv.y;
        stpol(q, i2, qv, pol, 11);
        stpol(uq, i2, sv, pol, 12);
        if (NT) st2_nt(x, i2, xv);  // (as in K4: x is not read again before the next step)
        else st2(x, i2, xv);
    }
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        const double qv = u[i] - alpha * aap[i], sv = u[i] + qv;
        q[i] = qv;
        uq[i] = sv;
        x[i] += alpha * sv;
    }
}

// C4: betaj = (rj1 <.> rhat) / (r <.> rhat) ; uj1 = rj1 ^+^ betaj .* q ; pj1 = uj1 ^+^ betaj .* (q ^+^ betaj .* p)
template <bool NT>
__global__ void __launch_bounds__(kBlock) cgs_c4_kernel(int64_t n, SolverScalars *sc, Parts rhonew, int par,
                                                         const double *r, const double *q, double *u, double *p, int pol) {
    __shared__ double s_red[4];
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kBlock, i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int done = sc->done;
    const double rho = sc->rho2[par];
    double pq[8];
    parts_issue(rhonew.p, rhonew.n, rhonew.stride, pq);
    const int64_t i0c = n2 > 0 ? min(i0, n2 - 1) : 0;
    double2 rv = make_double2(0.0, 0.0), qv = rv, pv = rv;
    if (n2 > 0) { rv = ld2s<NT>(r, i0c); qv = ld2s<NT>(q, i0c); pv = ld2s<NT>(p, i0c); }
    if (done) return;
    const double rn = block_sum(parts_fold(pq, rhonew.n), s_red);
    const double beta = rn / rho;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rn; }
    for (int64_t i2 = i0; i2 < n2; i2 += gs) {
        if (i2 != i0) { rv = ld2s<NT>(r, i2); qv = ld2s<NT>(q, i2); pv = ld2s<NT>(p, i2); }
        const double2 uv = make_double2(rv.x + beta * qv.x, rv.y + beta * qv.y);
        pv.x = uv.x + beta * (qv.x + beta * pv.x);
        pv.y = uv.y + beta * (qv.y + beta * pv.y);
        stpol(u, i2, uv, pol, 13);
        stpol(p, i2, pv, pol, 14);
    }
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        const double uv = r[i] + beta * q[i];
        u[i] = uv;
        p[i] = uv + beta * (q[i] + beta * p[i]);
    }
}

// C2 + C4 in one sweep (round 5; C2's u + q is built inside C3 -- spmv_wdia_march_kernel<.., 2> -- and nobody else reads q or u + q):
// q = u - alpha aap and u + q rebuilt here with C2's expressions, x updated, then C4's u and p from the new r.  Reads u, aap, x, r, p and
// writes x, u, p: 64 n bytes where C2 + C4 moved 88 n.  alpha was published by C3's prologue.
template <bool NT>
__global__ void __launch_bounds__(kBlock) cgs_c24_kernel(int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *aap,
                                                          double *u, double *p, double *x, int pol) {
    __shared__ double s_red[4];
    const int64_t n2 = n >> 1, gs = (int64_t)gridDim.x * kBlock, i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int done = sc->done;
    const double rho = sc->rho2[par], alpha = sc->alpha;
    double pq[8];
    parts_issue(rhonew.p, rhonew.n, rhonew.stride, pq);
    const int64_t i0c = n2 > 0 ? min(i0, n2 - 1) : 0;
    double2 rv = make_double2(0.0, 0.0), uv = rv, av = rv, pv = rv, xv = rv;
    if (n2 > 0) { rv = ld2s<NT>(r, i0c); uv = ld2s<NT>(u, i0c); av = ld2s<NT>(aap, i0c); pv = ld2s<NT>(p, i0c); xv = ld2s<NT>(x, i0c); }
    if (done) return;
    const double rn = block_sum(parts_fold(pq, rhonew.n), s_red);
    const double beta = rn / rho;
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rn; }
    for (int64_t i2 = i0; i2 < n2; i2 += gs) {
        if (i2 != i0) { rv = ld2s<NT>(r, i2); uv = ld2s<NT>(u, i2); av = ld2s<NT>(aap, i2); pv = ld2s<NT>(p, i2); xv = ld2s<NT>(x, i2); }
        const double2 qv = make_double2(uv.x - alpha * av.x, uv.y - alpha * av.y);
        const double2 sv = make_double2(uv.x + qv.x, uv.y + qv.y);
        xv.x += alpha * sv.x;
        xv.y += alpha * sv.y;
        if (NT) st2_nt(x, i2, xv);
        else st2(x, i2, xv);
        const double2 un = make_double2(rv.x + beta * qv.x, rv.y + beta * qv.y);
        pv.x = un.x + beta * (qv.x + beta * pv.x);
        pv.y = un.y + beta * (qv.y + beta * pv.y);
        stpol(u, i2, un, pol, 13);
        stpol(p, i2, pv, pol, 14);
    }
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        const double qv = u[i] - alpha * aap[i], sv = u[i] + qv;
        x[i] += alpha * sv;
        const double un = r[i] + beta * qv;
        u[i] = un;
        p[i] = un + beta * (qv + beta * p[i]);
    }
}
int launch_cgs_c24(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *aap, double *u, double *p, double *x) {
    ProfScope prof(c, SLA_KERNEL_CGS_C4, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, cgs_c24_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, aap, u, p, x, c->vec_policy);
    else
        SLA_KLAUNCH(c, cgs_c24_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, aap, u, p, x, 0);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_cgs_c2(sla_ctx *c, int64_t n, SolverScalars *sc, Parts apr, int par, Parts res, int count_iter,
                  const double *u, const double *aap, double *q, double *uq, double *x) {
    ProfScope prof(c, SLA_KERNEL_CGS_C2, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, cgs_c2_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, u, aap, q, uq, x, c->vec_policy);
    else
        SLA_KLAUNCH(c, cgs_c2_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, u, aap, q, uq, x, 0);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_cgs_c4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *q,
                  double *u, double *p) {
    ProfScope prof(c, SLA_KERNEL_CGS_C4, true);
    if (vec_stream_nt(c, n))
        SLA_KLAUNCH(c, cgs_c4_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, q, u, p, c->vec_policy);
    else
        SLA_KLAUNCH(c, cgs_c4_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, q, u, p, 0);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// CGNE N2: x1 = x ^+^ alphai .* p  (Sparse.hs:874)
__global__ void __launch_bounds__(kBlock) cgne_n2_kernel(int64_t n, SolverScalars *sc, const double *p, double *x) {
    if (sc->done) return;
    const double alpha = sc->alpha;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 pv = ld2(p, i2);
        double2 xv = ld2(x, i2);
        xv.x += alpha * pv.x;
        xv.y += alpha * pv.y;
        st2(x, i2, xv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) x[n - 1] += alpha * p[n - 1];
}
// CGNE N3, unfused (row-sharded path): beta = (r1.r1)/(r.r) ; p1 = t ^+^ beta .* p ; partial p1 . p1
__global__ void __launch_bounds__(kBlock) cgne_n3b_kernel(int64_t n, SolverScalars *sc, Parts rr1, int par, const double *t,
                                                           double *p, double *ppout) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double rr = reduce_parts(rr1.p, rr1.n, rr1.stride, s_red);
    const double beta = rr / sc->rho2[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rr; }
    double acc = 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 tv = ld2(t, i2);
        double2 pv = ld2(p, i2);
        pv.x = tv.x + beta * pv.x;
        pv.y = tv.y + beta * pv.y;
        st2(p, i2, pv);
        acc += pv.x * pv.x;
        acc += pv.y * pv.y;
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const double pv = t[n - 1] + beta * p[n - 1];
        p[n - 1] = pv;
        acc += pv * pv;
    }
    const double s = block_sum(acc, s_red);
    if (threadIdx.x == 0) ppout[blockIdx.x] = s;
}
int launch_cgne_n3b(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rr1, int par, const double *t, double *p, double *ppout) {
    hipLaunchKernelGGL(cgne_n3b_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rr1, par, t, p, ppout);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_cgne_n2(sla_ctx *c, int64_t n, SolverScalars *sc, const double *p, double *x) {
    hipLaunchKernelGGL(cgne_n2_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, p, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// BCG (extension: the commented bcgStep of Sparse.hs:899-909; `linSolve0 BCG_` keeps throwing like the reference, :1031)
// ---------------------------------------------------------------------------------------------
// B3: alpha = (r <.> rhat) / (aap <.> phat) ; x1 = x ^+^ alpha .* p ; r1 = r ^-^ alpha .* aap ;
//     rhat1 = rhat ^-^ alpha .* (transpose aa #> phat) ; partial r1 <.> rhat1          (5 reads + 3 writes of n doubles: 64 n bytes)
__global__ void __launch_bounds__(kBlock) bcg_b3_kernel(int64_t n, SolverScalars *sc, Parts app, int par, const double *p, const double *aap,
                                                         const double *atp, double *x, double *r, double *rhat, double *rrout) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double alpha = sc->rho2[par] / reduce_parts(app.p, app.n, app.stride, s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
    double acc = 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 pv = ld2(p, i2), av = ld2(aap, i2), tv = ld2(atp, i2);
        double2 xv = ld2(x, i2), rv = ld2(r, i2), hv = ld2(rhat, i2);
        xv.x += alpha * pv.x;
        xv.y += alpha * pv.y;
        rv.x -= alpha * av.x;
        rv.y -= alpha * av.y;
        hv.x -= alpha * tv.x;
        hv.y -= alpha * tv.y;
        st2(x, i2, xv);
        st2(r, i2, rv);
        st2(rhat, i2, hv);
        acc += rv.x * hv.x;
        acc += rv.y * hv.y;
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        x[i] += alpha * p[i];
        const double rv = r[i] - alpha * aap[i], hv = rhat[i] - alpha * atp[i];
        r[i] = rv;
        rhat[i] = hv;
        acc += rv * hv;
    }
    const double sum = block_sum(acc, s_red);
    if (threadIdx.x == 0) rrout[blockIdx.x] = sum;
}
// B4: beta = (r1 <.> rhat1) / (r <.> rhat) ; p1 = r1 ^+^ beta .* p ; phat1 = rhat1 ^+^ beta .* phat      (4 reads + 2 writes: 48 n bytes)
__global__ void __launch_bounds__(kBlock) bcg_b4_kernel(int64_t n, SolverScalars *sc, Parts rr1, int par, const double *r, const double *rhat,
                                                         double *p, double *phat) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double rr = reduce_parts(rr1.p, rr1.n, rr1.stride, s_red);
    const double beta = rr / sc->rho2[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rr; }
    SLA_VEC_LOOP_BEGIN(n)
        const double2 rv = ld2(r, i2), hv = ld2(rhat, i2);
        double2 pv = ld2(p, i2), qv = ld2(phat, i2);
        pv.x = rv.x + beta * pv.x;
        pv.y = rv.y + beta * pv.y;
        qv.x = hv.x + beta * qv.x;
        qv.y = hv.y + beta * qv.y;
        st2(p, i2, pv);
        st2(phat, i2, qv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        p[i] = r[i] + beta * p[i];
        phat[i] = rhat[i] + beta * phat[i];
    }
}
int launch_bcg_b3(sla_ctx *c, int64_t n, SolverScalars *sc, Parts app, int par, const double *p, const double *aap, const double *atp, double *x,
                  double *r, double *rhat, double *rrout) {
    SLA_KLAUNCH(c, bcg_b3_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, app, par, p, aap, atp, x, r, rhat, rrout);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bcg_b4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rr1, int par, const double *r, const double *rhat, double *p, double *phat) {
    SLA_KLAUNCH(c, bcg_b4_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rr1, par, r, rhat, p, phat);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// linSolve0 diagonal shortcut: reciprocal aa #> b  (Sparse.hs:1024-1025, Class.hs:174): every row holds
// exactly its diagonal entry, so val[i] is a_ii
__global__ void __launch_bounds__(kBlock) diag_solve_kernel(int64_t n, const double *diag, const double *b, double *x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        x[i] = (1.0 / diag[i]) * b[i];
}
int launch_diag_solve(sla_ctx *c, int64_t n, const double *diag, const double *b, double *x) {
    hipLaunchKernelGGL(diag_solve_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, diag, b, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// solver bookkeeping kernels (one workgroup)
// ---------------------------------------------------------------------------------------------
// end-of-batch residual test: same decision the next step's prologue would take
__global__ void __launch_bounds__(kBlock) check_kernel(SolverScalars *sc, Parts res) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double rn = sqrt(reduce_parts(res.p, res.n, res.stride, s_red));
    if (threadIdx.x == 0) {
        sc->resnorm = rn;
        if (sc->hist && sc->iters >= 1 && sc->iters <= sc->hist_cap) sc->hist[sc->iters - 1] = rn;   // (see residual_converged)
        if (rn <= sc->tol) { sc->done = 1; sc->flags |= SLA_FLAG_CONVERGED; }
        if (!is_finite(rn)) sc->flags |= SLA_FLAG_NONFINITE;
    }
}
int launch_check(sla_ctx *c, SolverScalars *sc, Parts res) {
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(kBlock), 0, stream_of(c), sc, res);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// rho2[0] = sum(rho) ; r0norm = sqrt(sum(r0sq)) ; tol = max tolAbs (tolRel * r0norm)  (Sparse.hs:1032-1037)
__global__ void __launch_bounds__(kBlock) init_scalars_kernel(SolverScalars *sc, Parts rho, Parts r0sq, double tol_abs,
                                                               double tol_rel, double *hist, int hist_cap) {
    __shared__ double s_red[4];
    const double rh = reduce_parts(rho.p, rho.n, rho.stride, s_red);
    const double r0 = sqrt(reduce_parts(r0sq.p, r0sq.n, r0sq.stride, s_red));
    if (threadIdx.x == 0) {
        sc->rho2[0] = rh;
        sc->rho2[1] = rh;
        sc->alpha = sc->omega = sc->beta = 0.0;
        sc->resnorm = __builtin_nan("");
        sc->r0norm = r0;
        sc->tol = fmax(tol_abs, tol_rel * r0);
        sc->hnorm = 0.0;
        sc->done = 0;
        sc->iters = 0;
        sc->flags = 0;
        sc->kdone = 0;
        sc->hist = hist;
        sc->hist_cap = hist_cap;
    }
}
__global__ void __launch_bounds__(kBlock) set_rho_kernel(SolverScalars *sc, Parts rho, int par) {
    __shared__ double s_red[4];
    const double v = reduce_parts(rho.p, rho.n, rho.stride, s_red);
    if (threadIdx.x == 0) sc->rho2[par] = v;
}
int launch_set_rho(sla_ctx *c, SolverScalars *sc, Parts rho, int par) {
    hipLaunchKernelGGL(set_rho_kernel, dim3(1), dim3(kBlock), 0, stream_of(c), sc, rho, par);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_init_scalars(sla_ctx *c, SolverScalars *sc, Parts rho, Parts r0sq, double tol_abs, double tol_rel, double *hist, int hist_cap) {
    hipLaunchKernelGGL(init_scalars_kernel, dim3(1), dim3(kBlock), 0, stream_of(c), sc, rho, r0sq, tol_abs, tol_rel, hist, hist_cap);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

}  // namespace sla
