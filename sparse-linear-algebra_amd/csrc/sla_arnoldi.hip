// sla_arnoldi.hip -- arnoldi (Numeric/LinearAlgebra/Sparse.hs:630-667) on a column-major device basis: classical Gram-Schmidt against
// the SAME A q_i in two passes over Q (h = Q^T w ; w -= Q h, ||w||^2), normalisation with the breakdown test (:665-667), and
// x += Q y for GMRES.  Tall-skinny passes at 0.25 flop/byte: HBM-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

// ---------------------------------------------------------------------------------------------
// Arnoldi (Sparse.hs:630-667): classical Gram-Schmidt against the SAME A q_i, two passes over Q
// ---------------------------------------------------------------------------------------------
// pass 1: parts[j * gridDim.x + block] = partial of (q_j <.> w), j < ncols     (hhcoli, :655)
// 2-D grid: blockIdx.y selects a group of NC columns.  One workgroup streaming all (up to 32) columns at once reads the basis at
// 4.6 TB/s; groups per workgroup (w re-read per group, from the caches) on the 2 M-row banded problem, GMRES(30) Arnoldi steps/s, same
// box, round 4 (loads of a group issued back to back): 1 column 6820, **2: 7120**, 3: 6990, 4: 6950, 8: 6520 (profiles/r04_ab_arnoldi.txt).
#ifndef SLA_ARN_DOTS_GROUP
#define SLA_ARN_DOTS_GROUP 2
#endif
constexpr int kArnDotsGroup = SLA_ARN_DOTS_GROUP;   // basis columns per workgroup of the dots pass
template <int NC>
__global__ void __launch_bounds__(kBlock) arn_dots_kernel(int64_t n, const double *Q, int64_t ldq, int ncols,
                                                           const double *w, double *parts, SolverScalars *sc) {
    __shared__ double s_w[4][NC];
    if (arn_stopped(sc)) return;
    Q += (int64_t)blockIdx.y * NC * ldq;
    parts += (int64_t)blockIdx.y * NC * gridDim.x;
    ncols = min(ncols - (int)blockIdx.y * NC, NC);
    double acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = 0.0;
    // No load of the streaming loop stands under a condition (round 4): with `if (j < ncols)` around each column the compiler put
    // s_waitcnt vmcnt(0) behind every single load -- one trip to memory per column and iteration, hidden by occupancy alone (the pass
    // ran at 5.1 TB/s).  A full group loads its NC columns back to back; only the last, partial group keeps the conditions.
    if (ncols == NC) {
        SLA_VEC_LOOP_BEGIN(n)
            const double2 wv = ld2(w, i2);
            double2 qv[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) qv[j] = ld2(Q + (int64_t)j * ldq, i2);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                acc[j] += qv[j].x * wv.x;
                acc[j] += qv[j].y * wv.y;
            }
        SLA_VEC_LOOP_END
    } else {
        SLA_VEC_LOOP_BEGIN(n)
            const double2 wv = ld2(w, i2);
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (j < ncols) {
                    const double2 qv = ld2(Q + (int64_t)j * ldq, i2);
                    acc[j] += qv.x * wv.x;
                    acc[j] += qv.y * wv.y;
                }
        SLA_VEC_LOOP_END
    }
    if (SLA_HAS_TAIL(n)) {
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < ncols) acc[j] += Q[(int64_t)j * ldq + n - 1] * w[n - 1];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const double s = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < ncols) {
        const int j = threadIdx.x;
        parts[(int64_t)j * gridDim.x + blockIdx.x] = ((s_w[0][j] + s_w[1][j]) + s_w[2][j]) + s_w[3][j];
    }
}

// pass 2: w := aqi ^-^ foldl' (^+^) (zipWith (.*) hhcoli qv)   (:657-658); partial ||w||^2; H column
// NT: the basis is read non-temporally in THIS pass when it overflows the memory-side cache: the columns the dots pass
// just allocated there then survive for the next pass instead of both passes cycling through an LRU that holds neither
// (GMRES(30) at 2 M rows, Q = 0.5 GB: +6 % steps/s).
template <bool NT>
__global__ void __launch_bounds__(kBlock) arn_update_kernel(int64_t n, const double *Q, int64_t ldq, int ncols,
                                                             const double *hp, int np, int cs, int stride, double *w,
                                                             double *pn, double *Hcol, SolverScalars *sc) {
    __shared__ double s_h[64];
    __shared__ double s_red[4];
    if (arn_stopped(sc)) return;
    // every workgroup re-reduces the ncols dot products in the same fixed order
    if (stride == 1 && (np & 63) == 0 && ((ncols + 3) >> 2) * (np >> 6) <= 16) {
        // the dots pass's own partials (single-rank contexts, round 4: no fold launch in between; arn_dots_grid sizes the pass so that
        // a wavefront's columns x partials fit sixteen loads per lane): ALL of them issued before the first add -- one round trip at the
        // head of the kernel, not one per column -- then per column the same order as the generic loop below: lane sums, wave_sum
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, M = np >> 6;
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = wave + 4 * (t / M), m = t - (t / M) * M;
            v[t] = j < ncols && t / M < 16 ? hp[(int64_t)min(j, ncols - 1) * cs + lane + 64 * m] : 0.0;
        }
        for (int jj = 0; wave + 4 * jj < ncols; ++jj) {
            double a = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t / M == jj) a += v[t];
            a = wave_sum(a);
            if (lane == 0) s_h[wave + 4 * jj] = a;
        }
    } else {
        for (int j = threadIdx.x >> 6; j < ncols; j += 4) {
            double a = 0.0;
            for (int i = threadIdx.x & 63; i < np; i += 64) a += hp[(int64_t)j * cs + (int64_t)i * stride];
            a = wave_sum(a);
            if ((threadIdx.x & 63) == 0) s_h[j] = a;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < ncols) Hcol[threadIdx.x] = s_h[threadIdx.x];
    double acc = 0.0;
    // The columns go eight (then four, then one to three) at a time, the loads of a group issued back to back and none under a
    // condition (round 4: the `if (j < ncols)` form of rounds 1-3 waited for every column's load on its own -- 4.3 TB/s); the
    // coefficients are LDS broadcasts.  The sum over the columns is formed in the same order, one column after the other.
#define SLA_ARN_COLS(G)                                                                  \
    {                                                                                    \
        double2 qv[G];                                                                   \
        _Pragma("unroll") for (int k = 0; k < G; ++k) qv[k] = ld2s<NT>(Q + (int64_t)(g + k) * ldq, i2); \
        _Pragma("unroll") for (int k = 0; k < G; ++k) {                                  \
            const double hk = s_h[g + k];                                                \
            t.x += hk * qv[k].x;                                                         \
            t.y += hk * qv[k].y;                                                         \
        }                                                                                \
        g += G;                                                                          \
    }
    SLA_VEC_LOOP_BEGIN(n)
        double2 wv = ld2(w, i2);
        double2 t = make_double2(0.0, 0.0);
        int g = 0;
        while (g + 8 <= ncols) SLA_ARN_COLS(8)
        if (g + 4 <= ncols) SLA_ARN_COLS(4)
        const int rem = ncols - g;
        if (rem == 3) SLA_ARN_COLS(3)
        else if (rem == 2) SLA_ARN_COLS(2)
        else if (rem == 1) SLA_ARN_COLS(1)
        wv.x -= t.x;
        wv.y -= t.y;
        st2(w, i2, wv);
        acc += wv.x * wv.x;
        acc += wv.y * wv.y;
    SLA_VEC_LOOP_END
#undef SLA_ARN_COLS
    if (SLA_HAS_TAIL(n)) {
        double t = 0.0;
        for (int j = 0; j < ncols; ++j) t += s_h[j] * Q[(int64_t)j * ldq + n - 1];
        const double wv = w[n - 1] - t;
        w[n - 1] = wv;
        acc += wv * wv;
    }
    const double s = block_sum(acc, s_red);
    if (threadIdx.x == 0) pn[blockIdx.x] = s;
}

// qip = normalize2 qipnn = (recip (norm2 w)) .* w ; h_{i+1,i} = norm2' w ; breakdown = nearZero (:659-667)
__global__ void __launch_bounds__(kBlock) arn_normalize_kernel(int64_t n, Parts nrm, const double *w, double *qnext,
                                                                double *hsub, SolverScalars *sc, int first) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double nn = sqrt(reduce_parts(nrm.p, nrm.n, nrm.stride, s_red));
    const double inv = 1.0 / nn;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (hsub) *hsub = nn;
        sc->hnorm = nn;
        if (hsub) sc->kdone += 1;
        // arnInit performs no breakdown test (:643-651); arnoldiStep does (:665-667).  Only the FLAG is raised here: `done` is
        // this kernel's own exit test, and workgroups starting after workgroup 0 wrote it would skip their part of q_{i+1}
        // (the reference appends the complete normalize2 result).  The next kernel of the chain (the SpMV of step i + 1:
        // spmv_prologue / arn_stopped) sees the flag -- written by an EARLIER launch, so every workgroup agrees -- exits and
        // promotes it to `done`.
        if (!first && fabs(nn) <= 1e-12) sc->flags |= SLA_FLAG_BREAKDOWN;
    }
    SLA_VEC_LOOP_BEGIN(n)
        const double2 wv = ld2(w, i2);
        st2(qnext, i2, make_double2(inv * wv.x, inv * wv.y));
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) qnext[n - 1] = inv * w[n - 1];
}

// x := x + sum_j y[j] q_j   (GMRES update x = x0 + Q_k y)
template <int NC>
__global__ void __launch_bounds__(kBlock) gemv_accum_kernel(int64_t n, const double *Q, int64_t ldq, int ncols,
                                                             const double *ycoef, double *x) {
    double h[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) h[j] = j < ncols ? ycoef[j] : 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        double2 xv = ld2(x, i2);
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < ncols) {
                const double2 qv = ld2(Q + (int64_t)j * ldq, i2);
                xv.x += h[j] * qv.x;
                xv.y += h[j] * qv.y;
            }
        st2(x, i2, xv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        double xv = x[n - 1];
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < ncols) xv += h[j] * Q[(int64_t)j * ldq + n - 1];
        x[n - 1] = xv;
    }
}

int arn_grid(int64_t n) {
    int g = vec_grid(n);
    return g > kArnGridMax ? kArnGridMax : g;
}

#define SLA_NC_DISPATCH(ncols, CALL)                              \
    do {                                                          \
        if ((ncols) <= 4) { CALL(4); }                            \
        else if ((ncols) <= 8) { CALL(8); }                       \
        else if ((ncols) <= 16) { CALL(16); }                     \
        else if ((ncols) <= 32) { CALL(32); }                     \
        else if ((ncols) <= 64) { CALL(64); }                     \
        else return fail(SLA_ERR_INVALID, "Krylov basis > 64 columns"); \
    } while (0)

// x-grid (= partials per column) of the dots pass.  Its 2-D grid has ceil(ncols / 4) column groups, so the chip is filled with ~1024
// workgroups in all when x = 64 floor(16 / groups): and then the update pass can fold the partials of its wavefronts' columns itself with
// sixteen loads per lane (arn_update_kernel's head) instead of behind a one-workgroup fold launch (round 4: a 4.8 us launch and a
// dependent dispatch less per Arnoldi step).
int arn_dots_grid(int64_t n, int ncols) {
    const int g = arn_grid(n), groups = (ncols + 3) / 4;   // (columns per wavefront of the update pass's head: four wavefronts)
    const int gx = 64 * std::max(1, 16 / groups);
    return g < 64 ? g : std::min(g & ~63, gx);
}
int launch_arn_dots(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *w, double *parts,
                    SolverScalars *sc) {
    const int g = arn_dots_grid(n, ncols);
    if (ncols < 1 || ncols > 64) return fail(SLA_ERR_INVALID, "Krylov basis: 1..64 columns");
    hipLaunchKernelGGL((arn_dots_kernel<kArnDotsGroup>), dim3(g, (ncols + kArnDotsGroup - 1) / kArnDotsGroup), dim3(kBlock), 0, stream_of(c),
                       n, Q, ldq, ncols, w, parts, sc);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_arn_update(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *hp, int np, int cs,
                      int stride, double *w, double *pn, double *Hcol, SolverScalars *sc) {
    const int g = arn_grid(n);
    // the basis read so far (ncols columns) against the memory-side cache
    const bool nt = c->vec_nt < 0 ? (int64_t)ncols * 8 * n > c->mall_bytes : c->vec_nt != 0;
    if (ncols < 1 || ncols > 64) return fail(SLA_ERR_INVALID, "Krylov basis: 1..64 columns");
    if (nt) hipLaunchKernelGGL((arn_update_kernel<true>), dim3(g), dim3(kBlock), 0, stream_of(c), n, Q, ldq, ncols, hp, np, cs, stride, w, pn, Hcol, sc);
    else hipLaunchKernelGGL((arn_update_kernel<false>), dim3(g), dim3(kBlock), 0, stream_of(c), n, Q, ldq, ncols, hp, np, cs, stride, w, pn, Hcol, sc);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_arn_normalize(sla_ctx *c, int64_t n, Parts nrm, const double *w, double *qnext, double *hsub,
                         SolverScalars *sc, int first) {
    hipLaunchKernelGGL(arn_normalize_kernel, dim3(arn_grid(n)), dim3(kBlock), 0, stream_of(c), n, nrm, w, qnext, hsub, sc, first);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_gemv_accum(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *ycoef_dev, double *x) {
    const int g = arn_grid(n);
#define CALL(NC) hipLaunchKernelGGL((gemv_accum_kernel<NC>), dim3(g), dim3(kBlock), 0, stream_of(c), n, Q, ldq, ncols, ycoef_dev, x)
    SLA_NC_DISPATCH(ncols, CALL);
#undef CALL
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

}  // namespace sla
