// sla_spmv_lpanel.hip -- (#>) for matrices with dense rows (hundreds to thousands of entries per row, random columns): x is cut into
// panels of 16384 columns kept in LDS by one 1024-thread workgroup per CU; (panel, row) segments streamed two at a time per
// wavefront, partial sums per panel, finished in ascending panel order with the fused epilogue.  Data/Sparse/Common.hs:242-260.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

// ---------------------------------------------------------------------------------------------
// LDS-panel SpMV for matrices with dense rows ("1 % density": hundreds of entries per row, random columns)
// ---------------------------------------------------------------------------------------------
// Random 8-byte gathers from the L2 move a 128-byte line into the L1 each (64 B/clk/CU): ~0.3 gathers per clock per
// CU, which bounds the stream kernel at ~1/4 of the HBM rate on such matrices.  The LDS serves the same gathers at
// 128 B/clk of useful data, so x is cut into equal panels of W <= kLpW columns: a workgroup keeps one panel of x in LDS and
// streams the (row, panel) segments of its row chunks -- one wavefront per segment, lanes striding over the entries
// in ascending order, two segments in flight per wavefront -- into per-panel partial sums.  Tasks (panel, row chunk)
// are dealt out panel-major in contiguous runs of equal entry counts (task_begin), so a workgroup reloads x about once.  lpanel_finish_kernel then
// adds the partials of a row in ascending panel order and runs the fused epilogue.
// PM: the entries come from the panel-major copy, whose columns are 16-bit offsets into the panel (W <= 16384): 10 B per entry
template <typename RP, int L, int R, int J, bool PM>
__global__ void __launch_bounds__(kLpBlock) spmv_lpanel_kernel(const RP *__restrict__ pp, const int32_t *__restrict__ col,
                                                               const double *__restrict__ val, const double *__restrict__ xg,
                                                               double *__restrict__ ypart, const int32_t *__restrict__ task_begin,
                                                               int rows, int n, int W, int chunk_rows, int C, int col_lo, int col_hi,
                                                               const SolverScalars *sc) {
    // L lanes per (row, panel) segment, R segments per lane group and round, J strided loads per segment and round: a
    // wavefront keeps (64 / L) * R segments = 64 * R * J entries in flight.  A round costs a memory round trip however
    // little it carries (measured: ~0.9 us), so short segments get narrow groups -- see the table at the launch.
    extern __shared__ double lp_xs[];
    if (sc && sc->done) return;
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    constexpr int GW = 64 / L, GPB = (kLpBlock / 64) * GW;   // lane groups per wavefront / per workgroup
    const int gid = wv * GW + ln / L, gl = ln % L;
    const int t0 = task_begin[blockIdx.x], t1 = task_begin[blockIdx.x + 1];
    int curp = -1;
    for (int t = t0; t < t1; ++t) {
        const int p = t / C, c = t - p * C;
        const int w0 = p * W;
        if (p != curp) {
            __syncthreads();
            const int wn = min(W, n - w0);
            // only [col_lo, col_hi] is referenced by these rows -- and, on a row slab gathering from its in-place halo
            // window, the only part of x that is backed by memory at all
            for (int j = tid; j < wn; j += kLpBlock) lp_xs[j] = (w0 + j >= col_lo && w0 + j <= col_hi) ? xg[w0 + j] : 0.0;
            __syncthreads();
            curp = p;
        }
        const int lo = c * chunk_rows, hi = min(rows, lo + chunk_rows);
        // segment (p, i): [pp[p][i], pp[p + 1][i]) of the row-major arrays, or -- panel-major copy, where the segments of a panel follow
        // each other and a segment's edge lines are its neighbours' -- [q[p rows + i], q[p rows + i + 1])
        const RP *ps = pp + (int64_t)p * rows, *pe = ps + (PM ? 1 : rows);
        const uint16_t *col16 = (const uint16_t *)col;
        double *yp = ypart + (int64_t)p * rows;
        for (int base = lo; base < hi; base += R * GPB) {   // (wavefront-uniform trip count)
            // (fetching the next round's segment pointers a round ahead was tried: no gain where each shape is used)
            RP k[R], e[R];
            double acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int i = base + r * GPB + gid;
                if constexpr (L == 64) i = __builtin_amdgcn_readfirstlane(i);   // one segment per wavefront: scalar pointer loads
                const bool has = i < hi;
                k[r] = (has ? ps[i] : 0) + (PM ? 2 : 1) * gl;
                e[r] = has ? pe[i] : 0;
                acc[r] = 0.0;
            }
            bool more = true;
            if constexpr (PM) {
                // panel-major copy: a lane takes PAIRS of consecutive entries (segments start on even entries; an odd segment ends
                // with a pad whose column is 0xffff): one 4-byte and one 16-byte load per pair
                static_assert(J % 2 == 0, "pairs");
                typedef double lp_f64x2 __attribute__((ext_vector_type(2)));
                while (more) {
                    uint32_t cj[R][J / 2];
                    lp_f64x2 vj[R][J / 2];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
#pragma unroll
                        for (int j = 0; j < J / 2; ++j) {
                            if (k[r] + 2 * L * j < e[r]) {
                                cj[r][j] = __builtin_nontemporal_load((const uint32_t *)(col16 + k[r] + 2 * L * j));
                                vj[r][j] = __builtin_nontemporal_load((const lp_f64x2 *)(val + k[r] + 2 * L * j));
                            }
                        }
                    }
                    bool mine = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
#pragma unroll
                        for (int j = 0; j < J / 2; ++j) {
                            if (k[r] + 2 * L * j < e[r]) {
                                acc[r] = acc[r] + vj[r][j].x * lp_xs[cj[r][j] & 0xffffu];
                                if ((cj[r][j] >> 16) != 0xffffu) acc[r] = acc[r] + vj[r][j].y * lp_xs[cj[r][j] >> 16];
                            }
                        }
                        k[r] += L * J;
                        mine |= k[r] - 2 * gl < e[r];
                    }
                    more = __builtin_amdgcn_ballot_w64(mine) != 0;
                }
            } else {
                while (more) {
                    int32_t cj[R][J];
                    double vj[R][J];
                    // all loads of the round in flight before the first use (nothing may touch a loaded value in this phase)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            if (k[r] + L * j < e[r]) {
                                cj[r][j] = __builtin_nontemporal_load(col + k[r] + L * j);
                                vj[r][j] = __builtin_nontemporal_load(val + k[r] + L * j);
                            }
                        }
                    }
                    bool mine = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            if (k[r] + L * j < e[r]) {
                                const double prod = vj[r][j] * lp_xs[cj[r][j] - w0];
                                acc[r] = acc[r] + prod;
                            }
                        }
                        k[r] += L * J;
                        mine |= k[r] - gl < e[r];   // (the segment's next base: uniform over the lane group)
                    }
                    more = __builtin_amdgcn_ballot_w64(mine) != 0;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = base + r * GPB + gid;
                double sum = acc[r];
                if constexpr (L == 64) {
                    sum = wave_sum(sum);
                } else {
#pragma unroll
                    for (int off = L / 2; off > 0; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
                }
                if (gl == 0 && i < hi) yp[i] = sum;
            }
        }
    }
}

// Panel-major second copy (lowering time): segment (p, i) of the row-major arrays goes to q[p rows + i].  Read row-major the
// segments of a panel are ~150 entries each, 2000 entries apart: their first and last 128-byte lines are fetched again for the
// neighbouring panels' segments, by other workgroups at other times -- 13 % of the kernel's HBM traffic on the 200 k x 2000-per-row
// matrix (PMC: 5.47 GB per launch for 4.83 GB of entries).  The copy's columns are 16-bit offsets into the panel: 10 B per entry.
// One wavefront per segment.
template <typename RP>
__global__ void __launch_bounds__(kBlock) lp_reorder_kernel(const RP *__restrict__ pp, const RP *__restrict__ q, const int32_t *__restrict__ col,
                                                            const double *__restrict__ val, uint16_t *__restrict__ col2, double *__restrict__ val2,
                                                            int64_t rows, int64_t nseg, int32_t W) {
    const int ln = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); s < nseg; s += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t b = (int64_t)pp[s], e = (int64_t)pp[s + rows], d = (int64_t)q[s];
        const int32_t w0 = (int32_t)(s / rows) * W;          // the segment's panel starts at column w0: 16-bit offsets (W <= 16384)
        for (int64_t k = b + ln; k < e; k += 64) {
            col2[d + (k - b)] = (uint16_t)(col[k] - w0);
            val2[d + (k - b)] = val[k];
        }
        if (((e - b) & 1) && ln == 0) {                      // segments are padded to whole pairs: the pad is recognised by its column
            col2[d + (e - b)] = 0xffffu;
            val2[d + (e - b)] = 0.0;
        }
    }
}

int launch_lp_reorder(sla_ctx *c, bool rp64, const void *pp, const void *q, const int32_t *col, const double *val, uint16_t *col2, double *val2,
                      int64_t rows, int64_t P, int32_t W) {
    const int64_t nseg = rows * P;
    const int grid = (int)std::min<int64_t>(64 * (int64_t)c->n_cu, std::max<int64_t>(1, (nseg + 3) / 4));
    if (rp64)
        hipLaunchKernelGGL(lp_reorder_kernel<int64_t>, dim3(grid), dim3(kBlock), 0, stream_of(c), (const int64_t *)pp, (const int64_t *)q, col, val, col2, val2, rows, nseg, W);
    else
        hipLaunchKernelGGL(lp_reorder_kernel<int32_t>, dim3(grid), dim3(kBlock), 0, stream_of(c), (const int32_t *)pp, (const int32_t *)q, col, val, col2, val2, rows, nseg, W);
    SLA_HIP_TRY(hipGetLastError());
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    return SLA_OK;
}

// y_i = sum over panels (ascending) of the partials + the fused epilogue; one lane per row.
template <int EPI, typename RP>
__global__ void __launch_bounds__(kBlock) lpanel_finish_kernel(SpmvArgs<RP> a, const double *__restrict__ ypart, int P) {
    __shared__ double s_red[4];
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.rows; row += (int64_t)gridDim.x * kBlock) {
        double acc = ypart[row];
        for (int p = 1; p < P; ++p) acc = acc + ypart[(int64_t)p * a.rows + row];
        spmv_epilogue<EPI, RP>(a, (int)row, acc, coef, acc1, acc2);
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT ||
                  EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (threadIdx.x == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (threadIdx.x == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, (int)threadIdx.x);
}

// launcher (called by launch_spmv, sla_spmv.hip): the panel kernel on its persistent task grid, then the finish kernel (row sums of
// the panel partials in ascending panel order + the fused epilogue) on `grid` workgroups
namespace {
template <int EPI, typename RP>
int launch_lpanel_t(const sla_csr *A, const SpmvArgs<RP> &a, int grid) {
    sla_ctx *c = A->ctx;
    // lane-group shape by mean segment length (A->lp_cfg, set at lowering): 64 lanes x 2 segments x 4 loads for long
    // segments, narrower groups with more segments per wavefront for short ones
#define SLA_LP_LAUNCH_PM(CFG, L_, R_, J_, PM_)                                                                                \
    {                                                                                                                       \
        const int attr_bit = 1 << (4 * CFG + 2 * (PM_ ? 1 : 0) + (std::is_same<RP, int32_t>::value ? 0 : 1));              \
        if (!(c->lp_attr & attr_bit)) {   /* per context = per device: 128 KiB of dynamic LDS */                            \
            SLA_HIP_TRY(hipFuncSetAttribute((const void *)spmv_lpanel_kernel<RP, L_, R_, J_, PM_>,                          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLpW * sizeof(double))));     \
            c->lp_attr |= attr_bit;                                                                                         \
        }                                                                                                                   \
        hipLaunchKernelGGL((spmv_lpanel_kernel<RP, L_, R_, J_, PM_>), dim3(A->lp_G), dim3(kLpBlock), kLpW * sizeof(double), \
                           stream_of(c), (const RP *)A->d_lpp, PM_ ? (const int32_t *)A->d_lpcol : a.col, PM_ ? A->d_lpval : a.val, a.x, \
                           A->d_lpy, A->d_lpt, a.rows, (int)A->n, A->lp_W, A->lp_chunk, A->lp_C, A->lp_col_lo, A->lp_col_hi,  \
                           (const SolverScalars *)a.sc);                                                                    \
    }
#define SLA_LP_LAUNCH(CFG, L_, R_, J_)                                  \
    case CFG:                                                           \
        if (A->d_lpcol) SLA_LP_LAUNCH_PM(CFG, L_, R_, J_, true)         \
        else SLA_LP_LAUNCH_PM(CFG, L_, R_, J_, false)                   \
        break;
    switch (A->lp_cfg) {
        SLA_LP_LAUNCH(1, 32, 4, 2)
        SLA_LP_LAUNCH(2, 16, 4, 2)
        SLA_LP_LAUNCH(3, 8, 4, 2)
        default:
        SLA_LP_LAUNCH(0, 64, kLpRowsInFlight, 4)
    }
#undef SLA_LP_LAUNCH_PM
#undef SLA_LP_LAUNCH
    SLA_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((lpanel_finish_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_lpy, A->lp_P);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
template <typename RP>
int launch_lpanel_rp(const sla_csr *A, int epi, const SpmvArgs<RP> &a, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_lpanel_t<EPI_NONE, RP>(A, a, grid);
        case EPI_DOT: return launch_lpanel_t<EPI_DOT, RP>(A, a, grid);
        case EPI_DOT2: return launch_lpanel_t<EPI_DOT2, RP>(A, a, grid);
        case EPI_DOT4: return launch_lpanel_t<EPI_DOT4, RP>(A, a, grid);
        case EPI_RES: return launch_lpanel_t<EPI_RES, RP>(A, a, grid);
        case EPI_AXPY_DOT: return launch_lpanel_t<EPI_AXPY_DOT, RP>(A, a, grid);
        case EPI_XPBY_NRM: return launch_lpanel_t<EPI_XPBY_NRM, RP>(A, a, grid);
        case EPI_SUB: return launch_lpanel_t<EPI_SUB, RP>(A, a, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_lpanel: unknown epilogue");
}
}  // namespace

// the finish kernel alone (the flat form, sla_spmv_lflat.hip, leaves its partials in the same layout)
int launch_lpanel_finish(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) {
    sla_ctx *c = A->ctx;
#define SLA_LPF(E) case E: hipLaunchKernelGGL((lpanel_finish_kernel<E, int32_t>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_lpy, A->lp_P); break
    switch (epi) {
        SLA_LPF(EPI_NONE); SLA_LPF(EPI_DOT); SLA_LPF(EPI_DOT2); SLA_LPF(EPI_DOT4); SLA_LPF(EPI_RES); SLA_LPF(EPI_AXPY_DOT); SLA_LPF(EPI_XPBY_NRM); SLA_LPF(EPI_SUB);
        default: return fail(SLA_ERR_INVALID, "launch_lpanel_finish: unknown epilogue");
    }
#undef SLA_LPF
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_spmv_lpanel(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) { return launch_lpanel_rp<int32_t>(A, epi, a, grid); }
int launch_spmv_lpanel(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid) { return launch_lpanel_rp<int64_t>(A, epi, a, grid); }

}  // namespace sla
