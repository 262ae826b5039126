// sla_spmv_stream.hip -- the general CSR-stream (#>) kernels: f64 values + i32 columns streamed once with coalesced non-temporal
// loads, products staged in LDS, rows reduced from LDS by one lane (short rows: the reference's ascending left fold bit for bit,
// Data/Sparse/Common.hs:242-260), by a sub-wavefront segment, by a wavefront or by the whole workgroup (long rows).
//   spmv_stream_kernel   the plain form
//   spmv_xwin_kernel     + a 768-entry window of x staged in LDS (entries clustered around the diagonal)
//   spmv_dual_kernel     K1 and the true residual of the previous iterate in one matrix sweep (linSolve0, Sparse.hs:1043-1052)
//   spmv_scalar_kernel   one lane per row (A/B baseline, SLA_SPMV_ALGO=scalar)
// (sla_spmv_pipe.hip holds the three-stage pipelined variant of the plain form.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

typedef int sla_i32x2 __attribute__((ext_vector_type(2)));
typedef double sla_f64x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// CSR-stream SpMV
// ---------------------------------------------------------------------------------------------
// (Measured on MI355X, 216^3 Laplacian: lane-contiguous 16-byte col / 32-byte val loads at arbitrary
// entry offsets ran at 3.95 TB/s against 4.87 TB/s for the lane-strided dword / dwordx2 form below.)
// Software-pipelined persistent loop.  Per row block: (1) the col/val/rowptr loads were issued one
// iteration earlier and are consumed now (gather x, products -> LDS), (2) the NEXT row block's loads
// are issued before the barrier so they fly during (3) the per-row reduction from LDS.  Row-block
// descriptors (rb, rbk) are indexed by the block number only, so they prefetch without a dependent
// chain.  s_prod / s_rp are double-buffered: one barrier per row block.
template <int EPI, typename RP>
__global__ void __launch_bounds__(kBlock, (kOcc8<EPI, RP>)) spmv_stream_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val, const int32_t *__restrict__ rb,
                                                                 const RP *__restrict__ rbk, const double *__restrict__ xg,
                                                                 int xcd_remap, int wide) {
    __shared__ double s_prod[2][kNnzPerRowBlock];
    __shared__ int s_rp[2][kMaxRowsPerRowBlock + 1];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;

    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    int b = wk.first;
    if (b < wk.last) {
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int32_t c[4];
        double v[4];
        RP rpn = 0;
        // issue the streaming loads of row block (r0_, k0_, k1_) into c / v / rpn
        // Wide form (default): every lane takes two PAIRS of consecutive entries -- one 8-byte col load and one 16-byte val load
        // each, from the even entry at or below the block's first (kb): 512 B / 1 KiB per wave-instruction instead of 256 / 512
        // with one entry per load.  Measured: -6 % time on the bare skeleton (tools/stream_width_probe.cpp), -4 % / -1 % on K1 of the
        // 216^3 Laplacian on two boxes.  A pair may reach one entry before or behind the block (another block's entry, or the
        // arrays' zeroed slack): loaded, never used.  The one block shape it cannot hold (1024 entries from an odd entry on)
        // takes the narrow form.
#define SLA_WIDE_OK(k0_, k1_) (wide && (int)((k1_) - (k0_)) + (int)((k0_) & 1) <= kNnzPerRowBlock)
#define SLA_ISSUE_LOADS(r0_, r1_, k0_, k1_)                                              \
        if ((k1_) - (k0_) <= (RP)kNnzPerRowBlock) {                                          \
            const int cnt_ = (int)((k1_) - (k0_));                                           \
            if (SLA_WIDE_OK(k0_, k1_)) {                                                     \
                const int odd_ = (int)((k0_) & 1);                                           \
                const RP kb_ = (k0_) - odd_;                                                 \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                              \
                    const int i = 2 * tid + j * 2 * kBlock;                                  \
                    if (i < cnt_ + odd_) {                                                   \
                        const sla_i32x2 cc = __builtin_nontemporal_load((const sla_i32x2 *)(col + kb_ + i)); \
                        const sla_f64x2 vv = __builtin_nontemporal_load((const sla_f64x2 *)(val + kb_ + i)); \
                        c[2 * j] = cc.x; c[2 * j + 1] = cc.y;                                \
                        v[2 * j] = vv.x; v[2 * j + 1] = vv.y;                                \
                    }                                                                        \
                }                                                                            \
            } else {                                                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                              \
                    const int i = tid + j * kBlock;                                          \
                    if (i < cnt_) {                                                          \
                        c[j] = __builtin_nontemporal_load(col + (k0_) + i);                \
                        v[j] = __builtin_nontemporal_load(val + (k0_) + i);                \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
            if (tid < (r1_) - (r0_)) rpn = rowptr[(r0_) + tid];                            \
        }
        SLA_ISSUE_LOADS(r0, r1, k0, k1)
        // Row-block descriptors are scalar loads.  They share lgkmcnt with LDS traffic and return out of
        // order, so consuming one drains all of them: fetch the descriptors of block b+2 AFTER block b+1's
        // have been consumed (right behind the prefetch loads), a whole iteration before they are needed.
        int nr0 = 0, nr1 = 0;
        RP nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) {
            nr0 = rb[b + wk.step];
            nr1 = rb[b + wk.step + 1];
            nk0 = rbk[b + wk.step];
            nk1 = rbk[b + wk.step + 1];
        }
#define SLA_FETCH_DESC()          \
        if (has_next2) {          \
            fr0 = rb[bnn];        \
            fr1 = rb[bnn + 1];    \
            fk0 = rbk[bnn];       \
            fk1 = rbk[bnn + 1];   \
        }
        int buf = 0;
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fr1 = 0;   // descriptors two row blocks ahead (scalar loads, issued below)
            RP fk0 = 0, fk1 = 0;
            const int nrows = r1 - r0;
            if (k1 - k0 <= (RP)kNnzPerRowBlock) {
                const int cnt = (int)(k1 - k0);
                double *prod = s_prod[buf];
                int *rp = s_rp[buf];
                if (tid < nrows) rp[tid] = (int)(rpn - k0);
                if (tid == 0) rp[nrows] = cnt;
                // one lane per row (below): its epilogue operands travel with the gathers.  Loaded inside the epilogue they were one more
                // exposed round trip between the fold and the y store (K3 with its two operands: 271 -> 228 us same-box at 216^3).
                // (Holding the y store back to the next block's gathers, so that its acknowledgement never stands alone in front of a
                // wait, was tried with it: four more VGPRs, the four-sum instantiation spills at 64 -- K3 back at 269 us.)
                const bool lane_per_row = nrows > 64 || cnt <= 8 * nrows;
                double wpre = 0.0, zpre = 0.0;
                if (lane_per_row && tid < nrows) spmv_operands<EPI, RP>(a, r0 + tid, wpre, zpre);
                if (SLA_WIDE_OK(k0, k1)) {
                    const int odd = (int)(k0 & 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 2 * tid + (j >> 1) * 2 * kBlock + (j & 1) - odd;   // (-1: the entry in front of an odd first entry)
                        if ((unsigned)i < (unsigned)cnt) prod[i] = v[j] * xg[c[j]];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = tid + j * kBlock;
                        if (i < cnt) prod[i] = v[j] * xg[c[j]];
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
                __syncthreads();
                if (lane_per_row) {
                    // one lane per row, ascending left fold: the reference's summation order exactly
                    if (tid < nrows) {
                        const int s = rp[tid], e = rp[tid + 1];
                        // column-panel passes continue the running sum of the previous panels: still one
                        // ascending left fold per row
                        double acc = a.yinit ? a.yinit[r0 + tid] : 0.0;
                        for (int k = s; k < e; ++k) acc += prod[k];
                        spmv_epilogue_pre<EPI, RP>(a, r0 + tid, acc, coef, acc1, acc2, wpre, zpre);
                    }
                } else {
                    // few, longer rows: a power-of-two segment of the wavefront per row
                    int np2 = 1;
                    while (np2 < nrows) np2 <<= 1;
                    const int tpr = min(64, kBlock / np2);
                    const int g = tid / tpr, l = tid - g * tpr;
                    double acc = 0.0;
                    if (g < nrows) {
                        const int e = rp[g + 1];
                        for (int k = rp[g] + l; k < e; k += tpr) acc += prod[k];
                    }
                    for (int off = tpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                    if (g < nrows && l == 0) {
                        if (a.yinit) acc += a.yinit[r0 + g];
                        spmv_epilogue<EPI, RP>(a, r0 + g, acc, coef, acc1, acc2);
                    }
                }
                buf ^= 1;
            } else if (nrows > 1 || k1 - k0 <= (RP)kWaveRowMax) {
                // up to 4 long rows (> 1024 entries each): one wavefront per row, no LDS, no barrier,
                // 4 coalesced col/val loads in flight per lane
                const int wv = tid >> 6, ln = tid & 63;
                if (wv < nrows) {
                    const RP s0 = rowptr[r0 + wv], s1 = rowptr[r0 + wv + 1];
                    double acc = 0.0;
                    RP k = s0 + ln;
                    for (; k + 192 < s1; k += 256) {
                        const int32_t c0 = __builtin_nontemporal_load(col + k);
                        const int32_t c1 = __builtin_nontemporal_load(col + k + 64);
                        const int32_t c2 = __builtin_nontemporal_load(col + k + 128);
                        const int32_t c3 = __builtin_nontemporal_load(col + k + 192);
                        const double v0 = __builtin_nontemporal_load(val + k);
                        const double v1 = __builtin_nontemporal_load(val + k + 64);
                        const double v2 = __builtin_nontemporal_load(val + k + 128);
                        const double v3 = __builtin_nontemporal_load(val + k + 192);
                        acc += v0 * xg[c0];
                        acc += v1 * xg[c1];
                        acc += v2 * xg[c2];
                        acc += v3 * xg[c3];
                    }
                    for (; k < s1; k += 64) acc += val[k] * xg[col[k]];
                    acc = wave_sum(acc);
                    if (ln == 0) {
                        if (a.yinit) acc += a.yinit[r0 + wv];
                        spmv_epilogue<EPI, RP>(a, r0 + wv, acc, coef, acc1, acc2);
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
            } else {
                // one very long row (> kWaveRowMax entries) owned by the whole workgroup
                double acc = 0.0;
                RP k = k0 + tid;
                for (; k + 3 * kBlock < k1; k += 4 * kBlock) {
                    const int32_t c0 = __builtin_nontemporal_load(col + k);
                    const int32_t c1 = __builtin_nontemporal_load(col + k + kBlock);
                    const int32_t c2 = __builtin_nontemporal_load(col + k + 2 * kBlock);
                    const int32_t c3 = __builtin_nontemporal_load(col + k + 3 * kBlock);
                    const double v0 = __builtin_nontemporal_load(val + k);
                    const double v1 = __builtin_nontemporal_load(val + k + kBlock);
                    const double v2 = __builtin_nontemporal_load(val + k + 2 * kBlock);
                    const double v3 = __builtin_nontemporal_load(val + k + 3 * kBlock);
                    acc += v0 * xg[c0];
                    acc += v1 * xg[c1];
                    acc += v2 * xg[c2];
                    acc += v3 * xg[c3];
                }
                for (; k < k1; k += kBlock) acc += val[k] * xg[col[k]];
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
                double s = block_sum(acc, s_red);
                if (tid == 0) {
                    if (a.yinit) s += a.yinit[r0];
                    spmv_epilogue<EPI, RP>(a, r0, s, coef, acc1, acc2);
                }
            }
            if (!has_next) break;
            b = bn;
            r0 = nr0;
            r1 = nr1;
            k0 = nk0;
            k1 = nk1;
            nr0 = fr0;
            nr1 = fr1;
            nk0 = fk0;
            nk1 = fk1;
        }
#undef SLA_FETCH_DESC
#undef SLA_ISSUE_LOADS
#undef SLA_WIDE_OK
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT ||
                  EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (tid == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (tid == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, tid);
}

// CSR-stream SpMV with an LDS-staged window of x.  For matrices whose entries cluster around the diagonal
// (stencils, banded) each row block loads x[wlo, wlo + kXWin) once with coalesced loads and serves every
// gather that falls inside it from LDS; only the far legs go to L1/L2.  Everything is single-buffered
// except the row offsets (two barriers per row block).
template <int EPI, typename RP>
__global__ void __launch_bounds__(kBlock, (kOcc8<EPI, RP>)) spmv_xwin_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val, const int32_t *__restrict__ rb,
                                                                 const RP *__restrict__ rbk, const double *__restrict__ xg,
                                                                 const int32_t *__restrict__ rbw, int32_t ncols, int xcd_remap) {
    __shared__ double s_prod[1][kNnzPerRowBlock];
    __shared__ int s_rp[2][kMaxRowsPerRowBlock + 1];
    __shared__ double s_xw[kXWin];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;

    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    int b = wk.first;
    if (b < wk.last) {
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int32_t c[4];
        double v[4];
        double xw[kXWin / kBlock];
        int wlo = rbw[b];
        RP rpn = 0;
        // issue the streaming loads of row block (r0_, k0_, k1_) into c / v / rpn, and its x window
#define SLA_ISSUE_LOADS(r0_, r1_, k0_, k1_, wlo_)                                           \
        if ((k1_) - (k0_) <= (RP)kNnzPerRowBlock) {                                          \
            const int cnt_ = (int)((k1_) - (k0_));                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                  \
                const int i = tid + j * kBlock;                                              \
                if (i < cnt_) {                                                              \
                    c[j] = __builtin_nontemporal_load(col + (k0_) + i);                    \
                    v[j] = __builtin_nontemporal_load(val + (k0_) + i);                    \
                }                                                                            \
            }                                                                                \
            if (tid < (r1_) - (r0_)) rpn = rowptr[(r0_) + tid];                            \
            _Pragma("unroll") for (int j = 0; j < kXWin / kBlock; ++j) {                     \
                const int i = (wlo_) + tid + j * kBlock;                                     \
                xw[j] = i < ncols ? xg[i] : 0.0;                                             \
            }                                                                                \
        }
        SLA_ISSUE_LOADS(r0, r1, k0, k1, wlo)
        // Row-block descriptors are scalar loads.  They share lgkmcnt with LDS traffic and return out of
        // order, so consuming one drains all of them: fetch the descriptors of block b+2 AFTER block b+1's
        // have been consumed (right behind the prefetch loads), a whole iteration before they are needed.
        int nr0 = 0, nr1 = 0, nwlo = 0;
        RP nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) {
            nwlo = rbw[b + wk.step];
            nr0 = rb[b + wk.step];
            nr1 = rb[b + wk.step + 1];
            nk0 = rbk[b + wk.step];
            nk1 = rbk[b + wk.step + 1];
        }
#define SLA_FETCH_DESC()          \
        if (has_next2) {          \
            fwlo = rbw[bnn];      \
            fr0 = rb[bnn];        \
            fr1 = rb[bnn + 1];    \
            fk0 = rbk[bnn];       \
            fk1 = rbk[bnn + 1];   \
        }
        int buf = 0;
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fr1 = 0, fwlo = 0;   // descriptors two row blocks ahead (scalar loads, issued below)
            RP fk0 = 0, fk1 = 0;
            const int nrows = r1 - r0;
            if (k1 - k0 <= (RP)kNnzPerRowBlock) {
                const int cnt = (int)(k1 - k0);
                double *prod = s_prod[0];
                int *rp = s_rp[buf];
                if (tid < nrows) rp[tid] = (int)(rpn - k0);
                if (tid == 0) rp[nrows] = cnt;
#pragma unroll
                for (int j = 0; j < kXWin / kBlock; ++j) s_xw[tid + j * kBlock] = xw[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < cnt) {
                        const unsigned off = (unsigned)(c[j] - wlo);   // x from the LDS window when inside it
                        const double xv = off < (unsigned)kXWin ? s_xw[off] : xg[c[j]];
                        prod[i] = v[j] * xv;
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                SLA_FETCH_DESC()
                __syncthreads();
                if (nrows > 64 || cnt <= 8 * nrows) {
                    // one lane per row, ascending left fold: the reference's summation order exactly
                    if (tid < nrows) {
                        const int s = rp[tid], e = rp[tid + 1];
                        double acc = 0.0;
                        for (int k = s; k < e; ++k) acc += prod[k];
                        spmv_epilogue<EPI, RP>(a, r0 + tid, acc, coef, acc1, acc2);
                    }
                } else {
                    // few, longer rows: a power-of-two segment of the wavefront per row
                    int np2 = 1;
                    while (np2 < nrows) np2 <<= 1;
                    const int tpr = min(64, kBlock / np2);
                    const int g = tid / tpr, l = tid - g * tpr;
                    double acc = 0.0;
                    if (g < nrows) {
                        const int e = rp[g + 1];
                        for (int k = rp[g] + l; k < e; k += tpr) acc += prod[k];
                    }
                    for (int off = tpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                    if (g < nrows && l == 0) spmv_epilogue<EPI, RP>(a, r0 + g, acc, coef, acc1, acc2);
                }
                buf ^= 1;
            } else if (nrows > 1 || k1 - k0 <= (RP)kWaveRowMax) {
                // up to 4 long rows (> 1024 entries each): one wavefront per row, no LDS, no barrier,
                // 4 coalesced col/val loads in flight per lane
                const int wv = tid >> 6, ln = tid & 63;
                if (wv < nrows) {
                    const RP s0 = rowptr[r0 + wv], s1 = rowptr[r0 + wv + 1];
                    double acc = 0.0;
                    RP k = s0 + ln;
                    for (; k + 192 < s1; k += 256) {
                        const int32_t c0 = __builtin_nontemporal_load(col + k);
                        const int32_t c1 = __builtin_nontemporal_load(col + k + 64);
                        const int32_t c2 = __builtin_nontemporal_load(col + k + 128);
                        const int32_t c3 = __builtin_nontemporal_load(col + k + 192);
                        const double v0 = __builtin_nontemporal_load(val + k);
                        const double v1 = __builtin_nontemporal_load(val + k + 64);
                        const double v2 = __builtin_nontemporal_load(val + k + 128);
                        const double v3 = __builtin_nontemporal_load(val + k + 192);
                        acc += v0 * xg[c0];
                        acc += v1 * xg[c1];
                        acc += v2 * xg[c2];
                        acc += v3 * xg[c3];
                    }
                    for (; k < s1; k += 64) acc += val[k] * xg[col[k]];
                    acc = wave_sum(acc);
                    if (ln == 0) spmv_epilogue<EPI, RP>(a, r0 + wv, acc, coef, acc1, acc2);
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                SLA_FETCH_DESC()
            } else {
                // one very long row (> kWaveRowMax entries) owned by the whole workgroup
                double acc = 0.0;
                RP k = k0 + tid;
                for (; k + 3 * kBlock < k1; k += 4 * kBlock) {
                    const int32_t c0 = __builtin_nontemporal_load(col + k);
                    const int32_t c1 = __builtin_nontemporal_load(col + k + kBlock);
                    const int32_t c2 = __builtin_nontemporal_load(col + k + 2 * kBlock);
                    const int32_t c3 = __builtin_nontemporal_load(col + k + 3 * kBlock);
                    const double v0 = __builtin_nontemporal_load(val + k);
                    const double v1 = __builtin_nontemporal_load(val + k + kBlock);
                    const double v2 = __builtin_nontemporal_load(val + k + 2 * kBlock);
                    const double v3 = __builtin_nontemporal_load(val + k + 3 * kBlock);
                    acc += v0 * xg[c0];
                    acc += v1 * xg[c1];
                    acc += v2 * xg[c2];
                    acc += v3 * xg[c3];
                }
                for (; k < k1; k += kBlock) acc += val[k] * xg[col[k]];
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                SLA_FETCH_DESC()
                const double s = block_sum(acc, s_red);
                if (tid == 0) spmv_epilogue<EPI, RP>(a, r0, s, coef, acc1, acc2);
            }
            if (!has_next) break;
            b = bn;
            r0 = nr0;
            r1 = nr1;
            k0 = nk0;
            k1 = nk1;
            wlo = nwlo;
            nwlo = fwlo;
            nr0 = fr0;
            nr1 = fr1;
            nk0 = fk0;
            nk1 = fk1;
        }
#undef SLA_FETCH_DESC
#undef SLA_ISSUE_LOADS
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT ||
                  EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (tid == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (tid == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, tid);
}

// Dual SpMV: ONE pass over the matrix applied to two vectors.  y = A x with the K1 epilogue (p1 += y . w)
// and, from the same col/val stream, partials of ||A x2 - b2||^2 (p2).  linSolve0 evaluates the true
// residual of the previous step's x' here, inside the next step's K1, instead of paying a third matrix
// sweep per iteration (36 nnz + 180 n  ->  24 nnz + 196 n bytes per reference-faithful iteration).
// Same row-block walk and software pipeline as spmv_stream_kernel; the two product arrays share the
// LDS budget, so the stage is single-buffered (two barriers per row block).
template <typename RP>
__global__ void __launch_bounds__(kBlock, 8) spmv_dual_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val, const int32_t *__restrict__ rb,
                                                                 const RP *__restrict__ rbk, const double *__restrict__ xg,
                                                               const double *__restrict__ x2, const double *__restrict__ b2,
                                                               int xcd_remap) {
    __shared__ double s_prod[2][kNnzPerRowBlock];
    __shared__ int s_rp[kMaxRowsPerRowBlock + 1];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI_DOT, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    int b = wk.first;
    if (b < wk.last) {
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int32_t c[4];
        double v[4];
        RP rpn = 0;
#define SLA_ISSUE_LOADS(r0_, r1_, k0_, k1_)                                              \
        if ((k1_) - (k0_) <= (RP)kNnzPerRowBlock) {                                          \
            const int cnt_ = (int)((k1_) - (k0_));                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                  \
                const int i = tid + j * kBlock;                                              \
                if (i < cnt_) {                                                              \
                    c[j] = __builtin_nontemporal_load(col + (k0_) + i);                    \
                    v[j] = __builtin_nontemporal_load(val + (k0_) + i);                    \
                }                                                                            \
            }                                                                                \
            if (tid < (r1_) - (r0_)) rpn = rowptr[(r0_) + tid];                            \
        }
        SLA_ISSUE_LOADS(r0, r1, k0, k1)
        // Row-block descriptors are scalar loads.  They share lgkmcnt with LDS traffic and return out of
        // order, so consuming one drains all of them: fetch the descriptors of block b+2 AFTER block b+1's
        // have been consumed (right behind the prefetch loads), a whole iteration before they are needed.
        int nr0 = 0, nr1 = 0;
        RP nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) {
            nr0 = rb[b + wk.step];
            nr1 = rb[b + wk.step + 1];
            nk0 = rbk[b + wk.step];
            nk1 = rbk[b + wk.step + 1];
        }
#define SLA_FETCH_DESC()          \
        if (has_next2) {          \
            fr0 = rb[bnn];        \
            fr1 = rb[bnn + 1];    \
            fk0 = rbk[bnn];       \
            fk1 = rbk[bnn + 1];   \
        }
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fr1 = 0;   // descriptors two row blocks ahead (scalar loads, issued below)
            RP fk0 = 0, fk1 = 0;
            const int nrows = r1 - r0;
            if (k1 - k0 <= (RP)kNnzPerRowBlock) {
                const int cnt = (int)(k1 - k0);
                if (tid < nrows) s_rp[tid] = (int)(rpn - k0);
                if (tid == 0) s_rp[nrows] = cnt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < cnt) {
                        s_prod[0][i] = v[j] * xg[c[j]];
                        s_prod[1][i] = v[j] * x2[c[j]];
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
                __syncthreads();
                if (nrows > 64 || cnt <= 8 * nrows) {
                    if (tid < nrows) {
                        const int s = s_rp[tid], e = s_rp[tid + 1];
                        double ya = 0.0, yb = 0.0;
                        for (int k = s; k < e; ++k) {
                            ya += s_prod[0][k];
                            yb += s_prod[1][k];
                        }
                        const int row = r0 + tid;
                        a.y[row] = ya;
                        acc1 += ya * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                } else {
                    int np2 = 1;
                    while (np2 < nrows) np2 <<= 1;
                    const int tpr = min(64, kBlock / np2);
                    const int g = tid / tpr, l = tid - g * tpr;
                    double ya = 0.0, yb = 0.0;
                    if (g < nrows) {
                        const int e = s_rp[g + 1];
                        for (int k = s_rp[g] + l; k < e; k += tpr) {
                            ya += s_prod[0][k];
                            yb += s_prod[1][k];
                        }
                    }
                    for (int off = tpr >> 1; off > 0; off >>= 1) {
                        ya += __shfl_xor(ya, off, 64);
                        yb += __shfl_xor(yb, off, 64);
                    }
                    if (g < nrows && l == 0) {
                        const int row = r0 + g;
                        a.y[row] = ya;
                        acc1 += ya * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                }
                __syncthreads();
            } else if (nrows > 1 || k1 - k0 <= (RP)kWaveRowMax) {
                const int wv = tid >> 6, ln = tid & 63;
                if (wv < nrows) {
                    const RP s0 = rowptr[r0 + wv], s1 = rowptr[r0 + wv + 1];
                    double ya = 0.0, yb = 0.0;
                    RP k = s0 + ln;
                    for (; k + 64 < s1; k += 128) {
                        const int32_t c0 = __builtin_nontemporal_load(col + k);
                        const int32_t c1 = __builtin_nontemporal_load(col + k + 64);
                        const double v0 = __builtin_nontemporal_load(val + k);
                        const double v1 = __builtin_nontemporal_load(val + k + 64);
                        ya += v0 * xg[c0];
                        yb += v0 * x2[c0];
                        ya += v1 * xg[c1];
                        yb += v1 * x2[c1];
                    }
                    for (; k < s1; k += 64) {
                        const int32_t cc = col[k];
                        const double vv = val[k];
                        ya += vv * xg[cc];
                        yb += vv * x2[cc];
                    }
                    ya = wave_sum(ya);
                    yb = wave_sum(yb);
                    if (ln == 0) {
                        const int row = r0 + wv;
                        a.y[row] = ya;
                        acc1 += ya * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
            } else {
                double ya = 0.0, yb = 0.0;
                for (RP k = k0 + tid; k < k1; k += kBlock) {
                    const int32_t cc = col[k];
                    const double vv = val[k];
                    ya += vv * xg[cc];
                    yb += vv * x2[cc];
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
                const double sa = block_sum(ya, s_red);
                const double sb = block_sum(yb, s_red);
                if (tid == 0) {
                    a.y[r0] = sa;
                    acc1 += sa * a.w[r0];
                    const double t = sb - b2[r0];
                    acc2 += t * t;
                }
            }
            if (!has_next) break;
            b = bn;
            r0 = nr0;
            r1 = nr1;
            k0 = nk0;
            k1 = nk1;
            nr0 = fr0;
            nr1 = fr1;
            nk0 = fk0;
            nk1 = fk1;
        }
#undef SLA_FETCH_DESC
#undef SLA_ISSUE_LOADS
    }
    const double s1 = block_sum(acc1, s_red);
    if (tid == 0) a.p1[blockIdx.x] = s1;
    const double s2 = block_sum(acc2, s_red);
    if (tid == 0) a.p2[blockIdx.x] = s2;
}

// One lane per row, grid-stride: the A/B baseline for the stream kernel (SLA_SPMV_ALGO=scalar).
template <int EPI, typename RP>
__global__ void __launch_bounds__(kBlock) spmv_scalar_kernel(SpmvArgs<RP> a, int xcd_remap) {
    __shared__ double s_red[4];
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.rows;
         row += (int64_t)gridDim.x * kBlock) {
        const RP s = a.rowptr[row], e = a.rowptr[row + 1];
        double acc = 0.0;
        for (RP k = s; k < e; ++k) {
            const double prod = a.val[k] * a.x[a.col[k]];
            acc = acc + prod;
        }
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

// ---------------------------------------------------------------------------------------------
// launchers (called by launch_spmv, sla_spmv.hip)
// ---------------------------------------------------------------------------------------------
namespace {
template <int EPI, typename RP>
int launch_stream_t(const sla_csr *A, const SpmvArgs<RP> &a, int grid) {
    sla_ctx *c = A->ctx;
    if (c->spmv_algo == 1)
        hipLaunchKernelGGL((spmv_scalar_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, c->xcd_remap);
    else if (stream_xwin_on(A))
        hipLaunchKernelGGL((spmv_xwin_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk,
                           a.x, A->d_rbw, (int32_t)A->n, c->xcd_remap);
    else
        hipLaunchKernelGGL((spmv_stream_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk,
                           a.x, c->xcd_remap, c->stream_wide);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
template <typename RP>
int launch_stream_rp(const sla_csr *A, int epi, const SpmvArgs<RP> &a, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_stream_t<EPI_NONE, RP>(A, a, grid);
        case EPI_DOT: return launch_stream_t<EPI_DOT, RP>(A, a, grid);
        case EPI_DOT2: return launch_stream_t<EPI_DOT2, RP>(A, a, grid);
        case EPI_DOT4: return launch_stream_t<EPI_DOT4, RP>(A, a, grid);
        case EPI_RES: return launch_stream_t<EPI_RES, RP>(A, a, grid);
        case EPI_AXPY_DOT: return launch_stream_t<EPI_AXPY_DOT, RP>(A, a, grid);
        case EPI_XPBY_NRM: return launch_stream_t<EPI_XPBY_NRM, RP>(A, a, grid);
        case EPI_SUB: return launch_stream_t<EPI_SUB, RP>(A, a, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_stream: unknown epilogue");
}
template <typename RP>
int launch_dual_rp(const sla_csr *A, const SpmvArgs<RP> &a, const double *x2, const double *b2, int grid) {
    sla_ctx *c = A->ctx;
    hipLaunchKernelGGL((spmv_dual_kernel<RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk, a.x, x2, b2,
                       c->xcd_remap);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
}  // namespace

// plain CSR-stream forms: spmv_stream_kernel / spmv_xwin_kernel, or the one-lane-per-row baseline (SLA_SPMV_ALGO=scalar)
int launch_spmv_stream(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) { return launch_stream_rp<int32_t>(A, epi, a, grid); }
int launch_spmv_stream(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid) { return launch_stream_rp<int64_t>(A, epi, a, grid); }
// K1 and the true residual of the previous iterate from one matrix sweep (linSolve0)
int launch_spmv_dual(const sla_csr *A, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid) { return launch_dual_rp<int32_t>(A, a, x2, b2, grid); }
int launch_spmv_dual(const sla_csr *A, const SpmvArgs<int64_t> &a, const double *x2, const double *b2, int grid) { return launch_dual_rp<int64_t>(A, a, x2, b2, grid); }

}  // namespace sla
