// sla_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels of the solver hot path.
//
// Everything here is HBM-bandwidth bound (0.17-0.25 flop/byte): no MFMA.  The design rules are
//   * wave64, 256-thread workgroups, persistent grids sized to the 256 CUs;
//   * CSR values / column indices streamed once with fully coalesced non-temporal loads and staged
//     as products in LDS ("CSR-stream"), rows reduced from LDS by one lane (short rows: bit-exact
//     with the reference's left fold), by a sub-wavefront segment, or by the whole block (long row);
//   * the x gather goes through L1/L2; the block->row-block map keeps each XCD's L2 on a contiguous
//     slab of rows (blocks are dispatched round-robin over the 8 XCDs);
//   * BLAS-1 work is fused into the SpMV epilogue or into 16-byte-per-lane streaming kernels;
//   * every reduction is two-stage and deterministic: producers write one partial per block, the
//     FIRST consumer kernel re-reduces the partials in a fixed order in its prologue and block 0
//     publishes the scalar for later kernels.  No atomics, no host round trip inside an iteration.
//
// SpMV variants, all sharing the row-block walk, the software pipeline and the epilogues (launch_spmv_t picks):
//   spmv_stream_kernel     general CSR-stream: val + i32 col, products staged in LDS
//   spmv_xwin_kernel       + a 768-entry window of x staged in LDS (entries clustered around the diagonal)
//   spmv_diag_kernel       + 1-byte dictionary codes instead of i32 columns (<= 256 distinct diagonals),
//                            gather moved to the row phase
//   spmv_wdia_kernel       per 128-row slice the union of its (offset, value) pairs with lane masks, all in SGPRs,
//                            two rows per lane: one 16-byte gather + 2 x (v_mul + v_add) per entry, no LDS, no
//                            barrier (constant-coefficient stencils)
//   spmv_vdict_kernel      1-byte codes into a table of (diagonal offset, value) pairs: no val stream at all
//                            (<= 256 distinct pairs: constant-coefficient stencils), 256-row blocks, lane per row
//   spmv_dual_kernel / spmv_dual_diag_kernel   K1 and the true residual of the previous iterate in one sweep
//   launch_spmv_panels     column-panel passes of spmv_stream_kernel for irregular matrices with x > L2
//   spmv_scalar_kernel     one lane per row (A/B baseline, SLA_SPMV_ALGO=scalar)
//
// Reference semantics implemented (file:line relative to the reference repo):
//   (#>)  Data/Sparse/Common.hs:242-260      (<.>)/norm2  Data/Sparse/SpVector.hs:116-129
//   bicgstabStep Numeric/LinearAlgebra/Sparse.hs:972-981   cgsStep :928-939   cgneStep :870-878
//   linSolve0 runIter :1043-1052               arnoldi :630-667
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

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
__global__ void __launch_bounds__(kBlock, 8) spmv_stream_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val, const int32_t *__restrict__ rb,
                                                                 const RP *__restrict__ rbk, const double *__restrict__ xg,
                                                                 int xcd_remap) {
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
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < cnt) prod[i] = v[j] * xg[c[j]];
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1) }
                SLA_FETCH_DESC()
                __syncthreads();
                if (nrows > 64 || cnt <= 8 * nrows) {
                    // one lane per row, ascending left fold: the reference's summation order exactly
                    if (tid < nrows) {
                        const int s = rp[tid], e = rp[tid + 1];
                        // column-panel passes continue the running sum of the previous panels: still one
                        // ascending left fold per row
                        double acc = a.yinit ? a.yinit[r0 + tid] : 0.0;
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
__global__ void __launch_bounds__(kBlock, 8) spmv_xwin_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
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

// ---------------------------------------------------------------------------------------------
// CSR-stream SpMV with dictionary-compressed column indices
// ---------------------------------------------------------------------------------------------
// Stencil / banded matrices have only a handful of distinct diagonals: col - row takes <= 256 values.
// For those the lowering keeps, next to the canonical i32 column array, one BYTE per entry (an index into a
// sorted table of the diagonal offsets).  This kernel streams val (8 B) + code (1 B) instead of val + col
// (12 B): 25 % less HBM traffic per entry.  The code only decodes next to its row index, so the gather moves
// from the product phase to the row phase: val / code are staged raw in LDS, then one lane per row (or a
// wavefront segment per row) decodes col = row + dict[code], takes x from the LDS window or from L2 and
// accumulates with SEPARATE multiply and add roundings (the reference's left fold, bit for bit).
template <int EPI, typename RP, bool XW>
__global__ void __launch_bounds__(kBlock, 8) spmv_diag_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr,
                                                               const uint8_t *__restrict__ code, const double *__restrict__ val,
                                                               const int32_t *__restrict__ rb, const RP *__restrict__ rbk,
                                                               const double *__restrict__ xg, const int32_t *__restrict__ rbw,
                                                               const int32_t *__restrict__ dict, int32_t ncols, int32_t grow0,
                                                               int xcd_remap) {
    __shared__ double s_val[kNnzPerRowBlock];
    __shared__ uint8_t s_code[kNnzPerRowBlock];
    __shared__ int s_rp[2][kMaxRowsPerRowBlock + 1];
    __shared__ double s_xw[XW ? kXWin : 1];
    __shared__ int s_dict[256];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    s_dict[tid] = dict[tid];
    __syncthreads();

    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    int b = wk.first;
    if (b < wk.last) {
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int32_t c[4];
        double v[4];
        double xw[XW ? kXWin / kBlock : 1];
        int wlo = XW ? rbw[b] : 0;
        RP rpn = 0;
#define SLA_ISSUE_LOADS(r0_, r1_, k0_, k1_, wlo_)                                           \
        if ((k1_) - (k0_) <= (RP)kNnzPerRowBlock) {                                          \
            const int cnt_ = (int)((k1_) - (k0_));                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                  \
                const int i = tid + j * kBlock;                                              \
                if (i < cnt_) {                                                              \
                    c[j] = __builtin_nontemporal_load(code + (k0_) + i);                     \
                    v[j] = __builtin_nontemporal_load(val + (k0_) + i);                      \
                }                                                                            \
            }                                                                                \
            if (tid < (r1_) - (r0_)) rpn = rowptr[(r0_) + tid];                              \
            if (XW) {                                                                        \
                _Pragma("unroll") for (int j = 0; j < kXWin / kBlock; ++j) {                 \
                    const int i = (wlo_) + tid + j * kBlock;                                 \
                    xw[j] = i < ncols ? xg[i] : 0.0;                                         \
                }                                                                            \
            }                                                                                \
        }
        SLA_ISSUE_LOADS(r0, r1, k0, k1, wlo)
        int nr0 = 0, nr1 = 0, nwlo = 0;
        RP nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) {
            if (XW) nwlo = rbw[b + wk.step];
            nr0 = rb[b + wk.step];
            nr1 = rb[b + wk.step + 1];
            nk0 = rbk[b + wk.step];
            nk1 = rbk[b + wk.step + 1];
        }
        // x[col] for col = global row + diagonal offset: LDS window first, L1/L2 otherwise
        auto xat = [&](int colg) -> double {
            if (XW) {
                const unsigned off = (unsigned)(colg - wlo);
                if (off < (unsigned)kXWin) return s_xw[off];
            }
            return xg[colg];
        };
        int buf = 0;
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fr1 = 0, fwlo = 0;
            RP fk0 = 0, fk1 = 0;
            const int nrows = r1 - r0;
            if (k1 - k0 <= (RP)kNnzPerRowBlock) {
                const int cnt = (int)(k1 - k0);
                int *rp = s_rp[buf];
                if (tid < nrows) rp[tid] = (int)(rpn - k0);
                if (tid == 0) rp[nrows] = cnt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < cnt) {
                        s_val[i] = v[j];
                        s_code[i] = (uint8_t)c[j];
                    }
                }
                if (XW) {
#pragma unroll
                    for (int j = 0; j < kXWin / kBlock; ++j) s_xw[tid + j * kBlock] = xw[j];
                }
                __syncthreads();
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
                if (nrows > 64 || cnt <= 8 * nrows) {
                    // one lane per row: decode, gather, multiply, add -- ascending, separately rounded
                    if (tid < nrows) {
                        const int s = rp[tid], e = rp[tid + 1];
                        const int grow = grow0 + r0 + tid;
                        double acc = 0.0;
                        {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
                            int k = s;
                            for (; k + 4 <= e; k += 4) {  // 4 gathers in flight, summed in order
                                const double x0 = xat(grow + s_dict[s_code[k]]);
                                const double x1 = xat(grow + s_dict[s_code[k + 1]]);
                                const double x2 = xat(grow + s_dict[s_code[k + 2]]);
                                const double x3 = xat(grow + s_dict[s_code[k + 3]]);
                                const double p0 = s_val[k] * x0, p1 = s_val[k + 1] * x1;
                                const double p2 = s_val[k + 2] * x2, p3 = s_val[k + 3] * x3;
                                acc = acc + p0;
                                acc = acc + p1;
                                acc = acc + p2;
                                acc = acc + p3;
                            }
                            for (; k < e; ++k) {
                                const double prod = s_val[k] * xat(grow + s_dict[s_code[k]]);
                                acc = acc + prod;
                            }
                        }
                        spmv_epilogue<EPI, RP>(a, r0 + tid, acc, coef, acc1, acc2);
                    }
                } else {
                    int np2 = 1;
                    while (np2 < nrows) np2 <<= 1;
                    const int tpr = min(64, kBlock / np2);
                    const int g = tid / tpr, l = tid - g * tpr;
                    double acc = 0.0;
                    if (g < nrows) {
                        const int e = rp[g + 1];
                        const int grow = grow0 + r0 + g;
                        for (int k = rp[g] + l; k < e; k += tpr) acc += s_val[k] * xat(grow + s_dict[s_code[k]]);
                    }
                    for (int off = tpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                    if (g < nrows && l == 0) spmv_epilogue<EPI, RP>(a, r0 + g, acc, coef, acc1, acc2);
                }
                __syncthreads();
                buf ^= 1;
            } else if (nrows > 1 || k1 - k0 <= (RP)kWaveRowMax) {
                const int wv = tid >> 6, ln = tid & 63;
                if (wv < nrows) {
                    const RP s0 = rowptr[r0 + wv], s1 = rowptr[r0 + wv + 1];
                    const int grow = grow0 + r0 + wv;
                    double acc = 0.0;
                    RP k = s0 + ln;
                    for (; k + 192 < s1; k += 256) {
                        const int c0 = __builtin_nontemporal_load(code + k);
                        const int c1 = __builtin_nontemporal_load(code + k + 64);
                        const int c2 = __builtin_nontemporal_load(code + k + 128);
                        const int c3 = __builtin_nontemporal_load(code + k + 192);
                        const double v0 = __builtin_nontemporal_load(val + k);
                        const double v1 = __builtin_nontemporal_load(val + k + 64);
                        const double v2 = __builtin_nontemporal_load(val + k + 128);
                        const double v3 = __builtin_nontemporal_load(val + k + 192);
                        acc += v0 * xg[grow + s_dict[c0]];
                        acc += v1 * xg[grow + s_dict[c1]];
                        acc += v2 * xg[grow + s_dict[c2]];
                        acc += v3 * xg[grow + s_dict[c3]];
                    }
                    for (; k < s1; k += 64) acc += val[k] * xg[grow + s_dict[code[k]]];
                    acc = wave_sum(acc);
                    if (ln == 0) spmv_epilogue<EPI, RP>(a, r0 + wv, acc, coef, acc1, acc2);
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
            } else {
                double acc = 0.0;
                const int grow = grow0 + r0;
                for (RP k = k0 + tid; k < k1; k += kBlock) acc += val[k] * xg[grow + s_dict[code[k]]];
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
                const double sum = block_sum(acc, s_red);
                if (tid == 0) spmv_epilogue<EPI, RP>(a, r0, sum, coef, acc1, acc2);
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

// Dual SpMV on the dictionary-compressed indices: K1 plus the true residual of the previous iterate from one
// sweep over val (8 B) + code (1 B); the column is decoded once per entry and used for both gathers.
template <typename RP, bool XW>
__global__ void __launch_bounds__(kBlock, 8) spmv_dual_diag_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr,
                                                               const uint8_t *__restrict__ code, const double *__restrict__ val,
                                                               const int32_t *__restrict__ rb, const RP *__restrict__ rbk,
                                                               const double *__restrict__ xg, const int32_t *__restrict__ rbw,
                                                               const int32_t *__restrict__ dict, int32_t ncols, int32_t grow0,
                                                               const double *__restrict__ x2, const double *__restrict__ b2,
                                                               int xcd_remap) {
    constexpr int EPI = EPI_DOT;
    __shared__ double s_val[kNnzPerRowBlock];
    __shared__ uint8_t s_code[kNnzPerRowBlock];
    __shared__ int s_rp[2][kMaxRowsPerRowBlock + 1];
    __shared__ double s_xw[XW ? kXWin : 1];
    __shared__ int s_dict[256];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    s_dict[tid] = dict[tid];
    __syncthreads();

    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(a.nrb, xcd_remap);
    int b = wk.first;
    if (b < wk.last) {
        int r0 = rb[b], r1 = rb[b + 1];
        RP k0 = rbk[b], k1 = rbk[b + 1];
        int32_t c[4];
        double v[4];
        double xw[XW ? kXWin / kBlock : 1];
        int wlo = XW ? rbw[b] : 0;
        RP rpn = 0;
#define SLA_ISSUE_LOADS(r0_, r1_, k0_, k1_, wlo_)                                           \
        if ((k1_) - (k0_) <= (RP)kNnzPerRowBlock) {                                          \
            const int cnt_ = (int)((k1_) - (k0_));                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                  \
                const int i = tid + j * kBlock;                                              \
                if (i < cnt_) {                                                              \
                    c[j] = __builtin_nontemporal_load(code + (k0_) + i);                     \
                    v[j] = __builtin_nontemporal_load(val + (k0_) + i);                      \
                }                                                                            \
            }                                                                                \
            if (tid < (r1_) - (r0_)) rpn = rowptr[(r0_) + tid];                              \
            if (XW) {                                                                        \
                _Pragma("unroll") for (int j = 0; j < kXWin / kBlock; ++j) {                 \
                    const int i = (wlo_) + tid + j * kBlock;                                 \
                    xw[j] = i < ncols ? xg[i] : 0.0;                                         \
                }                                                                            \
            }                                                                                \
        }
        SLA_ISSUE_LOADS(r0, r1, k0, k1, wlo)
        int nr0 = 0, nr1 = 0, nwlo = 0;
        RP nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) {
            if (XW) nwlo = rbw[b + wk.step];
            nr0 = rb[b + wk.step];
            nr1 = rb[b + wk.step + 1];
            nk0 = rbk[b + wk.step];
            nk1 = rbk[b + wk.step + 1];
        }
        // x[col] for col = global row + diagonal offset: LDS window first, L1/L2 otherwise
        auto xat = [&](int colg) -> double {
            if (XW) {
                const unsigned off = (unsigned)(colg - wlo);
                if (off < (unsigned)kXWin) return s_xw[off];
            }
            return xg[colg];
        };
        int buf = 0;
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fr1 = 0, fwlo = 0;
            RP fk0 = 0, fk1 = 0;
            const int nrows = r1 - r0;
            if (k1 - k0 <= (RP)kNnzPerRowBlock) {
                const int cnt = (int)(k1 - k0);
                int *rp = s_rp[buf];
                if (tid < nrows) rp[tid] = (int)(rpn - k0);
                if (tid == 0) rp[nrows] = cnt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < cnt) {
                        s_val[i] = v[j];
                        s_code[i] = (uint8_t)c[j];
                    }
                }
                if (XW) {
#pragma unroll
                    for (int j = 0; j < kXWin / kBlock; ++j) s_xw[tid + j * kBlock] = xw[j];
                }
                __syncthreads();
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
                if (nrows > 64 || cnt <= 8 * nrows) {
                    // one lane per row: decode, gather, multiply, add -- ascending, separately rounded
                    if (tid < nrows) {
                        const int s = rp[tid], e = rp[tid + 1];
                        const int grow = grow0 + r0 + tid;
                        double acc = 0.0, yb = 0.0;
                        {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
                            for (int k = s; k < e; ++k) {
                                const int cg = grow + s_dict[s_code[k]];
                                const double vv = s_val[k];
                                const double prod = vv * xat(cg);
                                const double prod2 = vv * x2[cg];
                                acc = acc + prod;
                                yb = yb + prod2;
                            }
                        }
                        const int row = r0 + tid;
                        a.y[row] = acc;
                        acc1 += acc * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                } else {
                    int np2 = 1;
                    while (np2 < nrows) np2 <<= 1;
                    const int tpr = min(64, kBlock / np2);
                    const int g = tid / tpr, l = tid - g * tpr;
                    double acc = 0.0, yb = 0.0;
                    if (g < nrows) {
                        const int e = rp[g + 1];
                        const int grow = grow0 + r0 + g;
                        for (int k = rp[g] + l; k < e; k += tpr) {
                            const int cg = grow + s_dict[s_code[k]];
                            acc += s_val[k] * xat(cg);
                            yb += s_val[k] * x2[cg];
                        }
                    }
                    for (int off = tpr >> 1; off > 0; off >>= 1) {
                        acc += __shfl_xor(acc, off, 64);
                        yb += __shfl_xor(yb, off, 64);
                    }
                    if (g < nrows && l == 0) {
                        const int row = r0 + g;
                        a.y[row] = acc;
                        acc1 += acc * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                }
                __syncthreads();
                buf ^= 1;
            } else if (nrows > 1 || k1 - k0 <= (RP)kWaveRowMax) {
                const int wv = tid >> 6, ln = tid & 63;
                if (wv < nrows) {
                    const RP s0 = rowptr[r0 + wv], s1 = rowptr[r0 + wv + 1];
                    const int grow = grow0 + r0 + wv;
                    double acc = 0.0, yb = 0.0;
                    for (RP k = s0 + ln; k < s1; k += 64) {
                        const int cg = grow + s_dict[code[k]];
                        const double vv = val[k];
                        acc += vv * xg[cg];
                        yb += vv * x2[cg];
                    }
                    acc = wave_sum(acc);
                    yb = wave_sum(yb);
                    if (ln == 0) {
                        const int row = r0 + wv;
                        a.y[row] = acc;
                        acc1 += acc * a.w[row];
                        const double t = yb - b2[row];
                        acc2 += t * t;
                    }
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
            } else {
                double acc = 0.0, yb = 0.0;
                const int grow = grow0 + r0;
                for (RP k = k0 + tid; k < k1; k += kBlock) {
                    const int cg = grow + s_dict[code[k]];
                    acc += val[k] * xg[cg];
                    yb += val[k] * x2[cg];
                }
                if (has_next) { SLA_ISSUE_LOADS(nr0, nr1, nk0, nk1, nwlo) }
                if (has_next2) {
                    if (XW) fwlo = rbw[bnn];
                    fr0 = rb[bnn];
                    fr1 = rb[bnn + 1];
                    fk0 = rbk[bnn];
                    fk1 = rbk[bnn + 1];
                }
                const double sum = block_sum(acc, s_red);
                const double sumb = block_sum(yb, s_red);
                if (tid == 0) {
                    a.y[r0] = sum;
                    acc1 += sum * a.w[r0];
                    const double t = sumb - b2[r0];
                    acc2 += t * t;
                }
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
#undef SLA_ISSUE_LOADS
    }
    const double s1 = block_sum(acc1, s_red);
    if (tid == 0) a.p1[blockIdx.x] = s1;
    const double s2 = block_sum(acc2, s_red);
    if (tid == 0) a.p2[blockIdx.x] = s2;
}


// ---------------------------------------------------------------------------------------------
// Value-indexed SpMV: one byte per stored entry
// ---------------------------------------------------------------------------------------------
// Stencil / constant-coefficient banded matrices repeat a handful of (diagonal offset, value) PAIRS:
// the 7-point Laplacian has 7.  When the lowering finds <= 256 distinct pairs (values compared by bit
// pattern, so the compression is lossless) and no row longer than kVdMaxRowNnz, it keeps one byte per entry
// that indexes a table of (offset, value).  This kernel then streams 1 B per entry + rowptr instead of
// 9-12 B: the matrix all but disappears from the HBM traffic and the sweep is bounded by the vectors
// (x, y and the fused-epilogue operand).  Fixed 256-row blocks, one lane per row: the row is the reference's
// ascending left fold with separately rounded multiply and add, bit for bit.  Code bytes are staged in LDS as
// whole dwords; the x window, the software pipeline (next block's loads fly during the row phase, block
// extents rowptr[256 b] come through scalar loads two blocks ahead) and the epilogues are those of
// spmv_diag_kernel.  DUAL adds the true residual of a second vector (linSolve0's fused check).
constexpr int kVdCodeDw = kBlock * 8;  // dwords of code staged per block (8 per lane)
template <int EPI, bool XW, bool DUAL>
__global__ void __launch_bounds__(kBlock, 8) spmv_vdict_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ rowptr,
                                                                const uint32_t *__restrict__ cw, const double *__restrict__ xg,
                                                                const int32_t *__restrict__ doff, const double *__restrict__ dval,
                                                                int32_t nblk, int32_t ncols, int32_t grow0,
                                                                const double *__restrict__ x2, const double *__restrict__ b2,
                                                                int xcd_remap) {
    __shared__ uint32_t s_cw[kVdCodeDw];
    __shared__ int s_rp[kBlock + 1];
    __shared__ double s_xw[XW ? kXWin : 1];
    __shared__ int s_doff[256];
    __shared__ double s_dval[256];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    s_doff[tid] = doff[tid];
    s_dval[tid] = dval[tid];
    __syncthreads();

    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(nblk, xcd_remap);
    const int rows = a.rows;
    const int wmax = max(0, ncols - kXWin);
    int b = wk.first;
    if (b < wk.last) {
        uint32_t c[8];
        double xw[XW ? kXWin / kBlock : 1];
        int rpn = 0;
        // extents of row block b_: rows [r0, r0 + nrows), entries [k0, k1)
#define SLA_VD_DESC(b_, r0_, nr_, k0_, k1_)          \
        r0_ = (b_) * kVdRows;                        \
        nr_ = min(kVdRows, rows - r0_);              \
        k0_ = rowptr[r0_];                           \
        k1_ = rowptr[r0_ + nr_];
#define SLA_VD_LOADS(r0_, nr_, k0_, k1_, wlo_)                                              \
        {                                                                                    \
            const int kb_ = (k0_) & ~3;                                                      \
            const int ndw_ = ((k1_) - kb_ + 3) >> 2;                                         \
            const uint32_t *src_ = cw + (kb_ >> 2);                                          \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                  \
                const int i = tid + j * kBlock;                                              \
                if (i < ndw_) c[j] = __builtin_nontemporal_load(src_ + i);                   \
            }                                                                                \
            if (tid < (nr_)) rpn = rowptr[(r0_) + tid];                                      \
            if (XW) {                                                                        \
                _Pragma("unroll") for (int j = 0; j < kXWin / kBlock; ++j) {                 \
                    const int i = (wlo_) + tid + j * kBlock;                                 \
                    xw[j] = i < ncols ? xg[i] : 0.0;                                         \
                }                                                                            \
            }                                                                                \
        }
#define SLA_VD_WLO(r0_) min(wmax, max(0, grow0 + (r0_) - kXWinHalo))
        int r0, nrows, k0, k1;
        SLA_VD_DESC(b, r0, nrows, k0, k1)
        int wlo = SLA_VD_WLO(r0);
        SLA_VD_LOADS(r0, nrows, k0, k1, wlo)
        int nr0 = 0, nnr = 0, nk0 = 0, nk1 = 0;
        if (b + wk.step < wk.last) { SLA_VD_DESC(b + wk.step, nr0, nnr, nk0, nk1) }
        auto xat = [&](int colg) -> double {
            if (XW) {
                const unsigned off = (unsigned)(colg - wlo);
                if (off < (unsigned)kXWin) return s_xw[off];
            }
            return xg[colg];
        };
        const uint8_t *cb = (const uint8_t *)s_cw;
        for (;;) {
            const int bn = b + wk.step, bnn = bn + wk.step;
            const bool has_next = bn < wk.last, has_next2 = bnn < wk.last;
            int fr0 = 0, fnr = 0, fk0 = 0, fk1 = 0;
            const int kb = k0 & ~3;
            {
                const int ndw = (k1 - kb + 3) >> 2;
                if (tid < nrows) s_rp[tid] = rpn - kb;
                if (tid == 0) s_rp[nrows] = k1 - kb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = tid + j * kBlock;
                    if (i < ndw) s_cw[i] = c[j];
                }
                if (XW) {
#pragma unroll
                    for (int j = 0; j < kXWin / kBlock; ++j) s_xw[tid + j * kBlock] = xw[j];
                }
            }
            __syncthreads();
            const int nwlo = has_next ? SLA_VD_WLO(nr0) : 0;
            if (has_next) { SLA_VD_LOADS(nr0, nnr, nk0, nk1, nwlo) }
            if (has_next2) { SLA_VD_DESC(bnn, fr0, fnr, fk0, fk1) }
            if (tid < nrows) {
                const int s = s_rp[tid], e = s_rp[tid + 1];
                const int grow = grow0 + r0 + tid;
                double acc = 0.0, yb = 0.0;
                {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
                    int k = s;
                    for (; k + 4 <= e; k += 4) {  // 4 gathers in flight, summed in order
                        const int c0 = cb[k], c1 = cb[k + 1], c2 = cb[k + 2], c3 = cb[k + 3];
                        const int g0 = grow + s_doff[c0], g1 = grow + s_doff[c1];
                        const int g2 = grow + s_doff[c2], g3 = grow + s_doff[c3];
                        const double v0 = s_dval[c0], v1 = s_dval[c1], v2 = s_dval[c2], v3 = s_dval[c3];
                        const double p0 = v0 * xat(g0), p1 = v1 * xat(g1);
                        const double p2 = v2 * xat(g2), p3 = v3 * xat(g3);
                        acc = acc + p0;
                        acc = acc + p1;
                        acc = acc + p2;
                        acc = acc + p3;
                        if constexpr (DUAL) {
                            const double q0 = v0 * x2[g0], q1 = v1 * x2[g1];
                            const double q2 = v2 * x2[g2], q3 = v3 * x2[g3];
                            yb = yb + q0;
                            yb = yb + q1;
                            yb = yb + q2;
                            yb = yb + q3;
                        }
                    }
                    for (; k < e; ++k) {
                        const int c0 = cb[k];
                        const int g0 = grow + s_doff[c0];
                        const double v0 = s_dval[c0];
                        const double p0 = v0 * xat(g0);
                        acc = acc + p0;
                        if constexpr (DUAL) {
                            const double q0 = v0 * x2[g0];
                            yb = yb + q0;
                        }
                    }
                }
                const int row = r0 + tid;
                if constexpr (DUAL) {
                    a.y[row] = acc;
                    acc1 += acc * a.w[row];
                    const double t = yb - b2[row];
                    acc2 += t * t;
                } else {
                    spmv_epilogue<EPI, int32_t>(a, row, acc, coef, acc1, acc2);
                }
            }
            if (!has_next) break;
            __syncthreads();
            b = bn;
            r0 = nr0;
            nrows = nnr;
            k0 = nk0;
            k1 = nk1;
            wlo = nwlo;
            nr0 = fr0;
            nnr = fnr;
            nk0 = fk0;
            nk1 = fk1;
        }
#undef SLA_VD_WLO
#undef SLA_VD_LOADS
#undef SLA_VD_DESC
    }
    if constexpr (DUAL || EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT ||
                  EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (tid == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (DUAL || EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (tid == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, tid);
}


// ---------------------------------------------------------------------------------------------
// Wave-sliced (offset, value) SpMV: the matrix lives in scalar registers
// ---------------------------------------------------------------------------------------------
// Same eligibility as the value-indexed form (<= 256 distinct (col - row, value) pairs), stored per SLICE of
// 128 consecutive rows -- one wavefront, TWO adjacent rows per lane -- as the sorted union of the pairs its
// rows use, each with two 64-bit lane masks (even rows / odd rows of the slice that hold it):
// {mask_even, mask_odd, value, offset} = 28 B per record, kept as four arrays.  A slice of the 7-point
// Laplacian is 7 records = 1.5 B per row, so the sweep moves the vectors and little else.
// Everything about an entry is wave-uniform, so the records come through SCALAR loads; a lane mask goes
// straight into EXEC (inverse ballot), the value is an SGPR operand of v_mul_f64 and the offset folds into
// the scalar base address of the gather.  What bounds such a kernel is the vector-memory instruction rate of
// the CU (a wave-wide 8-byte access costs as much as a 16-byte one: one-row-per-lane variants of this kernel
// and spmv_vdict_kernel both stalled at ~10 vector-memory instructions per 64 rows), hence the row pairs:
// every gather, the epilogue operand and the result are ONE 16-byte access per lane, 9 instructions per 128
// rows of the 7-point stencil.  The union is sorted by (offset, value bits) and a row holds at most one entry
// per offset, so every row still adds its products in ascending column order with separately rounded
// multiply and add: the reference's left fold, bit for bit.  Up to 8 gathers are in flight per lane.
typedef unsigned long long wd_u64x8 __attribute__((ext_vector_type(8), aligned(8)));
typedef double wd_f64x8 __attribute__((ext_vector_type(8), aligned(8)));
typedef int wd_i32x8 __attribute__((ext_vector_type(8), aligned(4)));

#if defined(SLA_WD_TRACE)
__device__ unsigned long long wd_trace[64 * 16 * 4];
extern "C" int sla_debug_wd_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(wd_trace), sizeof(wd_trace));
}
#define WD_STAMP(i)                                                                                            \
    if (trace_on && iter < 16) {                                                                               \
        const unsigned long long t_ = __builtin_readcyclecounter();                                            \
        if (lane == 0) wd_trace[((blockIdx.x >> 5) * 16 + iter) * 4 + (i)] = t_;                               \
        if ((i) == 3) ++iter;                                                                                  \
    }
#else
#define WD_STAMP(i)
#endif
struct WdRec {  // lane k < 8 holds record k of the slice's first chunk (both masks 0: no such record)
    unsigned long long me, mo;
    double v;
    int o;
};
__device__ __forceinline__ unsigned long long wd_lane_u64(unsigned long long x, int k) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)x, k), hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), k);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double wd_lane_f64(double x, int k) {
    return __longlong_as_double((long long)wd_lane_u64((unsigned long long)__double_as_longlong(x), k));
}

// Latency: a wavefront's chain per slice would be descriptor -> records -> gathers -> store, three dependent
// trips to memory.  The descriptor (scalar) is fetched two slices ahead and the records one slice ahead --
// by lanes 0..7, one record each, through the in-order vector queue; v_readlane moves a field to SGPRs when
// it is used -- so a wavefront waits on memory once per slice.
// Six workgroups per CU, not eight: 85 VGPRs instead of 64 end the spills of the fused epilogues, and the CU's L1
// serves more of the overlapping gathers with fewer wavefronts streaming through it (measured same-box:
// 8 / 7 / 6 / 5 / 4 per CU = 2890 / 3110 / 3170 / 3140 / 2940 BiCGSTAB it/s).
// VV: variable coefficients -- a record carries no scalar value but a block of 128 values laid out like the slice's
// rows (wvblk), fetched with one more 16-byte load per lane; everything else is shared.
// (VV with the four-sum epilogue: one workgroup per CU less -- 108 VGPRs and no scratch instead of 96 + 44 B of spills per lane)
template <int EPI, bool VV>
__global__ void __launch_bounds__(kBlock, VV ? (EPI == EPI_DOT4 ? kWdBlocksPerCuVV - 1 : kWdBlocksPerCuVV) : kWdBlocksPerCu) spmv_wdia_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ sptr,
                                                               const unsigned long long *__restrict__ wme,
                                                               const unsigned long long *__restrict__ wmo,
                                                               const double *__restrict__ wval, const int32_t *__restrict__ woff,
                                                               const double *__restrict__ wvblk, const double *__restrict__ xg, int32_t nblk, int32_t nslices,
                                                               int32_t grow0, int32_t xlen, const int32_t *__restrict__ sched, int xcd_remap, int stream_nt) {
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef;
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    const bool w_nt = stream_nt && a.w != xg + grow0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(nblk, xcd_remap);
    constexpr bool kUsesW = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT;
    constexpr bool kUsesZ = EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM || EPI == EPI_DOT4;   // (EPI_DOT4: read-only)
    // slice descriptor of workgroup step b (wave-uniform): first record, record count (0: nothing to do)
    // `sched` (optional) is the order in which the 512-row steps are visited (see csr_upload: steps a far diagonal
    // apart are made neighbours in time so that the three planes a 3-D stencil row touches meet in the L2)
    // cnt < 0: no slice (past the end); cnt == 0: a slice whose rows hold no entry -- its rows still run the epilogue
    // (r = b - A x, ||A x - b||, z -= alpha A p ... are defined on empty rows too)
    auto load_desc = [&](int b, int &blk, int &e0, int &cnt) {
        e0 = 0;
        cnt = -1;
        blk = 0;
        if (b < wk.last) {
            blk = sched ? sched[b] : b;
            const int s = blk * 4 + wave;
            if (s < nslices) {
                e0 = sptr[s];
                cnt = sptr[s + 1] - e0;
            }
        }
    };
    auto load_rec = [&](int e0, int cnt, WdRec &r) {
        r.me = 0ull;
        r.mo = 0ull;
        r.v = 0.0;
        r.o = 0;
        if (lane < 8 && lane < cnt) {
            r.me = wme[e0 + lane];
            r.mo = wmo[e0 + lane];
            if (!VV) r.v = wval[e0 + lane];
            r.o = woff[e0 + lane];
        }
    };
    // one slice between "gathers issued" and "folded": the gathered row pairs, the epilogue operands and the record
    // fields the fold needs
    struct Stage {
        wd_f64x2 xv[8];
        wd_f64x2 vv[VV ? 8 : 1];    // VV: the row pair's two values of record k
        wd_f64x2 wv, zv;
        unsigned long long me, mo;  // lane k: masks of record k
        double v;                   // lane k: value of record k
        int blk, e0, cnt;
    };
    // gathers of the 8 records held by lanes 0..7 of `r` (first record e0) for the row pair starting at `row`
    auto gather8 = [&](const WdRec &r, int e0, int row, wd_f64x2 *xv, wd_f64x2 *vv) {
        // byte offset of x[global row]; a record's diagonal offset moves the scalar base instead
        const uint32_t g8 = (uint32_t)(grow0 + row) * 8u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long mb = wd_lane_u64(r.me | r.mo, k);  // lanes with the entry in either row (0: no record k)
            const int ok = __builtin_amdgcn_readlane(r.o, k);
            // one 16-byte gather per lane.  When only one row of the pair holds the entry the other half is
            // loaded and ignored; at the two ends of x it lies in the guard slack (guard_malloc).
            if (__builtin_amdgcn_inverse_ballot_w64(mb)) {
                xv[k] = *(const wd_f64x2u *)((const char *)(xg + ok) + g8);
                if (VV) vv[k] = *(const wd_f64x2 *)(wvblk + ((size_t)(e0 + k) << 7) + 2 * lane);
            }
        }
    };
    // the products of up to 8 records folded into the row pair's sums, in record (= ascending column) order
    auto fold8 = [&](unsigned long long rme, unsigned long long rmo, double rv, int nrec, const wd_f64x2 *xv, const wd_f64x2 *vv,
                     double &ya, double &yb) {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= nrec) break;  // wave-uniform
            const unsigned long long me = wd_lane_u64(rme, k), mo = wd_lane_u64(rmo, k);
            // EXEC = the even rows that hold the entry, then the odd rows; v_mul_f64 then v_add_f64 (two
            // roundings).  All 64 lanes are active here (wave-uniform control flow only): EXEC goes back to -1.
            double p;
            if constexpr (VV) {
                asm volatile(
                    "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[va], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                    "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[vb], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                    "s_mov_b64 exec, -1"
                    : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(p)
                    : [me] "s"(me), [mo] "s"(mo), [va] "v"(vv[k].x), [vb] "v"(vv[k].y), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
            } else {
                const double vk = wd_lane_f64(rv, k);
                asm volatile(
                    "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[v], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                    "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[v], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                    "s_mov_b64 exec, -1"
                    : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(p)
                    : [me] "s"(me), [mo] "s"(mo), [v] "s"(vk), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
            }
        }
    };
    // issue the epilogue-operand loads and the gathers of the slice described by (blk, e0, cnt, r)
    auto issue = [&](Stage &st, int blk, int e0, int cnt, const WdRec &r) {
        st.blk = blk;
        st.e0 = e0;
        st.cnt = cnt;
        st.me = r.me;
        st.mo = r.mo;
        st.v = r.v;
        st.wv = wd_f64x2{0.0, 0.0};
        st.zv = wd_f64x2{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // "defined" without an instruction: a lane that does not load holds garbage,
            asm("" : "=v"(st.xv[k]));  // which EXEC never lets the fold use
            if (VV) asm("" : "=v"(st.vv[k]));
        }
        if (cnt < 0) return;
        const int row = (blk * 4 + wave) * 128 + 2 * lane;  // this lane's rows: row, row + 1
        const bool va = row < a.rows, vb = row + 1 < a.rows;
        if (vb) {
            // epilogue operands are single-use streams: past the caches when the vectors overflow them anyway (see
            // vec_stream_nt) -- except an operand that IS the gathered vector (K3: w = s = x), which must stay
            if constexpr (kUsesW) {
                if (EPI != EPI_AXPY_DOT || a.w)
                    st.wv = w_nt ? __builtin_nontemporal_load((const wd_f64x2 *)(a.w + row)) : *(const wd_f64x2 *)(a.w + row);
            }
            if constexpr (kUsesZ)
                st.zv = stream_nt ? __builtin_nontemporal_load((const wd_f64x2 *)(a.z + row)) : *(const wd_f64x2 *)(a.z + row);
        } else if (va) {
            if constexpr (kUsesW) { if (EPI != EPI_AXPY_DOT || a.w) st.wv.x = a.w[row]; }
            if constexpr (kUsesZ) st.zv.x = a.z[row];
        }
        gather8(r, e0, row, st.xv, st.vv);
    };
    // fold the slice's products row by row and run the epilogue
    auto fold = [&](const Stage &st) {
        if (st.cnt < 0) return;
        const int row = (st.blk * 4 + wave) * 128 + 2 * lane;
        const bool va = row < a.rows, vb = row + 1 < a.rows;
        double ya = 0.0, yb = 0.0;
        fold8(st.me, st.mo, st.v, st.cnt, st.xv, st.vv, ya, yb);
        // slices with more than 8 records (27-point stencils, wide bands): further chunks of 8, fetched, gathered and
        // folded one after the other (no pipelining across chunks)
        for (int c0 = 8; c0 < st.cnt; c0 += 8) {
            WdRec rr;
            load_rec(st.e0 + c0, st.cnt - c0, rr);
            wd_f64x2 xt[8], vt[VV ? 8 : 1];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm("" : "=v"(xt[k]));
                if (VV) asm("" : "=v"(vt[k]));
            }
            gather8(rr, st.e0 + c0, row, xt, vt);
            fold8(rr.me, rr.mo, rr.v, st.cnt - c0, xt, vt, ya, yb);
        }
        if (va) wd_epilogue<EPI>(a, row, vb, ya, yb, st.wv, st.zv, coef, acc1, acc2);
    };
    int b = wk.first;
#if defined(SLA_WD_TRACE)
    const bool trace_on = (blockIdx.x & 31) == 0 && wave == 0 && EPI == EPI_DOT;
    int iter = 0;
#endif
    if constexpr (kWdGatherStages == 2) {
        // descriptors three slices ahead, records two ahead, gathers one ahead: while slice i is folded the gathers of
        // slice i + 1 and the records of slice i + 2 are in flight
        int blk1, e01, cnt1, blk2, e02, cnt2, blk3, e03, cnt3;
        WdRec r1, r2;
        Stage sa, sb;
        {
            int blk0, e00, cnt0;
            WdRec r0;
            load_desc(b, blk0, e00, cnt0);
            load_desc(b + wk.step, blk1, e01, cnt1);
            load_desc(b + 2 * wk.step, blk2, e02, cnt2);
            load_rec(e00, cnt0, r0);
            load_rec(e01, cnt1, r1);
            issue(sa, blk0, e00, cnt0, r0);
        }
#define SLA_WD_STEP(cur, nxt)                         \
        {                                             \
            WD_STAMP(0)                               \
            load_rec(e02, cnt2, r2);                  \
            load_desc(b + 3 * wk.step, blk3, e03, cnt3); \
            issue(nxt, blk1, e01, cnt1, r1);          \
            WD_STAMP(1)                               \
            fold(cur);                                \
            WD_STAMP(3)                               \
            b += wk.step;                             \
            r1 = r2;                                  \
            blk1 = blk2; e01 = e02; cnt1 = cnt2;      \
            blk2 = blk3; e02 = e03; cnt2 = cnt3;      \
        }
        while (b < wk.last) {
            SLA_WD_STEP(sa, sb)
            if (b >= wk.last) break;
            SLA_WD_STEP(sb, sa)
        }
#undef SLA_WD_STEP
    } else {
        int blk_c, e0_c, cnt_c, blk_n, e0_n, cnt_n;
        WdRec rc, rn;
        Stage st;
        load_desc(b, blk_c, e0_c, cnt_c);
        load_desc(b + wk.step, blk_n, e0_n, cnt_n);
        load_rec(e0_c, cnt_c, rc);
        for (; b < wk.last; b += wk.step) {
            int blk_f, e0_f, cnt_f;
            WD_STAMP(0)
            load_rec(e0_n, cnt_n, rn);               // next slice's records: in flight behind this slice's gathers
            load_desc(b + 2 * wk.step, blk_f, e0_f, cnt_f);
            issue(st, blk_c, e0_c, cnt_c, rc);
            WD_STAMP(1)
            fold(st);
            WD_STAMP(3)
            blk_c = blk_n;
            e0_c = e0_n;
            cnt_c = cnt_n;
            rc = rn;
            blk_n = blk_f;
            e0_n = e0_f;
            cnt_n = cnt_f;
        }
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM) {
        const double s1 = block_sum(acc1, s_red);
        if (tid == 0) a.p1[blockIdx.x] = s1;
    }
    if constexpr (EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        const double s2 = block_sum(acc2, s_red);
        if (tid == 0) a.p2[blockIdx.x] = s2;
    }
    spmv_extra_partials<EPI>(a, s_red, tid);
}

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
template <typename RP, int L, int R, int J>
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
        const RP *ps = pp + (int64_t)p * rows, *pe = ps + rows;
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
                k[r] = (has ? ps[i] : 0) + gl;
                e[r] = has ? pe[i] : 0;
                acc[r] = 0.0;
            }
            bool more = true;
            while (more) {
                int32_t cj[R][J];
                double vj[R][J];
                // all loads of the round in flight before the first use
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

// do the solver's vectors (7 of n entries for BiCGSTAB) overflow the memory-side cache?  Then stream them past it.
static bool vec_stream_nt(const sla_ctx *c, int64_t n) {
    return c->vec_nt < 0 ? 7 * 8 * n > c->mall_bytes : c->vec_nt != 0;
}

// row-sharded overlap (sla_api.cpp: spmv_exchanged): interior / boundary launches of the wave-sliced forms
bool overlap_split(const sla_csr *A) {
    return A->ov_nint > 0 && A->ov_nbnd > 0 && A->ctx->overlap >= 0 && A->ctx->collectives && A->use_wdia && wd_on(A) && A->ctx->spmv_algo == 0 && !A->rp64;
}
int overlap_grid(const sla_csr *A, int part) {
    // the two launches write their fused partial sums into consecutive slots of ONE kMaxParts-slot array (interior first):
    // the interior grid leaves room for the boundary launch's (a 299-CU part at 6 workgroups per CU, or SLA_WD_GRID=2048,
    // would otherwise push the boundary partials into the next slot array)
    const int gb = std::max(1, std::min<int>(A->ov_nbnd, 256));
    if (part != 1) return gb;
    const int cap = std::min(A->wd_vv ? A->ctx->wd_grid_max_vv : wd_lds_on(A) ? wd_lds_grid(A) : A->ctx->wd_grid_max, kMaxParts - gb);
    return std::max(1, std::min<int>(A->ov_nint, cap));
}

int spmv_grid(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    // with column panels the fused partials are written by the LAST panel pass: its grid is the one that counts
    if (tiles_on(A)) return tiles_grid(A);
    if (!A->panels.empty() && c->panels && c->spmv_algo == 0) return spmv_grid(A->panels.back());
    int64_t g;
    if (c->spmv_algo == 1 || (A->use_lpanel && c->lpanel)) g = (A->rows + kBlock - 1) / kBlock;   // (lpanel: its finish kernel)
    else if (A->use_wdia && wd_on(A)) g = std::min<int64_t>(A->nblk_wd, A->wd_vv ? c->wd_grid_max_vv : wd_lds_on(A) ? wd_lds_grid(A) : c->wd_grid_max);
    else if (A->use_vdict && c->vdict) g = A->nblk_vd;
    else g = A->nrb;
    if (g < 1) g = 1;
    if (g > c->spmv_grid_max) g = c->spmv_grid_max;
    return (int)g;
}

template <int EPI, typename RP>
static int launch_spmv_t(const sla_csr *A, const SpmvLaunch &l);

// Column-panel SpMV for matrices whose gathers do not fit the XCD-private L2: the matrix is stored a second
// time panel-major (panel p holds the entries with column in [p W, (p+1) W), W * 8 B <= ~3 MB), and y = A x
// is evaluated as P passes y += A_p x in ascending panel order, so every gather of a pass hits a window of
// x that stays L2-resident.  Each pass continues the row's running sum (yinit), i.e. the per-row order is
// still one ascending left fold; the fused epilogue runs in the last pass only.
template <int EPI, typename RP>
static int launch_spmv_panels(const sla_csr *A, const SpmvLaunch &l) {
    const size_t P = A->panels.size();
    ProfScope prof(A->ctx, l.kernel_id);  // one timed interval for the whole panel sequence
    double *ytmp = l.y;
    if (!ytmp) ytmp = A->d_panel_y;  // epilogues that never store y (EPI_RES, EPI_AXPY_DOT, ...) still need the running sum
    for (size_t p = 0; p < P; ++p) {
        const bool last = p + 1 == P;
        SpmvLaunch lp = l;
        lp.yinit = p == 0 ? nullptr : ytmp;
        lp.kernel_id = -2;               // never matches: the enclosing scope does the timing
        int rc;
        // (a view holds only its panel's entries: it may use 32-bit row pointers where the parent needs 64-bit ones)
        const sla_csr *V = A->panels[p];
        if (!last) {
            lp.epi = EPI_NONE;
            lp.y = ytmp;
            lp.pres = nullptr;            // prologue checks / step bookkeeping happen once, in the last pass
            lp.step_begin = 0;
            lp.pa = nullptr;
            rc = V->rp64 ? launch_spmv_t<EPI_NONE, int64_t>(V, lp) : launch_spmv_t<EPI_NONE, int32_t>(V, lp);
        } else {
            rc = V->rp64 ? launch_spmv_t<EPI, int64_t>(V, lp) : launch_spmv_t<EPI, int32_t>(V, lp);
        }
        if (rc != SLA_OK) return rc;
    }
    return SLA_OK;
}

template <int EPI, typename RP>
static int launch_spmv_t(const sla_csr *A, const SpmvLaunch &l) {
    sla_ctx *c = A->ctx;
    if (!A->panels.empty() && c->panels && c->spmv_algo == 0 && !l.x2 && !l.in_panel) {
        SpmvLaunch lp = l;
        lp.in_panel = 1;
        return launch_spmv_panels<EPI, RP>(A, lp);
    }
    SpmvArgs<RP> a;
    a.rowptr = (const RP *)A->d_rowptr;
    a.col = A->d_col;
    a.val = A->d_val;
    a.x = l.x;
    a.y = l.y;
    a.rb = A->d_rb;
    a.rbk = (const RP *)A->d_rbk;
    a.nrb = A->nrb;
    a.rows = (int32_t)A->rows;
    a.w = l.w;
    a.z = l.z;
    a.p1 = l.p1;
    a.p2 = l.p2;
    a.p3 = l.p3;
    a.p4 = l.p4;
    a.acc3 = 0.0;
    a.acc4 = 0.0;
    a.sc = l.sc;
    a.pres = l.pres;
    a.npres = l.npres;
    a.pres_stride = l.pres_stride;
    a.pa = l.pa;
    a.pb = l.pb;
    a.npa = l.npa;
    a.pa_stride = l.pa_stride;
    a.step_begin = l.step_begin;
    a.yinit = l.yinit;
    const int grid = spmv_grid(A);
    ProfScope prof(c, l.kernel_id);
    if (A->use_lpanel && c->lpanel && c->spmv_algo == 0 && !l.x2 && !l.yinit) {
        // lane-group shape by mean segment length (A->lp_cfg, set at lowering): 64 lanes x 2 segments x 4 loads for long
        // segments, narrower groups with more segments per wavefront for short ones
#define SLA_LP_LAUNCH(CFG, L_, R_, J_)                                                                                         \
        case CFG: {                                                                                                             \
            const int attr_bit = 1 << (2 * CFG + (std::is_same<RP, int32_t>::value ? 0 : 1));                                   \
            if (!(c->lp_attr & attr_bit)) {   /* per context = per device: 128 KiB of dynamic LDS */                            \
                SLA_HIP_TRY(hipFuncSetAttribute((const void *)spmv_lpanel_kernel<RP, L_, R_, J_>,                               \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLpW * sizeof(double))));     \
                c->lp_attr |= attr_bit;                                                                                         \
            }                                                                                                                   \
            hipLaunchKernelGGL((spmv_lpanel_kernel<RP, L_, R_, J_>), dim3(A->lp_G), dim3(kLpBlock), kLpW * sizeof(double),      \
                               stream_of(c), (const RP *)A->d_lpp, a.col, a.val, a.x, A->d_lpy, A->d_lpt, a.rows, (int)A->n,       \
                               A->lp_W, A->lp_chunk, A->lp_C, A->lp_col_lo, A->lp_col_hi, (const SolverScalars *)a.sc);         \
        } break;
        switch (A->lp_cfg) {
            SLA_LP_LAUNCH(1, 32, 4, 2)
            SLA_LP_LAUNCH(2, 16, 4, 2)
            SLA_LP_LAUNCH(3, 8, 4, 2)
            default:
            SLA_LP_LAUNCH(0, 64, kLpRowsInFlight, 4)
        }
#undef SLA_LP_LAUNCH
        SLA_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL((lpanel_finish_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_lpy, A->lp_P);
        SLA_HIP_TRY(hipGetLastError());
        return SLA_OK;
    }
    if (A->use_wdia && wd_on(A) && c->spmv_algo == 0 && !l.x2) {
        if constexpr (std::is_same<RP, int32_t>::value) {
            const int32_t *sched = c->wd_tile != 0 ? A->d_wsched : nullptr;
            int32_t nblk_wd = A->nblk_wd;
            int grid = ::sla::spmv_grid(A);
            if (l.part == 1) { sched = A->d_ov_int; nblk_wd = A->ov_nint; grid = overlap_grid(A, 1); }
            else if (l.part == 2) { sched = A->d_ov_bnd; nblk_wd = A->ov_nbnd; grid = overlap_grid(A, 2); }
            if (wd_lds_on(A))
                return launch_wdia_lds(A, l.epi, a, sched, nblk_wd, grid, (vec_stream_nt(c, A->rows) ? 1 : 0) | (c->wd_nt_store ? 2 : 0));
            if (A->wd_vv)
                hipLaunchKernelGGL((spmv_wdia_kernel<EPI, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                                   A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                                   sched, c->xcd_remap, vec_stream_nt(c, A->rows) ? 1 : 0);
            else
                hipLaunchKernelGGL((spmv_wdia_kernel<EPI, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                                   A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                                   sched, c->xcd_remap, vec_stream_nt(c, A->rows) ? 1 : 0);
            SLA_HIP_TRY(hipGetLastError());
            return SLA_OK;
        }
    }
    if (A->use_vdict && c->vdict && c->spmv_algo == 0) {
        if constexpr (std::is_same<RP, int32_t>::value) {
            const bool xw = A->use_xwin && c->xwin;
#define SLA_VD_LAUNCH(E, XW_, DUAL_)                                                                                        \
            hipLaunchKernelGGL((spmv_vdict_kernel<E, XW_, DUAL_>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr,        \
                               (const uint32_t *)A->d_vcode, a.x, A->d_vdoff, A->d_vdval, A->nblk_vd, (int32_t)A->n,          \
                               (int32_t)A->row_begin, l.x2, l.b2, c->xcd_remap)
            if (l.x2) {
                if constexpr (EPI == EPI_DOT) {
                    if (xw) SLA_VD_LAUNCH(EPI_DOT, true, true);
                    else SLA_VD_LAUNCH(EPI_DOT, false, true);
                } else {
                    return fail(SLA_ERR_INVALID, "dual SpMV is only defined for the K1 epilogue");
                }
            } else if (xw) SLA_VD_LAUNCH(EPI, true, false);
            else SLA_VD_LAUNCH(EPI, false, false);
#undef SLA_VD_LAUNCH
            SLA_HIP_TRY(hipGetLastError());
            return SLA_OK;
        }
    }
    if (l.x2) {
        if constexpr (EPI == EPI_DOT) {
            if (A->use_diag && c->diag) {
                if (A->use_xwin && c->xwin)
                    hipLaunchKernelGGL((spmv_dual_diag_kernel<RP, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val,
                                       a.rb, a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, l.x2, l.b2, c->xcd_remap);
                else
                    hipLaunchKernelGGL((spmv_dual_diag_kernel<RP, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val,
                                       a.rb, a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, l.x2, l.b2, c->xcd_remap);
            } else
            hipLaunchKernelGGL((spmv_dual_kernel<RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk, a.x,
                               l.x2, l.b2, c->xcd_remap);
        } else {
            return fail(SLA_ERR_INVALID, "dual SpMV is only defined for the K1 epilogue");
        }
    } else if (c->spmv_algo == 1)
        hipLaunchKernelGGL((spmv_scalar_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, c->xcd_remap);
    else if (A->use_diag && c->diag) {
        if (A->use_xwin && c->xwin)
            hipLaunchKernelGGL((spmv_diag_kernel<EPI, RP, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val, a.rb,
                               a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, c->xcd_remap);
        else
            hipLaunchKernelGGL((spmv_diag_kernel<EPI, RP, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val, a.rb,
                               a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, c->xcd_remap);
    } else {
        if (A->use_xwin && c->xwin)
            hipLaunchKernelGGL((spmv_xwin_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk,
                               a.x, A->d_rbw, (int32_t)A->n, c->xcd_remap);
        else
            hipLaunchKernelGGL((spmv_stream_kernel<EPI, RP>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.rb, a.rbk,
                               a.x, c->xcd_remap);
    }
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

template <typename RP>
static int launch_spmv_rp(const sla_csr *A, const SpmvLaunch &l) {
    switch (l.epi) {
        case EPI_NONE: return launch_spmv_t<EPI_NONE, RP>(A, l);
        case EPI_DOT: return launch_spmv_t<EPI_DOT, RP>(A, l);
        case EPI_DOT2: return launch_spmv_t<EPI_DOT2, RP>(A, l);
        case EPI_DOT4: return launch_spmv_t<EPI_DOT4, RP>(A, l);
        case EPI_RES: return launch_spmv_t<EPI_RES, RP>(A, l);
        case EPI_AXPY_DOT: return launch_spmv_t<EPI_AXPY_DOT, RP>(A, l);
        case EPI_XPBY_NRM: return launch_spmv_t<EPI_XPBY_NRM, RP>(A, l);
        case EPI_SUB: return launch_spmv_t<EPI_SUB, RP>(A, l);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv: unknown epilogue");
}

int launch_spmv(const sla_csr *A, const SpmvLaunch &l) {
    if (tiles_on(A) && !l.x2 && !l.yinit) return launch_spmv_tiles(A, l);
    return A->rp64 ? launch_spmv_rp<int64_t>(A, l) : launch_spmv_rp<int32_t>(A, l);
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

#define SLA_VEC_LOOP_BEGIN(n)                                                        \
    const int64_t _n2 = (n) >> 1;                                                    \
    const int64_t _gs = (int64_t)gridDim.x * kBlock;                                 \
    for (int64_t i2 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i2 < _n2; i2 += _gs) {
#define SLA_VEC_LOOP_END }
#define SLA_HAS_TAIL(n) (((n) & 1) && blockIdx.x == 0 && threadIdx.x == 0)

typedef double sla_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_nt(const double *p, int64_t i2) {
    const sla_d2 t = __builtin_nontemporal_load(reinterpret_cast<const sla_d2 *>(p) + i2);
    return make_double2(t.x, t.y);
}
__device__ __forceinline__ double2 ld2_t(const double *p, int64_t i2) { return reinterpret_cast<const double2 *>(p)[i2]; }
__device__ __forceinline__ void st2_nt(double *p, int64_t i2, double2 v) {
    __builtin_nontemporal_store(sla_d2{v.x, v.y}, reinterpret_cast<sla_d2 *>(p) + i2);
}
__device__ __forceinline__ void st2_t(double *p, int64_t i2, double2 v) { reinterpret_cast<double2 *>(p)[i2] = v; }
#define ld2 ld2_t
#define st2 st2_t
// Non-temporal loads in the BiCGSTAB vector kernels when the solver's vectors cannot stay in the 256 MB memory-side
// cache anyway (template NT, chosen per launch by vec_stream_nt): +12 % iterations/s at 10 M rows (7 x 80 MB), -4...-6 %
// at 1-2 M rows where the whole working set is cache-resident and the hint only loses hits.
template <bool NT>
__device__ __forceinline__ double2 ld2s(const double *p, int64_t i2) { return NT ? ld2_nt(p, i2) : ld2_t(p, i2); }

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
                                                          const double *ap, double *s) {
    __shared__ double s_red[4];
    if (sc->done) return;
    // dual-SpMV flow: K1 of THIS step also evaluated the previous step's true residual; test it here
    if (res.p && residual_converged(sc, res.p, res.n, res.stride, s_red)) return;
    if (count_iter && blockIdx.x == 0 && threadIdx.x == 0) sc->iters += 1;
    const double alpha = sc->rho2[par] / reduce_parts(apr.p, apr.n, apr.stride, s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 a = ld2s<NT>(r, i2), b = ld2s<NT>(ap, i2);
        st2(s, i2, make_double2(a.x - alpha * b.x, a.y - alpha * b.y));
    SLA_VEC_LOOP_END
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
template <bool NT>
__global__ void __launch_bounds__(kBlock) bicg_k45_kernel(int64_t n, SolverScalars *sc, Parts ass, Parts asas, Parts tr0, Parts sr0,
                                                           int par, const double *s, const double *as, const double *ap, double *x,
                                                           double *r, double *p) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double num = reduce_parts(ass.p, ass.n, ass.stride, s_red);
    const double den = reduce_parts(asas.p, asas.n, asas.stride, s_red);
    const double t0 = reduce_parts(tr0.p, tr0.n, tr0.stride, s_red);
    const double s0 = reduce_parts(sr0.p, sr0.n, sr0.stride, s_red);
    const double omega = num / den, alpha = sc->alpha;
    const double rn = s0 - omega * t0;                       // = r_{j+1} . r0hat
    const double beta = rn / sc->rho2[par] * alpha / omega;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->omega = omega;
        sc->beta = beta;
        sc->rho2[par ^ 1] = rn;
    }
    SLA_VEC_LOOP_BEGIN(n)
        const double2 sv = ld2s<NT>(s, i2), av = ld2s<NT>(as, i2), vv = ld2s<NT>(ap, i2);
        double2 pv = ld2s<NT>(p, i2), xv = ld2s<NT>(x, i2);
        xv.x = (xv.x + alpha * pv.x) + omega * sv.x;
        xv.y = (xv.y + alpha * pv.y) + omega * sv.y;
        if (NT) st2_nt(x, i2, xv);  // nobody reads x before the next step's sweep
        else st2(x, i2, xv);
        const double2 rv = make_double2(sv.x - omega * av.x, sv.y - omega * av.y);
        if (NT) st2_nt(r, i2, rv);  // r is next read by K2, after K1 has streamed 250 MB: only p (K1's x) should stay cached
        else st2(r, i2, rv);
        pv.x = rv.x + beta * (pv.x - omega * vv.x);
        pv.y = rv.y + beta * (pv.y - omega * vv.y);
        st2(p, i2, pv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        x[i] = (x[i] + alpha * p[i]) + omega * s[i];
        const double rv = s[i] - omega * as[i];
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
    ProfScope prof(c, SLA_KERNEL_BICG_K2);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(bicg_k2_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, r, ap, s);
    else
        hipLaunchKernelGGL(bicg_k2_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, r, ap, s);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, const double *p, const double *s,
                   const double *as, const double *r0hat, double *x, double *r, double *prho) {
    ProfScope prof(c, SLA_KERNEL_BICG_K4);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(bicg_k4_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, p, s, as, r0hat, x, r, prho);
    else
        hipLaunchKernelGGL(bicg_k4_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, p, s, as, r0hat, x, r, prho);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k5(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *ap, double *p) {
    ProfScope prof(c, SLA_KERNEL_BICG_K5);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(bicg_k5_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, ap, p);
    else
        hipLaunchKernelGGL(bicg_k5_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, ap, p);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_bicg_k45(sla_ctx *c, int64_t n, SolverScalars *sc, Parts ass, Parts asas, Parts tr0, Parts sr0, int par, const double *s,
                    const double *as, const double *ap, double *x, double *r, double *p) {
    ProfScope prof(c, SLA_KERNEL_BICG_K45);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(bicg_k45_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p);
    else
        hipLaunchKernelGGL(bicg_k45_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, ass, asas, tr0, sr0, par, s, as, ap, x, r, p);
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
                                                         const double *aap, double *q, double *uq, double *x) {
    __shared__ double s_red[4];
    if (sc->done) return;
    if (res.p && residual_converged(sc, res.p, res.n, res.stride, s_red)) return;
    if (count_iter && blockIdx.x == 0 && threadIdx.x == 0) sc->iters += 1;
    const double alpha = sc->rho2[par] / reduce_parts(apr.p, apr.n, apr.stride, s_red);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = alpha;
    SLA_VEC_LOOP_BEGIN(n)
        const double2 uv = ld2s<NT>(u, i2), av = ld2s<NT>(aap, i2);
        double2 xv = ld2s<NT>(x, i2);
        const double2 qv = make_double2(uv.x - alpha * av.x, uv.y - alpha * av.y);
        const double2 sv = make_double2(uv.x + qv.x, uv.y + qv.y);
        xv.x += alpha * sv.x;
        xv.y += alpha * sv.y;
        st2(q, i2, qv);
        st2(uq, i2, sv);
        if (NT) st2_nt(x, i2, xv);  // (as in K4: x is not read again before the next step)
        else st2(x, i2, xv);
    SLA_VEC_LOOP_END
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
                                                         const double *r, const double *q, double *u, double *p) {
    __shared__ double s_red[4];
    if (sc->done) return;
    const double rn = reduce_parts(rhonew.p, rhonew.n, rhonew.stride, s_red);
    const double beta = rn / sc->rho2[par];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->rho2[par ^ 1] = rn; }
    SLA_VEC_LOOP_BEGIN(n)
        const double2 rv = ld2s<NT>(r, i2), qv = ld2s<NT>(q, i2);
        double2 pv = ld2s<NT>(p, i2);
        const double2 uv = make_double2(rv.x + beta * qv.x, rv.y + beta * qv.y);
        pv.x = uv.x + beta * (qv.x + beta * pv.x);
        pv.y = uv.y + beta * (qv.y + beta * pv.y);
        st2(u, i2, uv);
        st2(p, i2, pv);
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        const int64_t i = n - 1;
        const double uv = r[i] + beta * q[i];
        u[i] = uv;
        p[i] = uv + beta * (q[i] + beta * p[i]);
    }
}

int launch_cgs_c2(sla_ctx *c, int64_t n, SolverScalars *sc, Parts apr, int par, Parts res, int count_iter,
                  const double *u, const double *aap, double *q, double *uq, double *x) {
    ProfScope prof(c, SLA_KERNEL_CGS_C2);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(cgs_c2_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, u, aap, q, uq, x);
    else
        hipLaunchKernelGGL(cgs_c2_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, apr, par, res, count_iter, u, aap, q, uq, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
int launch_cgs_c4(sla_ctx *c, int64_t n, SolverScalars *sc, Parts rhonew, int par, const double *r, const double *q,
                  double *u, double *p) {
    ProfScope prof(c, SLA_KERNEL_CGS_C4);
    if (vec_stream_nt(c, n))
        hipLaunchKernelGGL(cgs_c4_kernel<true>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, q, u, p);
    else
        hipLaunchKernelGGL(cgs_c4_kernel<false>, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, sc, rhonew, par, r, q, u, p);
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
                                                               double tol_rel) {
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

int launch_init_scalars(sla_ctx *c, SolverScalars *sc, Parts rho, Parts r0sq, double tol_abs, double tol_rel) {
    hipLaunchKernelGGL(init_scalars_kernel, dim3(1), dim3(kBlock), 0, stream_of(c), sc, rho, r0sq, tol_abs, tol_rel);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// Arnoldi (Sparse.hs:630-667): classical Gram-Schmidt against the SAME A q_i, two passes over Q
// ---------------------------------------------------------------------------------------------
// pass 1: parts[j * gridDim.x + block] = partial of (q_j <.> w), j < ncols     (hhcoli, :655)
// 2-D grid: blockIdx.y selects a group of NC = 4 columns.  One workgroup streaming all (up to 32) columns at once reads
// the basis at 4.6 TB/s; four columns per workgroup (w re-read per group, from the caches) 6980 instead of 6520 Arnoldi
// steps/s on the 2 M-row banded problem (groups of 2 / 8 / 16: 6930 / 6940 / 6670).
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
template <int NC, bool NT>
__global__ void __launch_bounds__(kBlock) arn_update_kernel(int64_t n, const double *Q, int64_t ldq, int ncols,
                                                             const double *hp, int np, int cs, int stride, double *w,
                                                             double *pn, double *Hcol, SolverScalars *sc) {
    __shared__ double s_h[NC];
    __shared__ double s_red[4];
    if (arn_stopped(sc)) return;
    // every workgroup re-reduces the ncols dot products in the same fixed order
    for (int j = threadIdx.x >> 6; j < ncols; j += 4) {
        double a = 0.0;
        for (int i = threadIdx.x & 63; i < np; i += 64) a += hp[(int64_t)j * cs + (int64_t)i * stride];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) s_h[j] = a;
    }
    __syncthreads();
    double h[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) h[j] = j < ncols ? s_h[j] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x < ncols) Hcol[threadIdx.x] = s_h[threadIdx.x];
    double acc = 0.0;
    SLA_VEC_LOOP_BEGIN(n)
        double2 wv = ld2(w, i2);
        double2 t = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < ncols) {
                const double2 qv = ld2s<NT>(Q + (int64_t)j * ldq, i2);
                t.x += h[j] * qv.x;
                t.y += h[j] * qv.y;
            }
        wv.x -= t.x;
        wv.y -= t.y;
        st2(w, i2, wv);
        acc += wv.x * wv.x;
        acc += wv.y * wv.y;
    SLA_VEC_LOOP_END
    if (SLA_HAS_TAIL(n)) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (j < ncols) t += h[j] * Q[(int64_t)j * ldq + n - 1];
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

// ---------------------------------------------------------------------------------------------
// triangular solves (triLowerSolve / triUpperSolve, Sparse.hs:750-811), one dependency level per launch
// ---------------------------------------------------------------------------------------------
// Rows of one level depend only on rows of earlier levels (earlier launches), so plain loads of x are coherent.
// One lane per row: r = ascending left fold of l_ij * x_j over the triangle's side of the row (separately rounded
// multiply and add), x_i = (b_i - r) / t_ii -- the reference's arithmetic, bit for bit.  Latency-bound by nature:
// the schedule's depth times the launch latency is the floor.
__global__ void __launch_bounds__(kBlock) tri_level_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ tcol,
                                                             const double *__restrict__ tval, const double *__restrict__ tdiag,
                                                             const int32_t *__restrict__ order, int64_t first, int64_t count,
                                                             const double *__restrict__ b, double *x) {
    const int64_t t = first + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= first + count) return;
    const int64_t s = tptr[t], e = tptr[t + 1];
    const int i = order[t];
    const double d = tdiag[t], bi = b[i];
    double r = 0.0;
    {
#pragma clang fp contract(off)
        int64_t k = s;
        for (; k + 4 <= e; k += 4) {  // 4 gathers in flight, folded in order
            const double x0 = x[tcol[k]], x1 = x[tcol[k + 1]], x2 = x[tcol[k + 2]], x3 = x[tcol[k + 3]];
            const double p0 = tval[k] * x0, p1 = tval[k + 1] * x1, p2 = tval[k + 2] * x2, p3 = tval[k + 3] * x3;
            r = r + p0;
            r = r + p1;
            r = r + p2;
            r = r + p3;
        }
        for (; k < e; ++k) {
            const double prod = tval[k] * x[tcol[k]];
            r = r + prod;
        }
        x[i] = (bi - r) / d;
    }
}

__global__ void __launch_bounds__(kBlock) tri_sparsify_kernel(int64_t n, double *x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (fabs(x[i]) <= 1e-12) x[i] = 0.0;
}

int launch_tri_level(const sla_csr *T, const sla_tri_plan *p, int64_t first, int64_t count, const double *b, double *x) {
    if (count <= 0) return SLA_OK;
    const int grid = (int)((count + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(tri_level_kernel, dim3(grid), dim3(kBlock), 0, stream_of(T->ctx), p->d_tptr, p->d_tcol, p->d_tval, p->d_tdiag,
                       p->d_order, first, count, b, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_tri_sparsify(sla_ctx *c, int64_t n, double *x) {
    hipLaunchKernelGGL(tri_sparsify_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, stream_of(c), n, x);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

constexpr int kArnDotsGroup = 4;   // basis columns per workgroup of the dots pass
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

int launch_arn_dots(sla_ctx *c, int64_t n, const double *Q, int64_t ldq, int ncols, const double *w, double *parts,
                    SolverScalars *sc) {
    const int g = arn_grid(n);
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
    if (nt) {
#define CALL(NC) hipLaunchKernelGGL((arn_update_kernel<NC, true>), dim3(g), dim3(kBlock), 0, stream_of(c), n, Q, ldq, ncols, hp, np, cs, stride, w, pn, Hcol, sc)
        SLA_NC_DISPATCH(ncols, CALL);
#undef CALL
    } else {
#define CALL(NC) hipLaunchKernelGGL((arn_update_kernel<NC, false>), dim3(g), dim3(kBlock), 0, stream_of(c), n, Q, ldq, ncols, hp, np, cs, stride, w, pn, Hcol, sc)
        SLA_NC_DISPATCH(ncols, CALL);
#undef CALL
    }
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
