// sla_spmv_dict.hip -- dictionary-compressed (#>) kernels for stencil / banded structure (Data/Sparse/Common.hs:242-260 semantics,
// one lane per row: the reference's ascending left fold bit for bit).
//   spmv_diag_kernel / spmv_dual_diag_kernel   values + ONE BYTE per entry indexing a table of <= 256 diagonal offsets (9 B per entry)
//   spmv_vdict_kernel                          one byte per entry indexing a table of <= 256 (offset, value) pairs: no value stream
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

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
__global__ void __launch_bounds__(kBlock, (XW ? (EPI == 2 && sizeof(RP) == 4 ? 7 : (kOcc8<EPI, RP>)) : 8)) spmv_diag_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr,
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
                        // (the epilogue's operands are issued in front of the row's gathers, not behind its fold: sla_spmv_stream.hip)
                        double wpre, zpre;
                        spmv_operands<EPI, RP>(a, r0 + tid, wpre, zpre);
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
                        spmv_epilogue_pre<EPI, RP>(a, r0 + tid, acc, coef, acc1, acc2, wpre, zpre);
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
__global__ void __launch_bounds__(kBlock, (XW ? 6 : 8)) spmv_dual_diag_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr,
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
// launchers (called by launch_spmv, sla_spmv.hip)
// ---------------------------------------------------------------------------------------------
namespace {
template <int EPI, typename RP>
int launch_diag_t(const sla_csr *A, const SpmvArgs<RP> &a, int grid) {
    sla_ctx *c = A->ctx;
    if (diag_xwin_on(A))
        hipLaunchKernelGGL((spmv_diag_kernel<EPI, RP, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val, a.rb,
                           a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, c->xcd_remap);
    else
        hipLaunchKernelGGL((spmv_diag_kernel<EPI, RP, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val, a.rb,
                           a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, c->xcd_remap);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
template <typename RP>
int launch_diag_rp(const sla_csr *A, int epi, const SpmvArgs<RP> &a, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_diag_t<EPI_NONE, RP>(A, a, grid);
        case EPI_DOT: return launch_diag_t<EPI_DOT, RP>(A, a, grid);
        case EPI_DOT2: return launch_diag_t<EPI_DOT2, RP>(A, a, grid);
        case EPI_DOT4: return launch_diag_t<EPI_DOT4, RP>(A, a, grid);
        case EPI_RES: return launch_diag_t<EPI_RES, RP>(A, a, grid);
        case EPI_AXPY_DOT: return launch_diag_t<EPI_AXPY_DOT, RP>(A, a, grid);
        case EPI_XPBY_NRM: return launch_diag_t<EPI_XPBY_NRM, RP>(A, a, grid);
        case EPI_SUB: return launch_diag_t<EPI_SUB, RP>(A, a, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_diag: unknown epilogue");
}
template <typename RP>
int launch_dual_diag_rp(const sla_csr *A, const SpmvArgs<RP> &a, const double *x2, const double *b2, int grid) {
    sla_ctx *c = A->ctx;
    if (diag_xwin_on(A))
        hipLaunchKernelGGL((spmv_dual_diag_kernel<RP, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val,
                           a.rb, a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, x2, b2, c->xcd_remap);
    else
        hipLaunchKernelGGL((spmv_dual_diag_kernel<RP, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_code, a.val,
                           a.rb, a.rbk, a.x, A->d_rbw, A->d_dict, (int32_t)A->n, (int32_t)A->row_begin, x2, b2, c->xcd_remap);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
template <int EPI>
int launch_vdict_t(const sla_csr *A, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid) {
    sla_ctx *c = A->ctx;
    const bool xw = A->use_xwin && c->xwin;
#define SLA_VD_LAUNCH(E, XW_, DUAL_)                                                                                        \
    hipLaunchKernelGGL((spmv_vdict_kernel<E, XW_, DUAL_>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr,            \
                       (const uint32_t *)A->d_vcode, a.x, A->d_vdoff, A->d_vdval, A->nblk_vd, (int32_t)A->n,                  \
                       (int32_t)A->row_begin, x2, b2, c->xcd_remap)
    if (x2) {
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
}  // namespace

// values + 1-byte column codes (<= 256 distinct diagonals)
int launch_spmv_diag(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) { return launch_diag_rp<int32_t>(A, epi, a, grid); }
int launch_spmv_diag(const sla_csr *A, int epi, const SpmvArgs<int64_t> &a, int grid) { return launch_diag_rp<int64_t>(A, epi, a, grid); }
int launch_spmv_dual_diag(const sla_csr *A, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid) { return launch_dual_diag_rp<int32_t>(A, a, x2, b2, grid); }
int launch_spmv_dual_diag(const sla_csr *A, const SpmvArgs<int64_t> &a, const double *x2, const double *b2, int grid) { return launch_dual_diag_rp<int64_t>(A, a, x2, b2, grid); }
// one byte per entry into a table of (offset, value) pairs; x2 != null: the dual (K1 + residual) variant
int launch_spmv_vdict(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const double *x2, const double *b2, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_vdict_t<EPI_NONE>(A, a, x2, b2, grid);
        case EPI_DOT: return launch_vdict_t<EPI_DOT>(A, a, x2, b2, grid);
        case EPI_DOT2: return launch_vdict_t<EPI_DOT2>(A, a, x2, b2, grid);
        case EPI_DOT4: return launch_vdict_t<EPI_DOT4>(A, a, x2, b2, grid);
        case EPI_RES: return launch_vdict_t<EPI_RES>(A, a, x2, b2, grid);
        case EPI_AXPY_DOT: return launch_vdict_t<EPI_AXPY_DOT>(A, a, x2, b2, grid);
        case EPI_XPBY_NRM: return launch_vdict_t<EPI_XPBY_NRM>(A, a, x2, b2, grid);
        case EPI_SUB: return launch_vdict_t<EPI_SUB>(A, a, x2, b2, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_vdict: unknown epilogue");
}

}  // namespace sla
