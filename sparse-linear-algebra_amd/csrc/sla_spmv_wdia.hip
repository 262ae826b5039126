// sla_spmv_wdia.hip -- wave-sliced (offset, value) (#>): per 128-row slice the union of its (col - row, value) pairs with lane masks, all
// in SGPRs, two rows per lane: one 16-byte gather + 2 x (v_mul + v_add) per entry, no LDS, no barrier.  Constant-coefficient
// stencils, and (VV) any banded structure with per-row value blocks.  Data/Sparse/Common.hs:242-260 semantics, every row the
// reference's ascending left fold bit for bit.  (sla_spmv_wdia_lds.hip holds the uniform-record / LDS-window variant.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

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
// SF (round 5): BiCGSTAB's K2 folded into K3 -- the gathered vector is s = r - alpha Ap and is never stored: every gather loads the row
// pair of r (xg) AND of Ap (a.fs_ap) and the fold combines them with bicg_k2_kernel's multiply-add (alpha formed like there, from K1's
// partial sums a.pa); As . s takes s for the own rows the same way.  Twice the gathers (cache hits at the sizes this kernel serves), one
// launch and a 24 n-byte pass less per step; the K4+K5 sweep rebuilds s from r and Ap (bicg_k45_kernel<.., true>).  Eight more row pairs in
// registers: compiled for 4 (VV: 3) workgroups per CU, the surplus of the grid leaves at once (`active`).
constexpr int kWdBlocksPerCuSF = 4, kWdBlocksPerCuSFVV = 3;
template <int EPI, bool VV, bool SF = false>
__global__ void __launch_bounds__(kBlock, SF ? (VV ? kWdBlocksPerCuSFVV : kWdBlocksPerCuSF) : VV ? (EPI == EPI_DOT4 ? kWdBlocksPerCuVV4 : kWdBlocksPerCuVV) : kWdBlocksPerCu) spmv_wdia_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ sptr,
                                                               const unsigned long long *__restrict__ wme,
                                                               const unsigned long long *__restrict__ wmo,
                                                               const double *__restrict__ wval, const int32_t *__restrict__ woff,
                                                               const double *__restrict__ wvblk, const double *__restrict__ xg, int32_t nblk, int32_t nslices,
                                                               int32_t grow0, int32_t xlen, const int32_t *__restrict__ sched, int xcd_remap, int stream_nt, int active) {
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef = 0.0;   // (set by the prologue, which runs BEHIND the first descriptor / record loads: see below)
    double alpha = 0.0;  // SF: s = r - alpha Ap
    const double *apg = SF ? a.fs_ap : nullptr;
    auto form_alpha = [&]() {   // as bicg_k2_kernel forms it (behind the prologue: sc->done has been looked at)
        if constexpr (SF) {
            alpha = a.sc->rho2[(a.step_begin >> 1) & 1] / reduce_parts(a.pa, a.npa, a.pa_stride, s_red);
            if (blockIdx.x == 0 && tid == 0) a.sc->alpha = alpha;
        }
    };
    const bool w_nt = stream_nt && a.w != xg + grow0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(nblk, xcd_remap, active);
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
        wd_f64x2 qv[SF ? 8 : 1];    // SF: the same row pairs of Ap
        wd_f64x2 qw;                // SF: Ap on the lane's own rows
        wd_f64x2 wv, zv;
        unsigned long long me, mo;  // lane k: masks of record k
        double v;                   // lane k: value of record k
        int blk, e0, cnt;
    };
    // gathers of the 8 records held by lanes 0..7 of `r` (first record e0) for the row pair starting at `row`
    auto gather8 = [&](const WdRec &r, int e0, int row, wd_f64x2 *xv, wd_f64x2 *vv, wd_f64x2 *qv) {
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
                if constexpr (SF) qv[k] = *(const wd_f64x2u *)((const char *)(apg + ok) + g8);
                if (VV) vv[k] = *(const wd_f64x2 *)(wvblk + ((size_t)(e0 + k) << 7) + 2 * lane);
            }
        }
    };
    // the products of up to 8 records folded into the row pair's sums, in record (= ascending column) order
    auto fold8 = [&](unsigned long long rme, unsigned long long rmo, double rv, int nrec, const wd_f64x2 *xv, const wd_f64x2 *vv,
                     double &ya, double &yb) {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
        const unsigned long long ex0 = wd_save_exec();   // EXEC is put back to its value on entry (not to -1: were a
                                                                       // compiler-predicated region ever to enclose this, its dead lanes stay dead)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= nrec) break;  // wave-uniform
            const unsigned long long me = wd_lane_u64(rme, k), mo = wd_lane_u64(rmo, k);
            // EXEC = the even rows that hold the entry, then the odd rows; v_mul_f64 then v_add_f64 (two
            // roundings).  All 64 lanes are active here (wave-uniform control flow only): EXEC goes back to its entry value.
            double p;
            if constexpr (VV) {
                asm volatile(
                    "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[va], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                    "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[vb], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                    "s_mov_b64 exec, %[ex]"
                    : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(p)
                    : [ex] "s"(ex0), [me] "s"(me), [mo] "s"(mo), [va] "v"(vv[k].x), [vb] "v"(vv[k].y), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
            } else {
                const double vk = wd_lane_f64(rv, k);
                asm volatile(
                    "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[v], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                    "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[v], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                    "s_mov_b64 exec, %[ex]"
                    : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(p)
                    : [ex] "s"(ex0), [me] "s"(me), [mo] "s"(mo), [v] "s"(vk), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
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
            if (SF) asm("" : "=v"(st.qv[k]));
        }
        st.qw = wd_f64x2{0.0, 0.0};
        if (cnt < 0) return;
        const int row = (blk * 4 + wave) * 128 + 2 * lane;  // this lane's rows: row, row + 1
        const bool va = row < a.rows, vb = row + 1 < a.rows;
        if constexpr (SF) {   // w = s on the own rows: r and Ap there (the gathers' lines)
            if (vb) {
                st.wv = *(const wd_f64x2u *)(xg + grow0 + row);
                st.qw = *(const wd_f64x2u *)(apg + grow0 + row);
            } else if (va) {
                st.wv.x = xg[grow0 + row];
                st.qw.x = apg[grow0 + row];
            }
        }
        if (vb) {
            // epilogue operands are single-use streams: past the caches when the vectors overflow them anyway (see
            // vec_stream_nt) -- except an operand that IS the gathered vector (K3: w = s = x), which must stay
            if constexpr (kUsesW && !SF) {
                if (EPI != EPI_AXPY_DOT || a.w)
                    st.wv = w_nt ? __builtin_nontemporal_load((const wd_f64x2 *)(a.w + row)) : *(const wd_f64x2 *)(a.w + row);
            }
            if constexpr (kUsesZ)
                st.zv = stream_nt ? __builtin_nontemporal_load((const wd_f64x2 *)(a.z + row)) : *(const wd_f64x2 *)(a.z + row);
        } else if (va) {
            if constexpr (kUsesW && !SF) { if (EPI != EPI_AXPY_DOT || a.w) st.wv.x = a.w[row]; }
            if constexpr (kUsesZ) st.zv.x = a.z[row];
        }
        gather8(r, e0, row, st.xv, st.vv, st.qv);
    };
    auto combine8 = [&](wd_f64x2 *xv, const wd_f64x2 *qv) {   // SF: bicg_k2_kernel's multiply-add (lanes that did not load hold garbage the fold never uses)
        if constexpr (SF) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xv[k].x = __builtin_fma(-alpha, qv[k].x, xv[k].x);
                xv[k].y = __builtin_fma(-alpha, qv[k].y, xv[k].y);
            }
        }
    };
    // fold the slice's products row by row and run the epilogue
    auto fold = [&](Stage &st) {
        if (st.cnt < 0) return;
        const int row = (st.blk * 4 + wave) * 128 + 2 * lane;
        const bool va = row < a.rows, vb = row + 1 < a.rows;
        double ya = 0.0, yb = 0.0;
        combine8(st.xv, st.qv);
        if constexpr (SF) {
            st.wv.x = __builtin_fma(-alpha, st.qw.x, st.wv.x);
            st.wv.y = __builtin_fma(-alpha, st.qw.y, st.wv.y);
        }
        fold8(st.me, st.mo, st.v, st.cnt, st.xv, st.vv, ya, yb);
        // slices with more than 8 records (27-point stencils, wide bands): further chunks of 8, fetched, gathered and
        // folded one after the other (no pipelining across chunks)
        for (int c0 = 8; c0 < st.cnt; c0 += 8) {
            WdRec rr;
            load_rec(st.e0 + c0, st.cnt - c0, rr);
            wd_f64x2 xt[8], vt[VV ? 8 : 1], qt[SF ? 8 : 1];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm("" : "=v"(xt[k]));
                if (VV) asm("" : "=v"(vt[k]));
                if (SF) asm("" : "=v"(qt[k]));
            }
            gather8(rr, st.e0 + c0, row, xt, vt, qt);
            combine8(xt, qt);
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
            if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
            form_alpha();
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
        // The prologue (solver done? residual test, step count, alpha / beta from partials) runs HERE, behind the first
        // descriptor and record loads: its own loads share their round trip instead of preceding them -- the head of the kernel
        // was five dependent trips to memory (prologue, schedule, descriptor, records, gathers), a third of a 10 us launch at
        // 1 M rows.  A workgroup that must exit has only loaded a few words it does not use.
        if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
        form_alpha();
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

// launcher (called by launch_spmv, sla_spmv.hip): sched / nblk / grid select the whole walk or one part of the interior / boundary split
namespace {
template <int EPI>
int launch_wdia_t(const sla_csr *A, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk_wd, int grid, int stream_nt) {
    sla_ctx *c = A->ctx;
    // the four-sum instantiation of the variable-coefficient form holds one workgroup per CU less than the grid is sized for (its
    // launch bounds): the surplus workgroups leave at once instead of running a second round (2 M-row banded K3 26.7 -> 24.6 us)
    int active = grid;
    if (A->wd_vv && EPI == EPI_DOT4) active = std::min(grid, std::max(8, (kWdBlocksPerCuVV4 * c->n_cu) & ~7));
    if (a.fs_ap) {   // K2 folded into K3 (see the kernel's header)
        if constexpr (EPI == EPI_DOT4) {
            if (!a.sc || !a.pa) return fail(SLA_ERR_INVALID, "launch_spmv_wdia: fused s needs the solver scalars and K1's partial sums");
            active = std::min(grid, std::max(8, ((A->wd_vv ? kWdBlocksPerCuSFVV : kWdBlocksPerCuSF) * c->n_cu) & ~7));
            if (A->wd_vv)
                SLA_KLAUNCH(c, (spmv_wdia_kernel<EPI, true, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                                   A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                                   sched, c->xcd_remap, stream_nt, active);
            else
                SLA_KLAUNCH(c, (spmv_wdia_kernel<EPI, false, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                                   A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                                   sched, c->xcd_remap, stream_nt, active);
            SLA_HIP_TRY(hipGetLastError());
            return SLA_OK;
        }
        return fail(SLA_ERR_INVALID, "launch_spmv_wdia: fused s is defined for the four-sum epilogue only");
    }
    if (A->wd_vv)
        SLA_KLAUNCH(c, (spmv_wdia_kernel<EPI, true>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                           A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                           sched, c->xcd_remap, stream_nt, active);
    else
        SLA_KLAUNCH(c, (spmv_wdia_kernel<EPI, false>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, A->d_wptr, A->d_wme, A->d_wmo,
                           A->d_wval, A->d_woff, A->d_wvblk, a.x, nblk_wd, A->nslices, (int32_t)A->row_begin, (int32_t)A->n,
                           sched, c->xcd_remap, stream_nt, active);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}
}  // namespace

int launch_spmv_wdia(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk_wd, int grid, int stream_nt) {
    switch (epi) {
        case EPI_NONE: return launch_wdia_t<EPI_NONE>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_DOT: return launch_wdia_t<EPI_DOT>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_DOT2: return launch_wdia_t<EPI_DOT2>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_DOT4: return launch_wdia_t<EPI_DOT4>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_RES: return launch_wdia_t<EPI_RES>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_AXPY_DOT: return launch_wdia_t<EPI_AXPY_DOT>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_XPBY_NRM: return launch_wdia_t<EPI_XPBY_NRM>(A, a, sched, nblk_wd, grid, stream_nt);
        case EPI_SUB: return launch_wdia_t<EPI_SUB>(A, a, sched, nblk_wd, grid, stream_nt);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_wdia: unknown epilogue");
}

}  // namespace sla
