// sla_spmv_wdia_lds.hip -- wave-sliced stencil SpMV with uniform records and LDS-staged x windows (form "wdia+ldswin").
//
// For stencils with at most 8 (offset, value) pairs in the whole matrix (5-pt Poisson, 7-pt Laplacian: BASELINE configs 2 and 4).
// Measured on spmv_wdia_kernel (216^3, x and y warm in the memory-side cache): 34 us where the bare access pattern takes 27
// (tools/stencil_probe.cpp) -- that kernel is ISSUE-bound: ~300 instructions per 128 rows, most of them moving record fields
// (masks, value, offset of up to 8 records per slice) from the lanes that fetched them into SGPRs, plus 7 wave-wide gathers that
// touch nearly the same lines.  Two changes:
//  * uniform records: with <= 8 pairs every slice carries the SAME record list (the pairs in table order, absent ones with empty
//    masks), so values and offsets are kernel arguments and a slice is just its 16 lane masks -- 128 bytes, two scalar loads,
//    straight into the operands of s_mov_b64 exec;
//  * LDS windows: the four wavefronts of a workgroup stage the x windows of their 512-row step in LDS (WdWin: the matrix's
//    offsets clustered into runs; 216^3: 3 windows, 1974 elements = 4 aligned 16-byte loads per lane instead of 7 unaligned
//    gathers) and the records read LDS (ds_read2_b64).  The next step's windows are in flight in registers while the current
//    step is folded; two LDS buffers, one barrier and one memory round trip per step.
// Fold order, roundings and epilogue are those of spmv_wdia_kernel (sla_spmv_wdia.hip): the same bits
// (tests/test_gpu_value_indexed.py compares the two on every pattern).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

typedef unsigned long long wd_u64x8s __attribute__((ext_vector_type(8)));

// NP: pairs folded per row (the table padded with empty-mask pairs); NW: staging loads per lane (a buffer holds NW * 256 pairs)
template <int EPI, int NP, int NW>
__global__ void __launch_bounds__(kBlock, NW <= 4 ? 4 : 3) spmv_wdia_lds_kernel(SpmvArgs<int32_t> a, const wd_u64x8s *__restrict__ wum,
                                                                   const double *__restrict__ xg, int32_t nblk, int32_t nslices, int32_t grow0,
                                                                   int32_t xlo, int32_t xhi, const int32_t *__restrict__ sched, int xcd_remap, int stream_nt,
                                                                   WdWin win, WdUni uni) {
    __shared__ wd_f64x2 wd_buf[2][NW * 256];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef = 0.0;   // (set by the prologue, which runs behind the first window loads: see below)
    const bool w_nt = (stream_nt & 1) && a.w != xg + grow0;
    // an operand that IS the gathered vector (K3: As . s) is taken from the staged window instead of a sixth global load
    const bool w_lds = uni.lpos0 >= 0 && a.w == xg + grow0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double acc1 = 0.0, acc2 = 0.0;
    const RbWalk wk = rb_walk(nblk, xcd_remap);
    constexpr bool kUsesW = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT;
    constexpr bool kUsesZ = EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM || EPI == EPI_DOT4;   // (EPI_DOT4: read-only)
    // element 2 pb[k] of a buffer holds x[grow0 - par + 512 blk + omin[k]]: an even index, so the staging loads are aligned pairs
    // whatever the parity of a row slab's first row
    const int par = grow0 & 1;
    int goff[NW];                          // this lane's staging loads: x index relative to (grow0 - par + 512 blk); pair tid + 256 j
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int cpair = tid + 256 * j;
        goff[j] = -(1 << 30);              // past the last window: never inside x
        if (cpair < win.pairs) {
            int k = 0;
#pragma unroll
            for (int q = 1; q < kWdWinMax; ++q)
                if (q < win.n && cpair >= win.pb[q]) k = q;
            goff[j] = win.omin[k] + 2 * (cpair - win.pb[k]);
        }
    }
    // the pairs' values and LDS addresses live in VGPRs (the same in every lane): the SGPRs are needed for the 16 masks of a slice
    double pval[NP];
    uint32_t laddr[NP];                    // byte address of this lane's row pair inside the window of pair k, buffer 0
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        pval[k] = uni.val[k];
        laddr[k] = (uint32_t)(par + wave * 128 + 2 * lane + uni.lpos[k]) << 3;
        asm volatile("" : "+v"(pval[k]), "+v"(laddr[k]));
    }
    // the step visited at position b of the walk (wave-uniform, a scalar load two steps ahead); -1: none
    auto load_blk = [&](int b) -> int { return b < wk.last ? (sched ? sched[b] : b) : -1; };
    wd_f64x2 wr[NW];
    // the x windows of step blk into registers.  Only columns [xlo, xhi) are referenced by these rows and only they are known
    // to be readable (a sharded x is a slab with its halo): a pair that does not touch them -- or lies past the last window -- is
    // replaced by the pair at xlo and never used; a pair may touch ONE element outside (guard slack on both sides).  No
    // lane-dependent branch.
    auto load_windows = [&](int blk) {
        const int gb = grow0 - par + blk * 512 - (xlo - 1);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int gi = gb + goff[j];                          // relative to xlo - 1
            wr[j] = *(const wd_f64x2u *)(xg + (xlo - 1) + ((unsigned)gi < (unsigned)(xhi - xlo + 1) ? gi : 1));
        }
    };
    // the epilogue operands of a row pair: one 16-byte load each, no lane-dependent branch.  Rows past the end re-read the last
    // pair; the last row of an odd row count is the second element of the pair one row back (fix_operands).
    auto load_operands = [&](int row, wd_f64x2 &wv, wd_f64x2 &zv) {
        wv = wd_f64x2{0.0, 0.0};
        zv = wd_f64x2{0.0, 0.0};
        const int prow = max(0, min(row, a.rows - 2));            // (one row: its pair's second element is guard slack)
        if constexpr (kUsesW) {
            if ((EPI != EPI_AXPY_DOT || a.w) && !w_lds)
                wv = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.w + prow)) : *(const wd_f64x2u *)(a.w + prow);
        }
        if constexpr (kUsesZ)
            zv = (stream_nt & 1) ? __builtin_nontemporal_load((const wd_f64x2u *)(a.z + prow)) : *(const wd_f64x2u *)(a.z + prow);
    };
    auto fix_operands = [&](int row, wd_f64x2 &wv, wd_f64x2 &zv) {
        if (row + 1 == a.rows && row > 0) {
            wv.x = wv.y;
            zv.x = zv.y;
        }
    };
    auto stage = [&](int p) {
#pragma unroll
        for (int j = 0; j < NW; ++j) wd_buf[p][tid + 256 * j] = wr[j];
    };
    int b = wk.first;
    int blk_c = load_blk(b), blk_n = load_blk(b + wk.step), blk_f = load_blk(b + 2 * wk.step);
    if (b < wk.last) load_windows(blk_c);
    // (the prologue's loads -- solver scalars, partials -- share the round trip of the first windows instead of preceding it)
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    if (b < wk.last) stage(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the loop is entered with nothing in flight (see its end)
    __syncthreads();
    // Per step: this slice's masks (scalar), its epilogue operands and the NEXT step's windows are issued first and are in flight
    // together while this step is folded out of LDS.
    for (int p = 0; b < wk.last; b += wk.step, p ^= 1) {
        const bool more = b + wk.step < wk.last;   // (workgroup-uniform)
        const int s = blk_c * 4 + wave;
        const bool have = blk_c >= 0 && s < nslices;
        const int row = s * 128 + 2 * lane;
        wd_f64x2 wv, zv;
        load_operands(row, wv, zv);
        if (more) load_windows(blk_n);
        __builtin_amdgcn_sched_barrier(0);         // the vector loads are issued before anything waits for a scalar load
        wd_u64x8s me = {}, mo = {};
        if (have) {
            me = wum[2 * (size_t)s];
            mo = wum[2 * (size_t)s + 1];
        }
        const int blk_g = load_blk(b + 3 * wk.step);
        const bool va = have && row < a.rows, vb = have && row + 1 < a.rows;
        const char *lb = (const char *)wd_buf[p];
        double ya = 0.0, yb = 0.0;
        if (have) {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
            wd_f64x2 xv[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) xv[k] = *(const wd_f64x2u *)(lb + laddr[k]);
            const unsigned long long ex0 = wd_save_exec();   // (restored after every record: the value on entry, not -1)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                // EXEC = the even rows that hold the entry, then the odd rows; v_mul_f64 then v_add_f64 (two roundings).
                // All 64 lanes are active here (wave-uniform control flow only): EXEC goes back to its entry value.
                double pr;
                asm volatile(
                    "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[v], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                    "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[v], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                    "s_mov_b64 exec, %[ex]"
                    : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(pr)
                    : [ex] "s"(ex0), [me] "s"(me[k]), [mo] "s"(mo[k]), [v] "v"(pval[k]), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
            }
        }
        auto epilogue = [&]() {
            if (va) {
                fix_operands(row, wv, zv);
                if constexpr (kUsesW) {
                    if (w_lds) wv = *(const wd_f64x2u *)(lb + ((uint32_t)(par + wave * 128 + 2 * lane + uni.lpos0) << 3));
                }
                wd_epilogue<EPI>(a, row, vb, ya, yb, wv, zv, coef, acc1, acc2, (stream_nt & 2) != 0);
            }
        };
        // Everything issued at the top has to be here now (the windows are staged next); saying so keeps the compiler from waiting
        // conservatively in front of the next step's loads.  The wait stands BEFORE this step's y store: vmcnt counts stores too on
        // this part, and a wait behind the store held every step for the store's acknowledgement -- a full memory round trip with
        // nothing else in flight.  Now the store of step i is acknowledged while step i + 1 is folded (round 3, same-box at 216^3:
        // K1 57.3 -> 53.9 us, K3 56.8 -> 53.0 us).
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        if (more) stage(p ^ 1);
        epilogue();
        __syncthreads();
        blk_c = blk_n;
        blk_n = blk_f;
        blk_f = blk_g;
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

static int wd_lds_nw(const sla_csr *A) { return A->wd_win.pairs <= 4 * 256 ? 4 : kWdWinLoads; }

// its persistent grid: as many workgroups per CU as the two LDS buffers allow (at most 4: 128 VGPRs), a multiple of 8
int wd_lds_grid(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    const int lds = 2 * wd_lds_nw(A) * 256 * 16 + 64;
    int occ = std::max(1, std::min(4, (160 * 1024) / lds));
    if (c->wd_lds_occ > 0) occ = std::min(occ, c->wd_lds_occ);
    return std::min<int>(kMaxParts, std::max(8, (occ * c->n_cu) & ~7));
}

template <int EPI, int NP>
static int launch_np(const sla_csr *A, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk, int grid, int stream_nt) {
    sla_ctx *c = A->ctx;
#define SLA_WDL_LAUNCH(NW_)                                                                                                              \
    hipLaunchKernelGGL((spmv_wdia_lds_kernel<EPI, NP, NW_>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, (const wd_u64x8s *)A->d_wum, a.x, \
                       nblk, A->nslices, (int32_t)A->row_begin, A->wd_col_lo, A->wd_col_hi + 1, sched, c->xcd_remap, stream_nt, A->wd_win, A->wd_uni)
    if (wd_lds_nw(A) == 4) SLA_WDL_LAUNCH(4);
    else SLA_WDL_LAUNCH(kWdWinLoads);
#undef SLA_WDL_LAUNCH
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

template <int EPI>
static int launch_epi(const sla_csr *A, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk, int grid, int stream_nt) {
    const int n = A->wd_uni.n;
    if (n <= 5) return launch_np<EPI, 5>(A, a, sched, nblk, grid, stream_nt);
    if (n <= 7) return launch_np<EPI, 7>(A, a, sched, nblk, grid, stream_nt);
    return launch_np<EPI, 8>(A, a, sched, nblk, grid, stream_nt);
}

int launch_wdia_lds(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, const int32_t *sched, int32_t nblk, int grid, int stream_nt) {
    switch (epi) {
        case EPI_NONE: return launch_epi<EPI_NONE>(A, a, sched, nblk, grid, stream_nt);
        case EPI_DOT: return launch_epi<EPI_DOT>(A, a, sched, nblk, grid, stream_nt);
        case EPI_DOT2: return launch_epi<EPI_DOT2>(A, a, sched, nblk, grid, stream_nt);
        case EPI_DOT4: return launch_epi<EPI_DOT4>(A, a, sched, nblk, grid, stream_nt);
        case EPI_RES: return launch_epi<EPI_RES>(A, a, sched, nblk, grid, stream_nt);
        case EPI_AXPY_DOT: return launch_epi<EPI_AXPY_DOT>(A, a, sched, nblk, grid, stream_nt);
        case EPI_XPBY_NRM: return launch_epi<EPI_XPBY_NRM>(A, a, sched, nblk, grid, stream_nt);
        case EPI_SUB: return launch_epi<EPI_SUB>(A, a, sched, nblk, grid, stream_nt);
    }
    return fail(SLA_ERR_INVALID, "launch_wdia_lds: unknown epilogue");
}

}  // namespace sla
