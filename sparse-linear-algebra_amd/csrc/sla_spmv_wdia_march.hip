// sla_spmv_wdia_march.hip -- the LDS-window stencil SpMV walking a 3-D stencil plane by plane (form "wdia+march").
//
// spmv_wdia_lds_kernel stages THREE x windows per 512-row step: the in-plane window (512 + 2 D1 elements for offsets in [-D1, D1])
// and 512 elements of the plane behind and of the plane ahead (offsets -D, +D): 1974 elements per 512 rows at 216^3, 3.9 x the rows.
// Measured on the bare access pattern (tools/stencil_probe.cpp, profiles/r03_stencil_probe.txt): 35.3 us per pass over x + y at
// 216^3 where a plain copy of the same bytes takes 28.5 us -- the redundant window loads are L2 hits, but they share the L2's
// request path with the compulsory misses.
// Here a workgroup owns an in-plane TILE of 512 rows and walks a RUN of consecutive planes.  Per step it stages ONE window: the
// in-plane window of the plane ahead.  The plane behind is the buffer staged two steps ago, the current plane the one staged last
// step: four LDS buffers in rotation, one barrier per step, 944 elements per 512 rows (1.84 x).  The same probe: 31.8 us at 216^3,
// 49.4 us at 256^3 (= the plain copy).
//   * tasks = (tile, run), tile-major; XCD x takes the x-th contiguous eighth of them, so neighbouring tiles -- whose in-plane
//     windows overlap by 2 D1 elements -- share an L2;
//   * a plane is D rows and D need not be a multiple of 128 (216^2 = 364.5 slices): the lane masks are lowered a second time in
//     march order, 128 bytes per (tile, plane, wavefront), rows outside the plane / the matrix with empty masks;
//   * the window of plane k + 2, the epilogue operands of step k + 1 and the masks of step k are issued together and are in flight
//     while step k is folded: one memory round trip per step, as in spmv_wdia_lds_kernel, with half the bytes; the step loop is
//     unrolled four times so that the buffer of every pair is a compile-time offset.  (Two planes in flight -- a second register
//     set, counted waits -- measured same-box: K1 47.5 -> 49-50 us; the four-sum K3 then needs 133 VGPRs and drops to three
//     workgroups per CU: 48.4 -> 59.6 us; staging at the TOP of a step with compiler-counted waits, the loads in flight across
//     the barrier: K1 45.0 -> 46.3 us; profiles/r03_ab_march.txt);
//   * fold order, roundings and epilogues are spmv_wdia_kernel's (shared wd_epilogue): every row bit-identical to the other forms
//     (tests/test_gpu_value_indexed.py).  The partial sums of the fused dot products are grouped by task instead of by step.
// Taken for 5- and 7-pair stencils (one pair at -D, one at +D); a row slab of a sharded matrix is walked like a matrix of its own
// (its whole-slab launches: the interior / boundary launches of an overlapped exchange stay step-based), sla_lower.cpp: low_value_indexed.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

typedef unsigned long long wdm_u64x8s __attribute__((ext_vector_type(8)));

constexpr int kMarchBufBytes = 512 * 16;   // one staged window: <= 512 pairs

// WX: the epilogue operand w IS the gathered vector (K3: As . s) and comes from the staged window instead of a global load
// SF (round 5, BiCGSTAB's K2 folded into K3): the gathered vector does not exist in memory -- it is s = r - alpha Ap (Sparse.hs:975-976),
// built WHILE the windows are staged: xg is r, a.fs_ap is Ap, alpha = rho / (Ap . r0hat) from K1's partial sums (a.pa) exactly as
// bicg_k2_kernel forms it (same re-reduction, same division, the same fused multiply-add per element: the staged values are that kernel's
// bits).  Per step a workgroup loads two windows instead of one and no kernel writes or re-reads s: K2 + K3 streamed 24 n + 25 n bytes,
// this launch 32 n (r, Ap, r0hat in; As out); the fused K4+K5 sweep rebuilds s from r and Ap, which it reads anyway (bicg_k45_kernel<.., true>).
// SF = 2: the same for cgsStep (Sparse.hs:931-933) -- the gathered vector is u + q with q = u - alpha Ap (xg is u), cgs_c2_kernel's
// expressions; alpha is the prologue's coefficient (formed from a.pa like CGNE's); q, u + q and x are left to cgs_c24_kernel.
template <int EPI, int NP, bool WX, int SF>
__global__ void __launch_bounds__(kBlock, 3) spmv_wdia_march_kernel(SpmvArgs<int32_t> a, const wdm_u64x8s *__restrict__ wum, const double *__restrict__ xg,
                                                                    WdMarch m, int32_t grow0, int32_t xlo, int32_t xhi, int xcd_remap, int stream_nt, WdUni uni) {
    __shared__ wd_f64x2 wd_buf[4][512];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double coef = 0.0;
    double alpha = 0.0;                                // SF: s = r - alpha Ap
    const double *apg = SF ? a.fs_ap : nullptr;
    const bool w_nt = (stream_nt & 1) != 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double acc1 = 0.0, acc2 = 0.0;
    constexpr bool kUsesW = (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT) && !WX;
    constexpr bool kUsesZ = EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM || EPI == EPI_DOT4;
    // the tasks of this workgroup: XCD (= blockIdx.x % 8) x owns the x-th contiguous eighth
    int q, qstep, qlast;
    if (xcd_remap && (gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = (m.ntasks + 7) >> 3;
        q = xcd * per + (blockIdx.x >> 3);
        qstep = gridDim.x >> 3;
        qlast = min((xcd + 1) * per, m.ntasks);
    } else {
        q = blockIdx.x;
        qstep = gridDim.x;
        qlast = m.ntasks;
    }
    // pair values and LDS byte addresses (inside a buffer) in VGPRs, the same in every lane of a row pair
    double pval[NP];
    uint32_t laddr[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        pval[k] = uni.val[k];
        laddr[k] = (uint32_t)(wave * 128 + 2 * lane + uni.lpos[k]) << 3;
        asm volatile("" : "+v"(pval[k]), "+v"(laddr[k]));
    }
    const uint32_t laddr0 = (uint32_t)(wave * 128 + 2 * lane + uni.lpos0) << 3;
    const bool second = tid + 256 < m.pairs;          // this lane stages a second pair of the window
    const double *wp = a.w ? a.w : a.z;                // (EPI_AXPY_DOT without w: loaded all the same -- from z, always there -- not used)
    // The window of plane k of the current tile: pairs tid and tid + 256.  Only columns [xlo, xhi) are known to be readable: a pair
    // that does not touch them is replaced by the pair at xlo and never used (empty masks); one element of slack either side.  No
    // load of the step loop stands under a condition -- uniform or not: with a conditional load in the loop the compiler can no
    // longer count vmcnt and drains it in front of every use.
    long long xbase = 0;                               // tile * 512 + omin - (xlo - 1) + 2 tid, set per task
    const unsigned long long span = (unsigned long long)((long long)xhi - xlo + 1);
    // (SF: q0 / q1 are the same pairs of Ap; st combines)
    auto ld = [&](int k, wd_f64x2 &r0, wd_f64x2 &r1, wd_f64x2 &q0, wd_f64x2 &q1) {
        const long long g0 = (long long)k * m.D + xbase;                                   // relative to xlo - 1
        const long long g1 = g0 + (second ? 512 : 0);
        const long long o0 = (xlo - 1) + ((unsigned long long)g0 < span ? g0 : 1), o1 = (xlo - 1) + ((unsigned long long)g1 < span ? g1 : 1);
        r0 = *(const wd_f64x2u *)(xg + o0);
        r1 = *(const wd_f64x2u *)(xg + o1);
        if constexpr (SF) {
            q0 = *(const wd_f64x2u *)(apg + o0);
            q1 = *(const wd_f64x2u *)(apg + o1);
        }
    };
    auto st = [&](int buf, wd_f64x2 r0, wd_f64x2 r1, const wd_f64x2 &q0, const wd_f64x2 &q1) {
        if constexpr (SF == 1) {   // bicg_k2_kernel's expression: one fused multiply-add per element
            r0.x = __builtin_fma(-alpha, q0.x, r0.x);
            r0.y = __builtin_fma(-alpha, q0.y, r0.y);
            r1.x = __builtin_fma(-alpha, q1.x, r1.x);
            r1.y = __builtin_fma(-alpha, q1.y, r1.y);
        } else if constexpr (SF == 2) {   // cgs_c2_kernel's: q = u - alpha Ap (one multiply-add), then u + q
            r0.x = r0.x + __builtin_fma(-alpha, q0.x, r0.x);
            r0.y = r0.y + __builtin_fma(-alpha, q0.y, r0.y);
            r1.x = r1.x + __builtin_fma(-alpha, q1.x, r1.x);
            r1.y = r1.y + __builtin_fma(-alpha, q1.y, r1.y);
        }
        wd_buf[buf][tid] = r0;
        if (second) wd_buf[buf][tid + 256] = r1;
    };
    // SF: alpha as bicg_k2_kernel forms it, once per workgroup (behind the prologue: sc->done has been looked at)
    auto form_alpha = [&]() {
        if constexpr (SF == 1) {
            alpha = a.sc->rho2[(a.step_begin >> 1) & 1] / reduce_parts(a.pa, a.npa, a.pa_stride, s_red);
            if (blockIdx.x == 0 && tid == 0) a.sc->alpha = alpha;
        } else if constexpr (SF == 2) {
            alpha = coef;   // (spmv_prologue<EPI_AXPY_DOT> with a.pa: rho / sum, published as sc->alpha -- cgs_c2_kernel's alpha)
        }
    };
    // the epilogue operands of a row pair: one 16-byte load each.  Rows past the end re-read the last pair; the last row of an odd
    // row count is the second element of the pair one row back.
    auto load_operands = [&](int row, wd_f64x2 &wv, wd_f64x2 &zv) {
        wv = wd_f64x2{0.0, 0.0};
        zv = wd_f64x2{0.0, 0.0};
        const int prow = max(0, min(row, a.rows - 2));
        if constexpr (kUsesW) wv = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(wp + prow)) : *(const wd_f64x2u *)(wp + prow);
        if constexpr (kUsesZ) zv = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.z + prow)) : *(const wd_f64x2u *)(a.z + prow);
    };
    bool first = true;
    for (; q < qlast; q += qstep) {
        const int tile = q / m.S, seg = q - tile * m.S;
        const int k0 = seg * m.PS, k1 = min(k0 + m.PS, m.planes), nst = k1 - k0;
        const int pos = tile * 512 + wave * 128 + 2 * lane;      // this lane's row pair inside the plane
        const bool in_plane = pos < m.D;                         // (D is even: both rows or none)
        const bool wave_in = tile * 512 + wave * 128 < m.D;
        xbase = (long long)grow0 + (long long)tile * 512 + m.omin - (xlo - 1) + 2 * tid;   // (x is addressed by global column; rows are local)
        wd_f64x2 pa0, pa1, pb0, pb1, r0, r1, wv, zv, wvn, zvn;
        wd_f64x2 qa0 = {0.0, 0.0}, qa1 = qa0, qb0 = qa0, qb1 = qa0, q0 = qa0, q1 = qa0;
        ld(k0 - 1, pa0, pa1, qa0, qa1);
        ld(k0, pb0, pb1, qb0, qb1);
        ld(k0 + 1, r0, r1, q0, q1);
        load_operands(k0 * m.D + pos, wv, zv);

        if (first) {
            // (the prologue's loads -- solver scalars, partials -- share the round trip of the first windows)
            if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
            form_alpha();
            first = false;
        } else {
            __syncthreads();                                     // the previous task's last fold has read its buffers
        }
        st(3, pa0, pa1, qa0, qa1);
        st(0, pb0, pb1, qb0, qb1);
        st(1, r0, r1, q0, q1);
        __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): the loop is entered with nothing in flight
        __syncthreads();
        const wdm_u64x8s *wm = wum + 2 * (((size_t)tile * (size_t)m.planes + (size_t)k0) * 4 + (size_t)wave);
        for (int i0 = 0; i0 < nst; i0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i >= nst) break;
                const int kk = k0 + i;
                const int row = kk * m.D + pos;
                // issued first and in flight together while this step is folded out of LDS: the window of plane kk + 2 (the last
                // steps of a run re-load plane k1, cache hits), the operands of step i + 1, this step's masks
                ld(min(kk + 2, k1), r0, r1, q0, q1);
                load_operands(row + m.D, wvn, zvn);
                __builtin_amdgcn_sched_barrier(0);
                wdm_u64x8s me = {}, mo = {};
                if (wave_in) {
                    me = wm[8 * (size_t)i];
                    mo = wm[8 * (size_t)i + 1];
                }
                const bool va = in_plane && row < a.rows, vb = in_plane && row + 1 < a.rows;
                const char *lb = (const char *)wd_buf;
                double ya = 0.0, yb = 0.0;
                if (wave_in) {
#pragma clang fp contract(off)  // a*x then +: two roundings like the reference, never an FMA
                    wd_f64x2 xv[NP];
                    xv[0] = *(const wd_f64x2u *)(lb + ((u + 3) & 3) * kMarchBufBytes + laddr[0]);
#pragma unroll
                    for (int k = 1; k < NP - 1; ++k) xv[k] = *(const wd_f64x2u *)(lb + u * kMarchBufBytes + laddr[k]);
                    xv[NP - 1] = *(const wd_f64x2u *)(lb + ((u + 1) & 3) * kMarchBufBytes + laddr[NP - 1]);
                    const unsigned long long ex0 = wd_save_exec();   // (restored after every record: the value on entry, not -1)
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        double pr;
                        asm volatile(
                            "s_mov_b64 exec, %[me]\n\tv_mul_f64 %[p], %[v], %[xa]\n\tv_add_f64 %[ya], %[ya], %[p]\n\t"
                            "s_mov_b64 exec, %[mo]\n\tv_mul_f64 %[p], %[v], %[xb]\n\tv_add_f64 %[yb], %[yb], %[p]\n\t"
                            "s_mov_b64 exec, %[ex]"
                            : [ya] "+v"(ya), [yb] "+v"(yb), [p] "=&v"(pr)
                            : [ex] "s"(ex0), [me] "s"(me[k]), [mo] "s"(mo[k]), [v] "v"(pval[k]), [xa] "v"(xv[k].x), [xb] "v"(xv[k].y));
                    }
                    if constexpr (WX) wv = *(const wd_f64x2u *)(lb + u * kMarchBufBytes + laddr0);
                }
                // everything issued at the top is here now; the wait stands in front of this step's y store (vmcnt counts stores
                // too: behind it, every step would sit out the store's acknowledgement)
                __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
                st((u + 2) & 3, r0, r1, q0, q1);                 // plane kk + 2: read from step i + 1 on
                if (va) {
                    if (row + 1 == a.rows && row > 0) {          // the last row of an odd row count (load_operands)
                        if constexpr (!WX) wv.x = wv.y;          // (the window holds the row's own pair)
                        zv.x = zv.y;
                    }
                    wd_epilogue<EPI>(a, row, vb, ya, yb, wv, zv, coef, acc1, acc2, (stream_nt & 2) != 0);
                }
                __syncthreads();
                wv = wvn;
                zv = zvn;
            }
        }
    }
    if (first) {
        if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
        form_alpha();   // (workgroup 0 always has a task; the others' value is not used)
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

// one workgroup per task when the tasks fit the chip (they are sized for that, sla_lower.cpp), a multiple of 8
int wd_march_grid(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    const int slots = std::max(8, (std::max(1, c->wd_march_occ) * c->n_cu) & ~7);
    const int per = (A->wd_mg.ntasks + 7) / 8;
    return std::min<int>(kMaxParts & ~7, std::min(slots, 8 * per));
}

// The runs are sized so that the tasks fill the chip ONCE (sla_lower.cpp: 4 workgroups per CU).  An instantiation whose registers
// allow only 3 per CU (the CGS / CGNE epilogues, the four-sum K3: 133-154 VGPRs) would run such a grid in two rounds -- CGS's C3 took
// 71 us where 56 us suffice -- so its tasks are re-cut for 3 per CU at launch: fewer, longer runs; the workgroups beyond them write
// their zero partial sums and leave (the grid, which the consumers of the partial sums know, stays).  The masks are indexed by
// (tile, plane), not by run: any cut works.
template <int EPI, int NP, bool WX, int SF = 0>
static int march_occupancy() {
    static const int occ = [] {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, (const void *)spmv_wdia_march_kernel<EPI, NP, WX, SF>) != hipSuccess) { (void)hipGetLastError(); return 3; }
        return at.numRegs > 128 ? 3 : 4;
    }();
    return occ;
}
static WdMarch march_cut(const sla_csr *A, int occ) {
    WdMarch g = A->wd_mg;
    const int slots = std::max(1, std::min(occ, std::max(1, A->ctx->wd_march_occ))) * A->ctx->n_cu;
    if (g.ntasks <= slots) return g;                       // (already fits: the lowering's cut)
    g.S = std::max(1, std::min(g.planes, slots / g.T));
    g.PS = (g.planes + g.S - 1) / g.S;
    g.S = (g.planes + g.PS - 1) / g.PS;
    g.ntasks = g.T * g.S;
    return g;
}

template <int EPI>
static int launch_epi(const sla_csr *A, const SpmvArgs<int32_t> &a, int grid, int stream_nt) {
    sla_ctx *c = A->ctx;
#define SLA_WDM_LAUNCH(NP_, WX_, SF_)                                                                                                    \
    hipLaunchKernelGGL((spmv_wdia_march_kernel<EPI, NP_, WX_, SF_>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, (const wdm_u64x8s *)A->d_wum_m, \
                       a.x, march_cut(A, march_occupancy<EPI, NP_, WX_, SF_>()), (int32_t)A->row_begin, A->wd_col_lo, A->wd_col_hi + 1, c->xcd_remap, stream_nt, \
                       A->wd_muni)
    constexpr bool kMayWX = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4;
    const bool wx = kMayWX && a.w == a.x + A->row_begin;   // (w holds local rows, x is addressed by global column)
    if (A->wd_muni.n != 5 && A->wd_muni.n != 7) return fail(SLA_ERR_INVALID, "launch_wdia_march: 5 or 7 pairs");
    if (a.fs_ap) {   // K2 folded into K3: the gathered vector is built from r (= a.x) and Ap while the windows are staged
        if constexpr (EPI == EPI_DOT4) {
            if (!a.sc || !a.pa) return fail(SLA_ERR_INVALID, "launch_wdia_march: fused s needs the solver scalars and K1's partial sums");
            if (A->wd_muni.n == 5) SLA_WDM_LAUNCH(5, true, 1);
            else SLA_WDM_LAUNCH(7, true, 1);
            SLA_HIP_TRY(hipGetLastError());
            return SLA_OK;
        }
        if constexpr (EPI == EPI_AXPY_DOT) {   // cgsStep's C2 folded into C3
            if (!a.sc || !a.pa || !a.w) return fail(SLA_ERR_INVALID, "launch_wdia_march: fused u + q needs the solver scalars, C1's partial sums and rhat");
            if (A->wd_muni.n == 5) SLA_WDM_LAUNCH(5, false, 2);
            else SLA_WDM_LAUNCH(7, false, 2);
            SLA_HIP_TRY(hipGetLastError());
            return SLA_OK;
        }
        return fail(SLA_ERR_INVALID, "launch_wdia_march: a fused input vector is defined for the four-sum and the r - alpha A x epilogues only");
    }
    if constexpr (kMayWX) {
        if (wx) {
            if (A->wd_muni.n == 5) SLA_WDM_LAUNCH(5, true, 0);
            else SLA_WDM_LAUNCH(7, true, 0);
        }
    }
    if (!wx) {
        if (A->wd_muni.n == 5) SLA_WDM_LAUNCH(5, false, 0);
        else SLA_WDM_LAUNCH(7, false, 0);
    }
#undef SLA_WDM_LAUNCH
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// the register counts of all instantiations, asked once at lowering time (not inside a stream capture of the first solver step)
template <int EPI>
static void march_prepare_epi() {
    (void)march_occupancy<EPI, 5, false>();
    (void)march_occupancy<EPI, 7, false>();
    if constexpr (EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4) {
        (void)march_occupancy<EPI, 5, true>();
        (void)march_occupancy<EPI, 7, true>();
    }
    if constexpr (EPI == EPI_DOT4) {
        (void)march_occupancy<EPI, 5, true, 1>();
        (void)march_occupancy<EPI, 7, true, 1>();
    }
    if constexpr (EPI == EPI_AXPY_DOT) {
        (void)march_occupancy<EPI, 5, false, 2>();
        (void)march_occupancy<EPI, 7, false, 2>();
    }
}
void wd_march_prepare() {
    march_prepare_epi<EPI_NONE>();
    march_prepare_epi<EPI_DOT>();
    march_prepare_epi<EPI_DOT2>();
    march_prepare_epi<EPI_DOT4>();
    march_prepare_epi<EPI_RES>();
    march_prepare_epi<EPI_AXPY_DOT>();
    march_prepare_epi<EPI_XPBY_NRM>();
    march_prepare_epi<EPI_SUB>();
}

int launch_wdia_march(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid, int stream_nt) {
    switch (epi) {
        case EPI_NONE: return launch_epi<EPI_NONE>(A, a, grid, stream_nt);
        case EPI_DOT: return launch_epi<EPI_DOT>(A, a, grid, stream_nt);
        case EPI_DOT2: return launch_epi<EPI_DOT2>(A, a, grid, stream_nt);
        case EPI_DOT4: return launch_epi<EPI_DOT4>(A, a, grid, stream_nt);
        case EPI_RES: return launch_epi<EPI_RES>(A, a, grid, stream_nt);
        case EPI_AXPY_DOT: return launch_epi<EPI_AXPY_DOT>(A, a, grid, stream_nt);
        case EPI_XPBY_NRM: return launch_epi<EPI_XPBY_NRM>(A, a, grid, stream_nt);
        case EPI_SUB: return launch_epi<EPI_SUB>(A, a, grid, stream_nt);
    }
    return fail(SLA_ERR_INVALID, "launch_wdia_march: unknown epilogue");
}

}  // namespace sla
