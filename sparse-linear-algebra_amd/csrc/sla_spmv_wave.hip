// sla_spmv_wave.hip -- the plain CSR (#>) (f64 values + i32 columns, Data/Sparse/Common.hs:242-260) with WAVEFRONT-PRIVATE row
// blocks and ROW-PAIR stores (round 4).  Same bytes as spmv_stream_kernel, another structure:
//
//   * a block is 128 consecutive rows and belongs to ONE wavefront: its own LDS stage, its own pipeline, no workgroup barrier
//     anywhere in the walk (round 3's probe: the skeleton with wavefront-private blocks ran 5 % faster than the workgroup-level one);
//   * the block's entries go through LDS in chunks of 128 x PPL entries: every lane streams PPL PAIRS of consecutive entries per
//     chunk (one 8-byte column load and one 16-byte value load per pair, lane-strided: coalesced 512 B / 1 KiB wave-instructions),
//     gathers x, and stages the products;
//   * the fold is one lane per row, rows t and t + 64 of the block on lane t (neighbouring lanes read neighbouring rows' products:
//     the LDS access pattern of the lane-per-row fold of spmv_stream_kernel), ascending, one product at a time with FMA contraction
//     off in the sum -- the reference's left fold bit for bit, whatever the row length (a row may span chunks: the lane's running
//     sum continues);
//   * the 128 row sums are then turned into ROW PAIRS through the wave's LDS stage (lane u: rows 2u, 2u + 1) and leave through the
//     pair epilogue of the wave-sliced kernels (wd_epilogue): ONE 16-byte y store and one 16-byte load per epilogue operand per lane
//     -- the y store was the expensive stream of the CSR-stream kernel (80 MB of 8-byte stores = 34 % of its time, DESIGN.md
//     section 4), and 16 bytes per lane is what the vector kernels run at 10.6 B/clk/CU with.
//
// Taken for matrices with 32-bit row pointers whose longest row has <= kWvMaxRow entries (one lane folds a row: long rows belong
// to the segment / wavefront / workgroup reductions of spmv_stream_kernel, which stays the general kernel).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

// Ablation build (tools/build_variant_one.sh abl<N> sla_spmv_wave.hip -DSLA_WV_ABLATE=<N>; round 6): which stage of the
// kernel owns the gap to the streaming ceiling -- bits REMOVE one stage each (the results are then wrong; timing only):
//   1 the x gather (a constant instead), 2 the LDS product stage + lane-per-row fold (lanes add their own products), 4 the row-pair
//   transpose through LDS, 8 the epilogue (y store, operand arithmetic), 16 the epilogue operand loads (w / z).  0 = the product.
#ifndef SLA_WV_ABLATE
#define SLA_WV_ABLATE 0
#endif
constexpr int kWvAbl = SLA_WV_ABLATE;
// Round 6: the prefetching instantiations also fetch the NEXT block's row starts and epilogue operands one block early (behind the next
// chunk's streams).  Before, every block opened with three fresh vector loads (rowptr[r0 + lane], rowptr[r0 + 64 + lane], the operand pair)
// that the row-end shuffle needed at once: `s_waitcnt vmcnt(0)` at the top of the block loop (seen in the ISA) -- a full memory round trip per
// block with nothing of the wavefront in flight behind it, and the previous block's y store drained with it.  -DSLA_WV_PREOPS=0: the old order.
#ifndef SLA_WV_PREOPS
#define SLA_WV_PREOPS 1
#endif
// ... and HOLD A BLOCK'S y STORE BACK until the next block's gathers and prefetch streams have been issued (SLA_WV_DEFER, default on with the
// above).  vmcnt retires in order and counts stores: a store issued at the end of a block stands as the youngest operation in front of the
// wait that opens the next block -- its acknowledgement (a write round trip) was drained there block after block.  Issued BEHIND the next
// block's loads it has a whole block's time to retire before anything waits for it.  (Round 3 tried this on the workgroup-level kernel and
// lost to spills at 64 VGPRs; this instantiation holds 3 workgroups per CU and has the registers.)
#ifndef SLA_WV_DEFER
#define SLA_WV_DEFER 1
#endif

typedef int wv_i32x2 __attribute__((ext_vector_type(2)));
typedef double wv_f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool st_nt_of(int nt) { return (nt & 2) != 0; }

template <int PPL>
struct WvChunk {
    wv_i32x2 cc[PPL];
    wv_f64x2 vv[PPL];
};
// the streaming loads of the chunk [kb, kb + 128 PPL) of a block whose entries end at k1 (clamped, unconditional)
// -DSLA_WV_STREAM_SCOPE=1 / 2 (experiment, round 6): the streams as agent- / system-scope loads (sc1 / sc0 sc1: they miss in the L1 by definition) instead
// of non-temporal ones -- does the L1 then keep the x lines the four wavefronts of a workgroup share?
#ifndef SLA_WV_STREAM_SCOPE
#define SLA_WV_STREAM_SCOPE 0
#endif
template <int PPL>
__device__ __forceinline__ void wv_load(WvChunk<PPL> &c, const int32_t *__restrict__ col, const double *__restrict__ val, int kb, int k1, int lane) {
    const int kmax = max(0, (min(kb + 128 * PPL, k1) - 1) & ~1);     // last pair that holds a valid entry
#if SLA_WV_STREAM_SCOPE
    constexpr auto scope = SLA_WV_STREAM_SCOPE == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const unsigned long long u = __hip_atomic_load((const unsigned long long *)(col + min(kb + 2 * lane + 128 * j, kmax)), __ATOMIC_RELAXED, scope);
        c.cc[j].x = (int)(unsigned)u;
        c.cc[j].y = (int)(unsigned)(u >> 32);
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const double *pv = val + min(kb + 2 * lane + 128 * j, kmax);
        c.vv[j].x = __hip_atomic_load(pv, __ATOMIC_RELAXED, scope);
        c.vv[j].y = __hip_atomic_load(pv + 1, __ATOMIC_RELAXED, scope);
    }
#else
#pragma unroll
    for (int j = 0; j < PPL; ++j) c.cc[j] = __builtin_nontemporal_load((const wv_i32x2 *)(col + min(kb + 2 * lane + 128 * j, kmax)));
#pragma unroll
    for (int j = 0; j < PPL; ++j) c.vv[j] = __builtin_nontemporal_load((const wv_f64x2 *)(val + min(kb + 2 * lane + 128 * j, kmax)));
#endif
}

// The wave kernel loads entries in aligned PAIRS and gathers x for both halves with no range test: the pair that holds the last
// entry of a matrix with an odd entry count reads col[nnz] from the slack behind the array.  Zeroed slack meant x[0] -- harmless on
// one GPU, but on a row slab whose vector is addressed by global column (in-place halo exchange: base = x - first_row) x[0] lies
// first_row doubles in front of the allocation, beyond the 4 MiB guard from 512 K rows on.  The slack now repeats the last
// valid column of THIS matrix.
__global__ void col_slack_fill_kernel(int32_t *col, int64_t nnz) {
    if (nnz > 0 && threadIdx.x < kArraySlack / sizeof(int32_t)) col[nnz + threadIdx.x] = col[nnz - 1];
}
int launch_col_slack_fill(sla_ctx *c, int32_t *d_col, int64_t nnz) {
    if (!d_col || nnz <= 0) return SLA_OK;
    hipLaunchKernelGGL(col_slack_fill_kernel, dim3(1), dim3(64), 0, stream_of(c), d_col, nnz);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// PRE: the streams of the NEXT chunk (of this block or of the wavefront's next block) are issued before the current chunk is folded
// (a second register set: fewer wavefronts per CU, more bytes in flight per wavefront)
// (the two-pair instantiations of the three-operand epilogues are compiled for 7 workgroups per CU instead of 8: at 8 they spilled 14-16
// registers to scratch -- profiles/r04_kernel_resources.txt; the launch keeps its grid, the surplus workgroups queue behind the first round)
template <int EPI, int PPL, int OCC, bool PRE>
__global__ void __launch_bounds__(kBlock, (PPL == 2 && OCC == 8 && (EPI == EPI_DOT4 || EPI == EPI_AXPY_DOT)) ? 7 : OCC)
spmv_wave_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                 const double *__restrict__ xg, int nblk, int xcd_remap, int nt, int rl) {
    constexpr int CH = 128 * PPL;                       // entries per chunk
    __shared__ double s_prod[kBlock / 64][CH];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double coef;
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    double *prod = s_prod[wave];
    // the blocks of this wavefront: XCD x (workgroups b with b % 8 == x) takes the x-th contiguous eighth of the blocks, the four
    // wavefronts of a workgroup neighbouring blocks, so that each private L2 sees one sliding window of x
    // Round 6 (option wave_run): a wavefront walks RUNS of rl consecutive blocks before it jumps -- the x lines a block shares with its
    // successor (matrices whose columns lie near the diagonal: all of a stencil's legs but the far planes) are then re-used by the SAME
    // wavefront one block later instead of being fetched by four different wavefronts at once (rl = 1: the round-4 walk).  The unit
    // of the walk below is the run; `first`, `step`, `last` count runs.
    const int G = (int)gridDim.x;
    const int nrun = (nblk + rl - 1) / rl;
    int first, step, last;
    if (xcd_remap && (G & 7) == 0 && nrun >= 4 * G) {
        const int xcd = (int)blockIdx.x & 7, per = (nrun + 7) >> 3;
        first = xcd * per + ((int)blockIdx.x >> 3) * (kBlock / 64) + wave;
        step = (G >> 3) * (kBlock / 64);
        last = min((xcd + 1) * per, nrun);
    } else {
        first = (int)blockIdx.x * (kBlock / 64) + wave;
        step = G * (kBlock / 64);
        last = nrun;
    }
    constexpr bool kUsesW = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT;
    constexpr bool kUsesZ = EPI == EPI_DOT4 || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM;
    const bool w_nt = nt && a.w != xg, z_nt = nt && (const double *)a.z != xg;
    WvChunk<PPL> cur;
    int nk0 = 0, nk1 = 0;                                   // PRE: entry range of the block whose first chunk `cur` holds
    constexpr bool PREOPS = PRE && SLA_WV_PREOPS != 0;
    constexpr bool DEFER = PREOPS && SLA_WV_DEFER != 0 && !(kWvAbl & 8);
    wv_f64x2 pend_out = {0.0, 0.0};                         // DEFER: the row pair of the previous block that is still to be stored
    int pend_row = -1, pend_where = 0;
    bool pend_vb = false;
    auto flush = [&]() {
        if (pend_row >= 0 && pend_where) wd_store_pair(pend_where == 1 ? a.y : a.z, pend_row, pend_vb, pend_out, st_nt_of(nt));
        pend_row = -1;
    };
    int sa_n = 0, sb_n = 0;                                 // PREOPS: row starts / operand pairs of the block whose first chunk `cur` holds
    wv_f64x2 wv_n = {0.0, 0.0}, zv_n = {0.0, 0.0};
    auto load_ops = [&](int b, int &sa_o, int &sb_o, wv_f64x2 &wv_o, wv_f64x2 &zv_o) {
        const int r = b * 128, pr = min(r + 2 * lane, a.rows - 1);
        sa_o = rowptr[r + lane];
        sb_o = rowptr[r + 64 + lane];
        if constexpr (kUsesW && !(kWvAbl & 16)) {
            if (EPI != EPI_AXPY_DOT || a.w) wv_o = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.w + pr)) : *(const wd_f64x2u *)(a.w + pr);
        }
        if constexpr (kUsesZ && !(kWvAbl & 16)) zv_o = z_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.z + pr)) : *(const wd_f64x2u *)(a.z + pr);
    };
    if constexpr (PRE) {
        if (first < last) {
            nk0 = rowptr[first * rl * 128];
            nk1 = rowptr[first * rl * 128 + 128];
            wv_load<PPL>(cur, col, val, nk0 & ~1, nk1, lane);
            if constexpr (PREOPS) load_ops(first * rl, sa_n, sb_n, wv_n, zv_n);
        }
    }
    const bool st_nt = st_nt_of(nt);
    // Experiment of round 6 (option wave_sync, bit 2 of `nt`): the four wavefronts of a workgroup walk ADJACENT blocks, whose x windows near the
    // diagonal overlap -- but they drift apart, and the L1 fills of x per block (58 lines measured) are those of no sharing at all (54).  A raw
    // s_barrier at the top of every run keeps them within one block of each other so that their requests for the same x lines meet in the L1
    // (in flight or resident).  The trip count is made the workgroup's (wavefront 0's): a wavefront without a run left only meets the barrier.
    const bool wsync = (nt & 4) != 0;
    const int first0 = first - wave;
    const int nit = wsync ? (first0 < last ? (last - first0 + step - 1) / step : 0) : (first < last ? (last - first + step - 1) / step : 0);
    for (int it = 0, run = first; it < nit; ++it, run += step) {
    if (wsync) __builtin_amdgcn_s_barrier();
    if (run < last)
    for (int blk = run * rl, bend = min(run * rl + rl, nblk); blk < bend; ++blk) {
        const int r0 = blk * 128;
        // (rowptr carries 192 entries of padding = nnz behind its rows + 1 entries: every index below is readable and rows past the
        // end of the matrix are empty)
        const int k0 = PRE ? nk0 : rowptr[r0], k1 = PRE ? nk1 : rowptr[r0 + 128];
        int bn = blk;
        if constexpr (PRE) {                                // the next block's range: a scalar load issued a whole block early
            bn = blk + 1 < bend ? blk + 1 : min(run + step, last - 1) * rl;   // (past the last run: any valid block -- its chunk is loaded and dropped)
            nk0 = rowptr[bn * 128];
            nk1 = rowptr[bn * 128 + 128];
        }
        const int prow = r0 + 2 * lane;                   // this lane's row PAIR in the epilogue
        int sa, sb;
        wv_f64x2 wv = {0.0, 0.0}, zv = {0.0, 0.0};
        if constexpr (PREOPS) {
            sa = sa_n; sb = sb_n; wv = wv_n; zv = zv_n;     // fetched a block ago
        } else {
            load_ops(blk, sa, sb, wv, zv);
        }
        // row ends: the next lane's start; lane 63's rows end where rows 64 / 128 of the block start
        const int sb0 = __builtin_amdgcn_readfirstlane(sb);   // (outside the lane test: readfirstlane reads the first ACTIVE lane)
        int ea = __shfl_down(sa, 1, 64), eb = __shfl_down(sb, 1, 64);
        if (lane == 63) {
            ea = sb0;
            eb = k1;
        }
        double ya = 0.0, yb = 0.0;
        for (int kb = k0 & ~1;; kb += CH) {               // chunks: entries [kb, kb + CH) of the (even-aligned) stream
            const int kend = min(kb + CH, k1);
            if constexpr (!PRE) wv_load<PPL>(cur, col, val, kb, k1, lane);
            double xa[PPL], xb[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                if constexpr (kWvAbl & 1) {
                    xa[j] = 1.0 + (double)(cur.cc[j].x & 3);
                    xb[j] = 1.0 + (double)(cur.cc[j].y & 3);
                } else if constexpr (kWvAbl & 32) {   // (bit 32: only the FAR gathers -- columns outside [r0 - 256, r0 + 384) -- are issued: what an ideal, free
                                                      // LDS window for the near-diagonal part of a block would leave)
                    const unsigned da = (unsigned)(cur.cc[j].x - (r0 - 256)), db = (unsigned)(cur.cc[j].y - (r0 - 256));
                    xa[j] = da < 640u ? 1.0 + (double)(da & 3) : xg[cur.cc[j].x];
                    xb[j] = db < 640u ? 1.0 + (double)(db & 3) : xg[cur.cc[j].y];
                } else {
                    xa[j] = xg[cur.cc[j].x];
                    xb[j] = xg[cur.cc[j].y];
                }
            }
            WvChunk<PPL> nxt;
            if constexpr (PRE) {                            // next chunk of this block, else the first chunk of the wavefront's next block
                const bool more = kb + CH < k1;
                wv_load<PPL>(nxt, col, val, more ? kb + CH : (nk0 & ~1), more ? k1 : nk1, lane);
                if constexpr (PREOPS) {
                    if (!more) load_ops(bn, sa_n, sb_n, wv_n, zv_n);   // the next block's row starts and operands ride behind its first chunk
                }
                if constexpr (DEFER) flush();                          // the previous block's y store: the youngest operation, nobody waits for it soon
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                wv_f64x2 p;
                p.x = cur.vv[j].x * xa[j];
                p.y = cur.vv[j].y * xb[j];
                if constexpr (kWvAbl & 2) {
                    ya += p.x;
                    yb += p.y;
                } else {
                    *(wv_f64x2 *)(prod + 2 * lane + 128 * j) = p;
                }
            }
            if constexpr (PRE) cur = nxt;
            if constexpr (!(kWvAbl & 2)) {   // one lane per row, ascending, one product at a time: the reference's left fold (the products are rounded, the sum adds them).
                // Eight products of each of the lane's two rows are READ together (clamped addresses, all sixteen reads in flight) and then
                // added in order under their range tests: one LDS round trip per eight entries instead of one per entry.
                int ka = max(sa, kb) - kb, kb2 = max(sb, kb) - kb;
                const int ha = min(ea, kend) - kb, hb = min(eb, kend) - kb;
                while (ka < ha || kb2 < hb) {
                    double pa[8], pb[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pa[i] = prod[min(ka + i, CH - 1)];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pb[i] = prod[min(kb2 + i, CH - 1)];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (ka + i < ha) ya += pa[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (kb2 + i < hb) yb += pb[i];
                    ka += 8;
                    kb2 += 8;
                }
            }
            if (kb + CH >= k1) break;
        }
        // rows (t, t + 64) per lane -> row pairs (2u, 2u + 1) per lane through the wave's stage (in order behind the fold's reads)
        wv_f64x2 yp = {ya, yb};
        if constexpr (!(kWvAbl & 4)) {
            prod[lane] = ya;
            prod[64 + lane] = yb;
            yp = *(const wv_f64x2 *)(prod + 2 * lane);
        }
        if constexpr (kWvAbl & 8) {
            acc1 += yp.x + yp.y + wv.x + zv.y;          // (keeps everything above alive)
        } else if (prow < a.rows) {
            if constexpr (DEFER) {
                pend_vb = prow + 1 < a.rows;
                pend_where = wd_epilogue_calc<EPI>(a, pend_vb, yp.x, yp.y, wv, zv, coef, acc1, acc2, pend_out);
                pend_row = prow;
            } else {
                wd_epilogue<EPI>(a, prow, prow + 1 < a.rows, yp.x, yp.y, wv, zv, coef, acc1, acc2, st_nt);
            }
        }
    }
    }
    if constexpr (DEFER) flush();
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel with the one-deep ring UNROLLED BY TWO over two register sets (round 6, option wave_flat).  spmv_wave_kernel's prefetching
// instantiation ends every chunk with `cur = nxt`: a register copy that needs EVERY load of the next chunk, i.e. a drain of the wavefront's
// whole memory pipeline (the held-back y store included) once per chunk.  Here the walk is flattened into ONE sequence of chunks -- block
// begin (row starts, row ends, the next block's scalar range) and block end (row-pair transpose, epilogue, pending store) are wave-uniform
// branches inside the chunk step -- and the step is instantiated twice, (A, B) and (B, A): the next chunk's streams land in the set the
// following step reads, nothing is copied, and the only waits left are the ones data flow needs (the gathers before the products, the next
// chunk's columns before its gathers).  Same loads, same products, same fold order, same epilogue: bit-identical to spmv_wave_kernel.
// MEASURED (profiles/r06_ab_wave_flat.txt, same box, three interleaved passes): K1 213-216 us against 201-208 us for the nested loops -- slower, as round 3's
// three-stage ring was: with the drains gone a wavefront keeps more in flight, and that only lengthens the queues of a memory path that is already
// saturated by this access mix.  Kept as an A/B knob (wave_flat = 1), off by default.
template <int EPI, int PPL, int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
spmv_wave_flat_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                      const double *__restrict__ xg, int nblk, int xcd_remap, int nt, int rl, double *__restrict__ dump) {
    constexpr int CH = 128 * PPL;
    __shared__ double s_prod[kBlock / 64][CH];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double coef;
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    double *prod = s_prod[wave];
    const int G = (int)gridDim.x;
    const int nrun = (nblk + rl - 1) / rl;
    int first, step, last;
    if (xcd_remap && (G & 7) == 0 && nrun >= 4 * G) {
        const int xcd = (int)blockIdx.x & 7, per = (nrun + 7) >> 3;
        first = xcd * per + ((int)blockIdx.x >> 3) * (kBlock / 64) + wave;
        step = (G >> 3) * (kBlock / 64);
        last = min((xcd + 1) * per, nrun);
    } else {
        first = (int)blockIdx.x * (kBlock / 64) + wave;
        step = G * (kBlock / 64);
        last = nrun;
    }
    constexpr bool kUsesW = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT;
    constexpr bool kUsesZ = EPI == EPI_DOT4 || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM;
    const bool w_nt = nt && a.w != xg, z_nt = nt && (const double *)a.z != xg;
    const bool st_nt = st_nt_of(nt);
    auto load_ops = [&](int b, int &sa_o, int &sb_o, wv_f64x2 &wv_o, wv_f64x2 &zv_o) {
        const int r = b * 128, pr = min(r + 2 * lane, a.rows - 1);
        sa_o = rowptr[r + lane];
        sb_o = rowptr[r + 64 + lane];
        if constexpr (kUsesW) {
            if (EPI != EPI_AXPY_DOT || a.w) wv_o = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.w + pr)) : *(const wd_f64x2u *)(a.w + pr);
        }
        if constexpr (kUsesZ) zv_o = z_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.z + pr)) : *(const wd_f64x2u *)(a.z + pr);
    };
    // The pending store is ONE unconditional 16-byte store per step on every path: lanes with nothing to store (no block yet, rows past the end,
    // the single last row of an odd row count) write their pair to a 1 KiB dump of the context's scratch instead.  A store inside a divergent
    // branch makes the number of outstanding operations path-dependent, and the compiler then waits with vmcnt(0) -- for the store as well --
    // wherever it needs one of the loads in front of it; with the same operations on every path it counts.  (The odd last row, if any, is an
    // 8-byte store of its own behind a wave-uniform test.)
    wv_f64x2 pend_out = {0.0, 0.0};
    int pend_row = -1, pend_where = 0;
    bool pend_vb = false;
    constexpr bool kStores = EPI != EPI_RES;
    auto flush = [&]() {
        if constexpr (kStores) {
            const bool live = pend_row >= 0 && pend_where != 0;
            double *base = pend_where == 2 ? a.z : a.y;
            double *dst = (live && pend_vb) ? base + pend_row : dump + 2 * lane;
            if (st_nt) __builtin_nontemporal_store(pend_out, (wv_f64x2 *)dst);
            else *(wv_f64x2 *)dst = pend_out;
            if (__builtin_amdgcn_ballot_w64(live && !pend_vb) != 0) {
                if (live && !pend_vb) base[pend_row] = pend_out.x;
            }
        }
        pend_row = -1;
    };
    bool have = first < last;
    if (have) {
        // the walk's state: the block under way (run, blk, bend), its entry range [k0, k1) and the chunk [kb, kb + CH) the current set holds;
        // the block after it (bn: -1 = none) with its scalar range and, from its first chunk's step on, its row starts and operands
        int run = first, blk = first * rl, bend = min(first * rl + rl, nblk);
        // (scalars: the walk's control state must live in SGPRs -- a VGPR copy of k0 made the compiler wait for EVERY outstanding load at the loop
        // header before it could compare it)
        int k0 = __builtin_amdgcn_readfirstlane(rowptr[blk * 128]), k1 = __builtin_amdgcn_readfirstlane(rowptr[blk * 128 + 128]), kb = k0 & ~1;
        int bn = -1, bn_run = 0, nk0 = 0, nk1 = 0;
        int sa = 0, sb = 0, ea = 0, eb = 0, sa_n = 0, sb_n = 0, prow = 0;
        wv_f64x2 wv = {0.0, 0.0}, zv = {0.0, 0.0}, wv_n = {0.0, 0.0}, zv_n = {0.0, 0.0};
        double ya = 0.0, yb = 0.0;
        WvChunk<PPL> A, B;
        wv_load<PPL>(A, col, val, kb, k1, lane);
        load_ops(blk, sa_n, sb_n, wv_n, zv_n);
        flush();   // (nothing pending: a store into the dump -- the loop is entered with the operation sequence a step leaves behind, so the
                   // compiler's wait counts at the loop header are the same from both sides)
        auto chunk_step = [&](WvChunk<PPL> &cur, WvChunk<PPL> &nxt) {
            if (kb == (k0 & ~1)) {                              // ---- block begin (wave-uniform) ----
                sa = sa_n; sb = sb_n; wv = wv_n; zv = zv_n;
                prow = blk * 128 + 2 * lane;
                const int sb0 = __builtin_amdgcn_readfirstlane(sb);
                ea = __shfl_down(sa, 1, 64);
                eb = __shfl_down(sb, 1, 64);
                if (lane == 63) { ea = sb0; eb = k1; }
                ya = yb = 0.0;
                if (blk + 1 < bend) { bn = blk + 1; bn_run = run; }
                else if (run + step < last) { bn = (run + step) * rl; bn_run = run + step; }
                else { bn = -1; bn_run = run; }
                const int bq = bn >= 0 ? bn : blk;              // (no next block: any valid one -- its loads are issued and dropped)
                nk0 = __builtin_amdgcn_readfirstlane(rowptr[bq * 128]);
                nk1 = __builtin_amdgcn_readfirstlane(rowptr[bq * 128 + 128]);
            }
            const int kend = min(kb + CH, k1);
            double xa[PPL], xb[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                xa[j] = xg[cur.cc[j].x];
                xb[j] = xg[cur.cc[j].y];
            }
            const bool more = kb + CH < k1;
            wv_load<PPL>(nxt, col, val, more ? kb + CH : (nk0 & ~1), more ? k1 : nk1, lane);
            // (the row starts and operands of the block the NEXT chunk belongs to: the next block's behind the last chunk, this block's own again
            // otherwise -- three small loads that keep the operation count of a step the same on every path, see flush())
            load_ops(more ? blk : (bn >= 0 ? bn : blk), sa_n, sb_n, wv_n, zv_n);
            flush();                                            // the previous block's y store: behind this chunk's gathers and the next one's streams
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                wv_f64x2 p;
                p.x = cur.vv[j].x * xa[j];
                p.y = cur.vv[j].y * xb[j];
                *(wv_f64x2 *)(prod + 2 * lane + 128 * j) = p;
            }
            {   // the lane-per-row fold of spmv_wave_kernel, unchanged
                int ka = max(sa, kb) - kb, kb2 = max(sb, kb) - kb;
                const int ha = min(ea, kend) - kb, hb = min(eb, kend) - kb;
                while (ka < ha || kb2 < hb) {
                    double pa[8], pb[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pa[i] = prod[min(ka + i, CH - 1)];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pb[i] = prod[min(kb2 + i, CH - 1)];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (ka + i < ha) ya += pa[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (kb2 + i < hb) yb += pb[i];
                    ka += 8;
                    kb2 += 8;
                }
            }
            if (more) {
                kb += CH;
                return;
            }
            // ---- block end (wave-uniform): row pairs through the stage, epilogue arithmetic, the store left pending ----
            prod[lane] = ya;
            prod[64 + lane] = yb;
            const wv_f64x2 yp = *(const wv_f64x2 *)(prod + 2 * lane);
            if (prow < a.rows) {
                pend_vb = prow + 1 < a.rows;
                pend_where = wd_epilogue_calc<EPI>(a, pend_vb, yp.x, yp.y, wv, zv, coef, acc1, acc2, pend_out);
                pend_row = prow;
            }
            if (bn < 0) {
                have = false;
                return;
            }
            if (bn_run != run) { run = bn_run; bend = min(run * rl + rl, nblk); }
            blk = bn;
            k0 = nk0;
            k1 = nk1;
            kb = k0 & ~1;
        };
        // (the first step is peeled: the loop header is then reached from the end of a step on both sides -- entry and back edge -- and the
        // compiler's wait counts there are those of the steady state; entered from the set-up code it fell back to vmcnt(0))
        chunk_step(A, B);
        while (have) {
            chunk_step(B, A);
            if (!have) break;
            chunk_step(A, B);
        }
        flush();
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

// ---------------------------------------------------------------------------------------------------------------------------------
// CONTINUOUS CHUNKS (round 6, option wave_cc): the chunk walk decoupled from the blocks.  In spmv_wave_kernel every 128-row block starts a chunk of
// its own, so a block of E entries costs ceil(E / CH) chunk iterations -- 4 for the 3175 entries of a block of the FEM-shaped matrix (e05r0000 tiled),
// the last one nearly empty, and an iteration costs about the same full or empty (measured: 512-entry chunks, 7 per block, run 881 us where 1024-entry
// ones, 4 per block, run 695).  Here a wavefront walks RUNS of `rl` consecutive blocks and streams the run's entries -- contiguous in the CSR
// arrays -- in back-to-back chunks: E_run / CH iterations, one partial chunk per RUN.  A chunk may hold the end of one block and the start of the
// next (or several short blocks): the fold loops over the blocks that intersect it; a block that ends inside the chunk is finished on the spot
// (row-pair transpose through its own LDS words, epilogue, store) and the next one takes over the lanes -- its row starts and operands were fetched
// while its predecessor was being folded.  Same loads, same products, same per-row fold order, same epilogue: bit-identical to spmv_wave_kernel.
template <int EPI, int PPL, int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
spmv_wave_cc_kernel(SpmvArgs<int32_t> a, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                    const double *__restrict__ xg, int nblk, int xcd_remap, int nt, int rl) {
    constexpr int CH = 128 * PPL;
    __shared__ double s_prod[kBlock / 64][CH];
    __shared__ double s_tr[kBlock / 64][128];             // the row-pair transpose of a block that ends inside a chunk (the products stay where they are)
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double coef;
    if (!spmv_prologue<EPI, int32_t>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    double *prod = s_prod[wave], *tr = s_tr[wave];
    const int G = (int)gridDim.x;
    const int nrun = (nblk + rl - 1) / rl;
    int first, step, last;
    if (xcd_remap && (G & 7) == 0 && nrun >= 4 * G) {
        const int xcd = (int)blockIdx.x & 7, per = (nrun + 7) >> 3;
        first = xcd * per + ((int)blockIdx.x >> 3) * (kBlock / 64) + wave;
        step = (G >> 3) * (kBlock / 64);
        last = min((xcd + 1) * per, nrun);
    } else {
        first = (int)blockIdx.x * (kBlock / 64) + wave;
        step = G * (kBlock / 64);
        last = nrun;
    }
    constexpr bool kUsesW = EPI == EPI_DOT || EPI == EPI_DOT2 || EPI == EPI_DOT4 || EPI == EPI_RES || EPI == EPI_SUB || EPI == EPI_AXPY_DOT;
    constexpr bool kUsesZ = EPI == EPI_DOT4 || EPI == EPI_AXPY_DOT || EPI == EPI_XPBY_NRM;
    const bool w_nt = nt && a.w != xg, z_nt = nt && (const double *)a.z != xg;
    const bool st_nt = st_nt_of(nt);
    auto load_ops = [&](int b, int &sa_o, int &sb_o, wv_f64x2 &wv_o, wv_f64x2 &zv_o) {
        const int r = b * 128, pr = min(r + 2 * lane, a.rows - 1);
        sa_o = rowptr[r + lane];
        sb_o = rowptr[r + 64 + lane];
        if constexpr (kUsesW) {
            if (EPI != EPI_AXPY_DOT || a.w) wv_o = w_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.w + pr)) : *(const wd_f64x2u *)(a.w + pr);
        }
        if constexpr (kUsesZ) zv_o = z_nt ? __builtin_nontemporal_load((const wd_f64x2u *)(a.z + pr)) : *(const wd_f64x2u *)(a.z + pr);
    };
    WvChunk<PPL> cur;
    // the run whose first chunk `cur` holds, fetched a run early: its block range, its entry range, the row starts / operands of its first block
    int nb0 = 0, nb1 = 0, nkA = 0, nkB = 0;
    int sa_n = 0, sb_n = 0;
    wv_f64x2 wv_n = {0.0, 0.0}, zv_n = {0.0, 0.0};
    auto run_range = [&](int run, int &b0, int &b1, int &kA, int &kB) {
        b0 = run * rl;
        b1 = min(run * rl + rl, nblk);
        kA = __builtin_amdgcn_readfirstlane(rowptr[b0 * 128]);
        kB = __builtin_amdgcn_readfirstlane(rowptr[b1 * 128]);   // (rows past the end are empty: rowptr is padded with nnz)
    };
    if (first < last) {
        run_range(first, nb0, nb1, nkA, nkB);
        wv_load<PPL>(cur, col, val, nkA & ~1, nkB, lane);
        load_ops(nb0, sa_n, sb_n, wv_n, zv_n);
    }
    for (int run = first; run < last; run += step) {
        const int b1 = nb1, kA = nkA, kB = nkB;
        int blk = nb0;
        // the block under way: its entries end at k1, this lane's rows blk * 128 + lane / + 64 + lane with starts sa / sb and ends ea / eb
        int k1 = __builtin_amdgcn_readfirstlane(rowptr[min(blk + 1, b1) * 128]);   // (the block's entries end here; they start where the lanes' row starts say)
        int sa = sa_n, sb = sb_n;
        wv_f64x2 wv = wv_n, zv = zv_n;
        auto row_ends = [&](int &ea_o, int &eb_o) {
            const int sb0 = __builtin_amdgcn_readfirstlane(sb);
            ea_o = __shfl_down(sa, 1, 64);
            eb_o = __shfl_down(sb, 1, 64);
            if (lane == 63) { ea_o = sb0; eb_o = k1; }
        };
        int ea, eb;
        row_ends(ea, eb);
        double ya = 0.0, yb = 0.0;
        // the block after it inside the run (its row starts and operands: loaded while this one is folded); the next run's first block otherwise
        const int run_n = min(run + step, last - 1);                       // (past the last run: any valid run -- loaded and dropped)
        run_range(run_n, nb0, nb1, nkA, nkB);
        load_ops(blk + 1 < b1 ? blk + 1 : nb0, sa_n, sb_n, wv_n, zv_n);
        for (int kb = kA & ~1;; kb += CH) {
            const int kend = min(kb + CH, kB);
            double xa[PPL], xb[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                xa[j] = xg[cur.cc[j].x];
                xb[j] = xg[cur.cc[j].y];
            }
            WvChunk<PPL> nxt;
            const bool more = kb + CH < kB;
            wv_load<PPL>(nxt, col, val, more ? kb + CH : (nkA & ~1), more ? kB : nkB, lane);
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                wv_f64x2 p;
                p.x = cur.vv[j].x * xa[j];
                p.y = cur.vv[j].y * xb[j];
                *(wv_f64x2 *)(prod + 2 * lane + 128 * j) = p;
            }
            cur = nxt;
            for (;;) {   // the blocks that intersect the chunk [kb, kend)
                {   // this block's share of the chunk: the lane-per-row fold of spmv_wave_kernel on [max(k0, kb), min(k1, kend))
                    const int lim = min(k1, kend);
                    int ka = max(sa, kb) - kb, kb2 = max(sb, kb) - kb;
                    const int ha = min(ea, lim) - kb, hb = min(eb, lim) - kb;
                    while (ka < ha || kb2 < hb) {
                        double pa[8], pb[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) pa[i] = prod[min(ka + i, CH - 1)];
#pragma unroll
                        for (int i = 0; i < 8; ++i) pb[i] = prod[min(kb2 + i, CH - 1)];
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (ka + i < ha) ya += pa[i];
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (kb2 + i < hb) yb += pb[i];
                        ka += 8;
                        kb2 += 8;
                    }
                }
                if (k1 > kend) break;                 // the block goes on in the next chunk
                // ---- the block ends inside this chunk: row pairs, epilogue, store ----
                tr[lane] = ya;
                tr[64 + lane] = yb;
                const wv_f64x2 yp = *(const wv_f64x2 *)(tr + 2 * lane);
                const int prow = blk * 128 + 2 * lane;
                if (prow < a.rows) wd_epilogue<EPI>(a, prow, prow + 1 < a.rows, yp.x, yp.y, wv, zv, coef, acc1, acc2, st_nt);
                if (++blk >= b1) break;               // the run is done (kend == kB)
                // ---- the next block of the run takes over the lanes ----
                k1 = __builtin_amdgcn_readfirstlane(rowptr[min(blk + 1, b1) * 128]);
                sa = sa_n; sb = sb_n; wv = wv_n; zv = zv_n;
                row_ends(ea, eb);
                ya = yb = 0.0;
                load_ops(blk + 1 < b1 ? blk + 1 : nb0, sa_n, sb_n, wv_n, zv_n);
            }
            if (!more) break;
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

bool wave_on(const sla_csr *A) {
    const sla_ctx *c = A->ctx;
    return c->stream_wave > 0 && !A->rp64 && A->rows > 0 && A->max_row_nnz <= kWvMaxRow && c->spmv_algo == 0;
}
// Variant = (entry pairs per lane and chunk, workgroups per CU, next-chunk prefetch).  Option stream_wave: 1 = automatic; otherwise
// the decimal code  PRE * 1000 + OCC * 100 + PPL  of one of the instantiations below (A/B runs, tools/wave_ab.py).
struct WvVariant { int ppl, occ, pre; };
// Measured on the 216^3 Laplacian as plain CSR (7 entries per row: 896 per block; same box, two interleaved runs each,
// profiles/r04_wave_variants.txt): spmv_stream_kernel K1 238 us | (2,8) 240 | (4,6) 222 | (6,5) 245 (768-entry chunks cut a block 768 + 128) |
// (8,4) 218 | (8,3) 218 | (12,3) 240 | (16,2) 285 | (8,3)+prefetch 212; on a faster box stream 222 | (4,8) 206 | (7,5) 206 | (4,6) 201 |
// (8,4) 197.  Few, fat wavefronts with a whole block in flight win, like in the tile kernel; the random 1 M-row matrix (33 per row, L2-gather
// bound) is within +-1.5 % on all of them.
static const WvVariant kWvVariants[] = {{2, 8, 0}, {4, 6, 0}, {8, 4, 0}, {8, 3, 1}};
static WvVariant wave_variant(const sla_csr *A, int epi) {
    const int code = A->ctx->stream_wave;
    if (code > 1)
        for (const WvVariant &v : kWvVariants)
            if (v.pre * 1000 + v.occ * 100 + v.ppl == code) return v;
    (void)epi;
    // automatic: the chunk that holds an average block -- 2 / 4 entry pairs per lane for short rows (the loads of a chunk are issued
    // unconditionally: a chunk much larger than the block only issues clamped loads), else 8 pairs with the next chunk prefetched
    const int64_t nblk = (A->rows + 127) / 128, avg = nblk > 0 ? A->nnz / nblk : 0;
    // Short launches of single-chunk blocks (round 6, tools/wave_variant_small_ab.py, profiles/r06_ab_wave_variant_small.txt): with <= ~5 blocks per
    // wavefront the cross-block prefetch has little to prefetch for, a fourth workgroup per CU buys more -- 7-per-row Laplacian as plain CSR, K1 same
    // box (8, 4) against (8, 3, prefetch): 262 k rows 7.8 / 9.2 us, 1 M 22.4 / 24.2, 2 M 44.4 / 45.6, 4 M 88.3 / 87.7, 6.9 M 138.6 / 140.2 (noise);
    // blocks of several chunks (e05r0000 tiled, 3176 per block, 1 M rows) keep the prefetch: 79.7 / 76.8.
    if (avg > 512 && avg <= 1024 && nblk <= (int64_t)64 * A->ctx->n_cu) return kWvVariants[2];
    return avg <= 256 ? kWvVariants[0] : avg <= 512 ? kWvVariants[1] : kWvVariants[3];
}
int wave_grid(const sla_csr *A) {
    const int64_t nblk = (A->rows + 127) / 128;
    // (one grid for every epilogue of a matrix: the consumers of the fused partial sums know spmv_grid(A); an instantiation that holds
    // fewer workgroups per CU than that runs a partial second round)
    const int occ = wave_variant(A, EPI_NONE).occ;
    const int over = A->ctx->wave_over > 0 ? A->ctx->wave_over : 1;
    int64_t g = std::min<int64_t>((nblk + 3) / 4, (int64_t)over * occ * A->ctx->n_cu);
    g = std::min<int64_t>(g, A->ctx->spmv_grid_max);
    if (g >= 8) g &= ~7;                                // a multiple of 8: one share per XCD
    return (int)std::max<int64_t>(1, g);
}

template <int EPI>
static int launch_wave_t(const sla_csr *A, const SpmvArgs<int32_t> &a, int grid) {
    sla_ctx *c = A->ctx;
    const int nblk = (int)((A->rows + 127) / 128);
    const int nt = (vec_stream_nt(c, A->rows) ? 1 : 0) | (c->wd_nt_store ? 2 : 0) | (c->wave_sync ? 4 : 0);
    const WvVariant v = wave_variant(A, EPI);
#define SLA_WV(P, O, R)                                                                                                                    \
    if (v.ppl == P && v.occ == O && v.pre == R)                                                                                            \
        SLA_KLAUNCH(c, (spmv_wave_kernel<EPI, P, O, R != 0>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.x, nblk, \
                           c->xcd_remap, nt, std::max(1, c->wave_run))
    if (v.ppl == 8 && v.occ == 3 && v.pre == 1 && c->wave_cc)
        SLA_KLAUNCH(c, (spmv_wave_cc_kernel<EPI, 8, 3>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.x, nblk, c->xcd_remap, nt,
                    std::max(1, c->wave_cc));   // (wave_cc = the run length: blocks whose entries are streamed as one sequence of chunks)
    else if (v.ppl == 8 && v.occ == 3 && v.pre == 1 && c->wave_flat)
        SLA_KLAUNCH(c, (spmv_wave_flat_kernel<EPI, 8, 3>), dim3(grid), dim3(kBlock), 0, stream_of(c), a, a.rowptr, a.col, a.val, a.x, nblk, c->xcd_remap, nt,
                    std::max(1, c->wave_run), c->d_result + 2048);   // (128 doubles of the context's scratch: the dump of the unconditional store)
    else SLA_WV(2, 8, 0);
    else SLA_WV(4, 6, 0);
    else SLA_WV(8, 4, 0);
    else SLA_WV(8, 3, 1);
#undef SLA_WV
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

int launch_spmv_wave(const sla_csr *A, int epi, const SpmvArgs<int32_t> &a, int grid) {
    switch (epi) {
        case EPI_NONE: return launch_wave_t<EPI_NONE>(A, a, grid);
        case EPI_DOT: return launch_wave_t<EPI_DOT>(A, a, grid);
        case EPI_DOT2: return launch_wave_t<EPI_DOT2>(A, a, grid);
        case EPI_DOT4: return launch_wave_t<EPI_DOT4>(A, a, grid);
        case EPI_RES: return launch_wave_t<EPI_RES>(A, a, grid);
        case EPI_AXPY_DOT: return launch_wave_t<EPI_AXPY_DOT>(A, a, grid);
        case EPI_XPBY_NRM: return launch_wave_t<EPI_XPBY_NRM>(A, a, grid);
        case EPI_SUB: return launch_wave_t<EPI_SUB>(A, a, grid);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_wave: unknown epilogue");
}

}  // namespace sla
