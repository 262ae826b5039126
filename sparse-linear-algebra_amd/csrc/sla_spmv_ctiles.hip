// sla_spmv_ctiles.hip -- (#>) on CU-WIDE tiles (round 5): the row-slice x column-panel form of sla_spmv_tiles.hip with the slice
// owned by the whole workgroup instead of one wavefront, and the row sums added in RELAXED order.
// Reference semantics: Data/Sparse/Common.hs:242-260; contract: |dy_i| <= nnz_i eps sum_j |a_ij x_j|.
//
// Why.  With x in the L2 the tile kernel of rounds 2-4 ran at the L2's REQUEST rate: every 8-byte gather fetched its own 128-byte
// line (181 G gathers/s, 1.02 requests per entry; profiles/r04_pmc_l1_l2_tile_kernel_10m.txt).  tools/gather_share_probe.cpp
// (profiles/r05_gather_share_probe.txt) shows what lifts that: when the 64 lanes of ONE gather instruction hold 64 CONSECUTIVE
// entries of a column-sorted run with d entries per line, neighbouring lanes share lines and the L1 asks the L2 once per line --
// 218 (random) -> 307 (d = 1) -> 393 (d = 2) -> 485 (d = 4) G gathers/s.  d = rows of the slice x entries per row x 16 / columns, so
// the slice has to be as tall as the LDS allows: all 19584 row sums of the CU in ONE array shared by its four wavefronts (d = 1.03
// on BASELINE config 3a, 31 at 100 entries per row and 1 M columns; a wavefront-private slice of 4896 rows has a quarter of that).
//
// The price: a row's products now come from different wavefronts at different times.  An order-preserving version of this kernel was
// built and measured (layer-0 entries shared, later layers by row-owning wavefronts, a workgroup barrier in front of every phase:
// bit-exact, 1.83 ms on config 3a against 1.85 for the wavefront-private form -- the barriers and the unshared second phase eat the
// gain; profiles/r05_ab_ctile_kernel.txt) and dropped.  This kernel adds the products with LDS floating-point atomics (ds_add_f64)
// in whatever order the wavefronts reach them: every product is still rounded separately (no FMA), a row's sum is the same set of
// numbers as the reference's fold, added in another order -- within nnz_i eps sum_j |a_ij x_j| of it, and NOT reproducible bit for
// bit from run to run.  That is why this order is an OPT-IN (tile_relaxed = 1) since the end of round 6: the default deals a tile's column-sorted run so
// that every row belongs to one wavefront (row-owned layout, same kernel: in-order LDS adds of one wavefront are the reference's left fold, bit for bit and
// reproducible), at 1.62 ms against this order's 1.35 on config 3a.
//
// Layout (sla_lower_tiles.cpp / sla_tiles_build.hip): inside a tile (slice x panel) the entries are sorted by column and dealt to the
// four wavefronts in 64-entry groups round-robin (group g -> wavefront g & 3); stored [slice][wavefront][panel], 12 B each:
// the value and (row - slice_row0) << shift | (col - panel * 2^shift)   (15 + 17 bits);
// ctoff[(s * 4 + w) * (P + 1) + j] = first entry of wavefront w's share of tile (s, j), relative to the slice's first entry.
// Panel pacing, the look-ahead poll and the three-chunk software pipeline are those of sla_spmv_tiles.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

#ifndef SLA_CT_U
#define SLA_CT_U 20     // 64-entry groups per chunk: 1280 gathers in flight per wavefront + the next chunk's streams (config 3a: 8: 1.49 ms, 12: 1.47, 16: 1.40, 20: 1.33; 23 is the last that fits 256 + 256 registers)
#endif
#ifndef SLA_CT_CROSS
#define SLA_CT_CROSS 1  // chunks run over one tile boundary
#endif
#ifndef SLA_CT_VAL_LATE
#define SLA_CT_VAL_LATE 0   // (1: value loads issued with the gathers -- deeper chunks fit the registers, but the fold then waits for HBM: 1.60 ms against 1.34 at 20 groups)
#endif
#ifndef SLA_CT_SPIN
#define SLA_CT_SPIN 2000
#endif
#ifndef SLA_CT_U_DENSE
#define SLA_CT_U_DENSE 12   // ... and for tiles with many entries per x line (d >= 4: the gathers are nearly free there and the launch is bound by the
                            // streams -- the deep chunks' register shuffling through the AGPRs costs more than their depth gains: 200 per row at
                            // 1 M rows 464 us with 12 groups, 572 with 20)
#endif

template <int U>
struct CtChunk {
    uint32_t idx[U];
    double val[U];
    double xv[U];
    uint32_t start;   // its first entry, relative to the slice's
    int cnt;      // entries of this chunk (<= 64 * kCtU)
    int split;    // its entries [0, split) belong to tile `panel`, the others to `panel2`
    int panel, panel2;
};

// LDS only: the streams and gathers of the next chunks stay in flight across it
__device__ __forceinline__ void ct_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int U>
__device__ __forceinline__ void ct_fold_chunk(double *yl, const CtChunk<U> &c, int shift, int lane) {
#pragma clang fp contract(off)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cg = c.cnt - 64 * u;                 // valid lanes of this group (wavefront-uniform)
        if (cg <= 0) break;
        const double p = c.val[u] * c.xv[u];           // (separately rounded: the sum below is an add, never an FMA)
        if (lane < cg) unsafeAtomicAdd(yl + (c.idx[u] >> shift), p);   // ds_add_f64
    }
}

template <int EPI, typename RP, int kCtU>
__global__ void __launch_bounds__(kBlock, 1)
spmv_ctile_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ srow, const uint32_t *__restrict__ toff,
                  const uint32_t *__restrict__ tidx, const double *__restrict__ tval, const double *__restrict__ xg, int S, int P,
                  int shift, unsigned *prog, int slack, const int32_t *__restrict__ vis, int v0, int nv, int dlim, const double *__restrict__ dummy) {
    __shared__ double s_y[kCtRows];
    __shared__ double s_red[4];
    __shared__ int s_prog[kBlock / 64];
    __shared__ int s_poll[kBlock / 64][64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    double *yl = s_y;
    const uint32_t cmask = (1u << shift) - 1u;
    if (tid < kBlock / 64) s_prog[tid] = 0;
    for (int i = tid; i < (kBlock / 64) * 64; i += kBlock) (&s_poll[0][0])[i] = 0;
    __syncthreads();
    // panel pacing (see sla_spmv_tiles.hip): a wavefront does not START panel step q before every workgroup of its XCD has finished
    // issuing step q - slack; progress in plain stores to a per-XCD slot table, polled one step ahead by direct-to-LDS loads
    const int xcd = (int)blockIdx.x & 7;
    const int nwg_xcd = ((int)gridDim.x - xcd + 7) >> 3;
    const int rounds = (S + (int)gridDim.x - 1) / (int)gridDim.x;
    if (vis == nullptr) nv = P;
    using Chunk = CtChunk<kCtU>;
    int *slots = prog ? (int *)prog + xcd * 256 : nullptr;
    int *myslot = slots ? slots + ((int)blockIdx.x >> 3) : nullptr;
    bool pace = slack > 0 && slots != nullptr && nwg_xcd <= 64;
    auto publish = [&](int done) {                               // this wavefront has finished issuing `done` panel steps
        if (!slots) return;
        if (lane == 0) {
            __hip_atomic_store(&s_prog[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int stored = -1;
            for (;;) {   // (store, re-read the minimum, store again until stable: see sla_spmv_tiles.hip)
                asm volatile("" ::: "memory");
                int m = done;
#pragma unroll
                for (int w = 0; w < kBlock / 64; ++w) m = min(m, __hip_atomic_load(&s_prog[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (m == stored) break;
                __hip_atomic_store(myslot, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                stored = m;
            }
        }
    };
    int *pollw = s_poll[wave];
    const unsigned poll_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(&s_poll[wave][0]));
    auto wait_for = [&](int need) {                              // until all workgroups of the XCD have finished `need` steps
        if (!pace || need <= 0) return;
        int v = 0x7fffffff;
        if (lane < nwg_xcd) v = __hip_atomic_load(pollw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        while (__ballot(v < need) != 0) {
            v = 0x7fffffff;
            if (lane < nwg_xcd) v = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(v < need) == 0) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SLA_CT_SPIN) { pace = false; return; }
        }
        {
            const int *src = slots + min(lane, nwg_xcd - 1);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(poll_lds) : "memory");
        }
    };
    for (int round = 0; round < rounds; ++round) {
        const int s = (int)blockIdx.x + round * (int)gridDim.x;   // workgroup-uniform
        if (s >= S) {   // no (more) slices: count as finished with everything
            publish(0x7fffffff);
            break;
        }
        const int r0 = __builtin_amdgcn_readfirstlane(srow[s]), nr = __builtin_amdgcn_readfirstlane(srow[s + 1]) - r0;
        if (a.yinit) {
            for (int r = tid; r < nr; r += kBlock) yl[r] = a.yinit[r0 + r];
        } else {
            for (int r = tid; r < nr; r += kBlock) yl[r] = 0.0;
        }
        RP base = rowptr[r0];
        if constexpr (sizeof(RP) == 4) {
            base = (RP)__builtin_amdgcn_readfirstlane((int)base);
        } else {
            base = (RP)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)base >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)base));
        }
        const uint32_t *tp = toff + ((size_t)s * 4 + (size_t)wave) * (size_t)(P + 1);
        // The ranges of the panel steps come through the SCALAR cache, one step ahead (and the panel of a pass's visiting list two
        // ahead): a per-lane block of offsets re-read every 64 steps, as the wavefront-private kernel keeps it, puts s_waitcnt vmcnt(0)
        // -- the whole load pipeline drained -- on EVERY path through advance(), because the wait-count pass cannot tell the path that
        // skipped the reload from the one that took it.
        int q = -1, pj = 0;                      // panel step, its panel
        uint32_t k = 0, k1 = 0;
        auto panel_of = [&](int qq) -> int { return vis ? vis[v0 + min(qq, nv - 1)] : min(qq, nv - 1); };
        int npj = panel_of(0), pn2 = panel_of(1);
        uint32_t nk = tp[npj], nk1 = tp[npj + 1];
        auto advance = [&]() -> bool {
            while (k >= k1) {
                if (q >= 0) publish(round * nv + q + 1);   // done issuing the tile of panel step q
                ++q;
                if (q >= nv) return false;
                wait_for(round * nv + q - slack + 1);
                pj = npj;
                k = nk;
                k1 = nk1;
                npj = pn2;                                 // step q + 1
                nk = tp[npj];
                nk1 = tp[npj + 1];
                pn2 = panel_of(q + 2);
            }
            return true;
        };
        bool live = true;
        // A chunk holds up to 64 kCtU CONSECUTIVE entries of the wavefront's stream and may run over ONE tile boundary (SLA_CT_CROSS): the
        // entries [0, split) gather from panel, the rest from panel2.  (A wavefront's share of a tile is ~33 groups on config 3a: chunks
        // that stop at every tile end leave a quarter of every third chunk empty.)  The ranges of consecutive panel steps are adjacent in
        // memory when the panels are visited in ascending order; a pass of an overlapped all-gather visits them in the plan's order, and a
        // chunk then ends with its tile.
        auto issue = [&](Chunk &c) {   // precondition k < k1; lanes past the end re-read the last entry
            const uint32_t start = k;
            uint32_t take = min((uint32_t)(64 * kCtU), k1 - k);
            c.panel = pj;
            c.panel2 = pj;
            c.split = (int)take;
            k += take;
#if SLA_CT_CROSS
            if (take < (uint32_t)(64 * kCtU)) {   // the tile ends inside the chunk: go on with the next non-empty one if it follows in memory
                if (!advance()) {
                    live = false;
                } else if (k == start + take) {
                    const uint32_t more = min((uint32_t)(64 * kCtU) - take, k1 - k);
                    c.panel2 = pj;
                    k += more;
                    take += more;
                }
            }
#endif
            c.cnt = (int)take;
            c.start = start;
            const uint32_t *ip = tidx + (base + (RP)start);
#if !SLA_CT_VAL_LATE
            const double *vp = tval + (base + (RP)start);
#endif
#pragma unroll
            for (int u = 0; u < kCtU; ++u) {
                const uint32_t i = (uint32_t)min(lane + 64 * u, c.cnt - 1);
                c.idx[u] = __builtin_nontemporal_load(ip + i);
#if !SLA_CT_VAL_LATE
                c.val[u] = __builtin_nontemporal_load(vp + i);
#endif
            }
        };
        auto gather = [&](Chunk &c) {
            const char *xa = (const char *)(c.panel < 0 ? dummy : xg + ((size_t)c.panel << shift)), *xb = (const char *)(c.panel2 < 0 ? dummy : xg + ((size_t)c.panel2 << shift));
#if SLA_CT_VAL_LATE   // the values are not needed before the fold: their loads go out with the gathers, one pipeline stage later (8 B per entry
            // less to hold per chunk in its first stage: deeper chunks fit the register file)
            const double *vp = (c.cnt ? tval + (base + (RP)c.start) : tval);
#pragma unroll
            for (int u = 0; u < kCtU; ++u) c.val[u] = __builtin_nontemporal_load(vp + (uint32_t)min(lane + 64 * u, c.cnt ? c.cnt - 1 : dlim));
#endif
#pragma unroll
            for (int u = 0; u < kCtU; ++u) {
                const char *b = (lane + 64 * u < c.split) ? xa : xb;
                c.xv[u] = *(const double *)(b + (uint32_t)((c.idx[u] & cmask) << 3));
            }
        };
        auto fold = [&](const Chunk &c) { ct_fold_chunk<kCtU>(yl, c, shift, lane); };
        // three chunks in flight per wavefront; the loop issues the SAME loads on every path (past the slice's last chunk: empty chunks)
        Chunk A, B, C;
        auto next = [&](Chunk &c) {
            if (live && k >= k1 && !advance()) live = false;
            if (live) {
                issue(c);
            } else {
                c.cnt = 0;
                c.start = 0;
                c.split = 0;
                c.panel2 = -1;   // gathers from the matrix's all-zero dummy panel: whatever column bits the re-read entries carry, the address
                c.panel = -1;    // is inside an allocation (panel 0 may lie outside a window-mode slab's guard, a partial last panel ends early)
#pragma unroll
                for (int u = 0; u < kCtU; ++u) {
                    const int i = min(lane + 64 * u, dlim);
                    c.idx[u] = __builtin_nontemporal_load(tidx + i);
#if !SLA_CT_VAL_LATE
                    c.val[u] = __builtin_nontemporal_load(tval + i);
#endif
                }
            }
        };
        ct_barrier();                                       // the sums are initialised
        next(A);
        next(B);
        gather(A);
        for (;;) {
            if (A.cnt == 0) break;
            next(C); gather(B); fold(A);
            if (B.cnt == 0) break;
            next(A); gather(C); fold(B);
            if (C.cnt == 0) break;
            next(B); gather(A); fold(C);
        }
        ct_barrier();                                       // every wavefront's atomics have landed
        for (int r = tid; r < nr; r += kBlock) spmv_epilogue<EPI, RP>(a, r0 + r, yl[r], coef, acc1, acc2);
        ct_barrier();                                       // the next slice's initialisation overwrites the sums
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

int ctiles_grid(const sla_csr *A) { return (int)std::max<int64_t>(1, std::min<int64_t>(A->tl_S, (int64_t)A->ctx->n_cu)); }

template <int EPI, typename RP>
static int launch_ctiles_t(const sla_csr *A, const SpmvLaunch &l) {
    sla_ctx *c = A->ctx;
    SpmvArgs<RP> a{};
    a.rowptr = (const RP *)A->d_rowptr;
    a.col = A->d_col;
    a.val = A->d_val;
    a.x = l.x;
    a.y = l.y;
    a.rows = (int32_t)A->rows;
    a.w = l.w;
    a.z = l.z;
    a.p1 = l.p1;
    a.p2 = l.p2;
    a.p3 = l.p3;
    a.p4 = l.p4;
    a.sc = l.sc;
    a.pres = l.pres;
    a.npres = l.npres;
    a.pres_stride = l.pres_stride;
    a.pa = l.pa;
    a.pb = l.pb;
    a.npa = l.npa;
    a.pa_stride = l.pa_stride;
    a.step_begin = l.step_begin;
    a.yinit = l.tv1 >= 0 ? l.yinit : nullptr;   // (running row sums of the earlier passes of an overlapped all-gather)
    const int32_t *vis = l.tv1 >= 0 ? l.tvis : nullptr;
    const int nv = l.tv1 >= 0 ? l.tv1 - l.tv0 : A->tl_P;
    if (l.tv1 >= 0 && (!vis || l.tv0 < 0 || nv < 1 || l.tv1 > A->tl_P)) return fail(SLA_ERR_INVALID, "launch_spmv_tiles: bad panel pass");
    ProfScope prof(c, l.kernel_id);
    if (A->d_tlprog) SLA_HIP_TRY(hipMemsetAsync(A->d_tlprog, 0, A->tlprog_bytes, stream_of(c)));   // the pacing table of this launch
    // chunk depth by the tiles' density d = entries per 128-byte line of x per slice sweep (see SLA_CT_U_DENSE)
    const double d = (double)A->nnz / (double)std::max<int64_t>(1, A->tl_S) * 16.0 / (double)std::max<int64_t>(1, A->n);
    const bool deep = c->tile_depth == 0 ? d < 4.0 : c->tile_depth == 2;
    if (deep)
        hipLaunchKernelGGL((spmv_ctile_kernel<EPI, RP, SLA_CT_U>), dim3(ctiles_grid(A)), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_tlrow, A->d_tloff,
                           A->d_tlidx, A->d_tlval, l.x, A->tl_S, A->tl_P, A->tl_shift, A->d_tlprog, c->xcd8 == 1 ? c->tile_slack : 0, vis, l.tv0, nv,
                           (int)std::min<int64_t>(64 * SLA_CT_U - 1, A->nnz - 1), A->d_tldummy);
    else
        hipLaunchKernelGGL((spmv_ctile_kernel<EPI, RP, SLA_CT_U_DENSE>), dim3(ctiles_grid(A)), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_tlrow, A->d_tloff,
                           A->d_tlidx, A->d_tlval, l.x, A->tl_S, A->tl_P, A->tl_shift, A->d_tlprog, c->xcd8 == 1 ? c->tile_slack : 0, vis, l.tv0, nv,
                           (int)std::min<int64_t>(64 * SLA_CT_U_DENSE - 1, A->nnz - 1), A->d_tldummy);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

template <typename RP>
static int launch_ctiles_rp(const sla_csr *A, const SpmvLaunch &l) {
    switch (l.epi) {
        case EPI_NONE: return launch_ctiles_t<EPI_NONE, RP>(A, l);
        case EPI_DOT: return launch_ctiles_t<EPI_DOT, RP>(A, l);
        case EPI_DOT2: return launch_ctiles_t<EPI_DOT2, RP>(A, l);
        case EPI_DOT4: return launch_ctiles_t<EPI_DOT4, RP>(A, l);
        case EPI_RES: return launch_ctiles_t<EPI_RES, RP>(A, l);
        case EPI_AXPY_DOT: return launch_ctiles_t<EPI_AXPY_DOT, RP>(A, l);
        case EPI_XPBY_NRM: return launch_ctiles_t<EPI_XPBY_NRM, RP>(A, l);
        case EPI_SUB: return launch_ctiles_t<EPI_SUB, RP>(A, l);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_ctiles: unknown epilogue");
}

int launch_spmv_ctiles(const sla_csr *A, const SpmvLaunch &l) {
    return A->rp64 ? launch_ctiles_rp<int64_t>(A, l) : launch_ctiles_rp<int32_t>(A, l);
}

}  // namespace sla
