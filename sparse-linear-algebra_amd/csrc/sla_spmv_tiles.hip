// sla_spmv_tiles.hip -- (#>) for irregular matrices whose x does not fit one XCD's L2 (BASELINE config 3a: 10 M rows,
// ~33 uniformly random columns per row).  Reference semantics: Data/Sparse/Common.hs:242-260 (ascending left fold per row).
//
// What bounds this case (tools/gather_probe.cpp, profiles/r02_gather_probe.txt): a random 8-byte gather moves a whole
// 128-byte line into the L1.  Served from an L2-resident window (<= 3 MiB) the chip sustains ~225 G gathers/s; from the
// memory-side cache or HBM (x = 80 MB) only ~55 G/s -- 330 M gathers = 1.45 ms against 6 ms.  So the gathers must hit
// the L2, i.e. the matrix has to be walked in column panels of <= 2 MiB of x.  Round 1 did that with P = 26 full passes
// over the rows (y, rowptr re-streamed per pass: 2.7x the algorithmic bytes, PMC).  Here the row sums stay ON CHIP:
//
//   * rows are cut into slices of <= 4896 rows (equal entry counts), one wavefront per slice, its row sums in LDS (38 KiB: four
//     wavefronts fill the CU's 160 KiB; 10 M rows = two rounds of the persistent grid -- at 4096 rows three: 2.04 -> 1.91 ms);
//   * the slice's entries are stored tile-major -- ordered by (panel, row, column), 12 B each: the value and one dword
//     (row - slice_row0) << 18 | (col - panel * 2^18) -- so a wavefront streams them with coalesced non-temporal loads,
//     gathers x from the panel all wavefronts of the XCD are on at that moment, and adds the products into LDS;
//   * inside a tile the entries are ordered by (layer, column), layer = rank of the entry inside its (row, panel) segment in
//     ascending column order: 64 consecutive entries of one layer are 64 different rows, so a group of 64 products goes
//     into the row sums with ONE conflict-free LDS read-modify-write (y_r = y_r + a_rc x_c, separately rounded) and no
//     cross-lane work; a group that spans a layer boundary (bit 31 of the index dword) is split there and done in as many passes.
//     Column order inside a layer (round 4) keeps a wavefront's 64 gathers inside a 30-40 KB stretch of the panel: fewer pages
//     and L2 channels per instruction (tools/gather_locality_probe.cpp: 179 -> 243 G gathers/s on the bare pattern);
//     A row's products are therefore added ONE BY ONE in ascending column order (panels ascending, layers ascending): every
//     row is the reference's left fold bit for bit, whatever its length (an instruction-count matter too: the first
//     version chained the runs of a (row, column)-sorted tile through DPP shifts -- ~300 instructions per 64 entries,
//     issue-bound at 130 G entries/s even with x in the L2; this one needs ~25);
//   * all wavefronts start at panel 0 and carry equal work, so they sweep the panels in near lock step and the XCD's
//     L2 holds the current panel; x itself (80 MB) sits in the 256 MiB memory-side cache between the sweeps;
//   * y is written once per row, in the fused epilogue (dot / dot2 / residual / CGS / CGNE variants): HBM traffic =
//     12 B per entry + the vectors = the algorithmic bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

#ifndef SLA_TILE_U
#define SLA_TILE_U 12
#endif
#ifndef SLA_TILE_SPIN
#define SLA_TILE_SPIN 2000   // polls before a wavefront stops pacing for the rest of the launch
#endif
#ifndef SLA_TILE_UNIFORM
#define SLA_TILE_UNIFORM 1
#endif
constexpr int kTilePollQ = (kTileBlocksPerCu * 32 + 63) / 64;   // 64-slot groups of the look-ahead poll (<= 32 CUs per XCD)
constexpr int kTileU = SLA_TILE_U;   // 64-entry groups per chunk: 768 gathers in flight per wavefront (+ the next chunk's streams)

struct TileChunk {
    uint32_t idx[kTileU];
    double val[kTileU];
    double xv[kTileU];
    int cnt;      // entries of this chunk (<= 64 * kTileU)
    int panel;
};

// Add one chunk (kTileU groups of 64 consecutive entries of one tile; the chunk's entries [0, cnt) are valid, the lanes past
// them hold copies of the last entry) into the slice's row sums yl[].
__device__ __forceinline__ void tile_fold_chunk(double *yl, const TileChunk &c, int shift, int lane) {
#pragma clang fp contract(off)
#pragma unroll
    for (int u = 0; u < kTileU; ++u) {
        const int cg = c.cnt - 64 * u;                 // valid lanes of this group (wavefront-uniform)
        if (cg <= 0) break;
        const uint32_t rl = (c.idx[u] & 0x7fffffffu) >> shift;
        const double p = c.val[u] * c.xv[u];
        const bool act = lane < cg;
        // layer boundaries: bit 31 marks the first entry of a layer >= 1 of the tile (inside a layer every row occurs once, in column
        // order since round 4); lane 0 always starts a pass
        unsigned long long B = __ballot(act && lane > 0 && (c.idx[u] >> 31) != 0);
        if (B == 0) {                                  // one layer: 64 different rows
            if (act) yl[rl] = yl[rl] + p;
        } else {
            int lo = 0;
            for (;;) {                                 // passes [lo, hi) between boundaries, in order
                const int hi = B ? __builtin_ctzll(B) : 64;
                if (act && lane >= lo && lane < hi) yl[rl] = yl[rl] + p;
                if (!B) break;
                B &= B - 1;
                lo = hi;
            }
        }
    }
}

template <int EPI, typename RP>
__global__ void __launch_bounds__(kBlock, kTileBlocksPerCu)
spmv_tile_kernel(SpmvArgs<RP> a, const RP *__restrict__ rowptr, const int32_t *__restrict__ srow, const uint32_t *__restrict__ toff,
                 const uint32_t *__restrict__ tidx, const double *__restrict__ tval, const double *__restrict__ xg, int S, int P,
                 int shift, unsigned *prog, int slack, const int32_t *__restrict__ vis, int v0, int nv, int pfd, int apoll, int64_t ncols, int dlim, const double *__restrict__ dummy) {
    // One PASS of an overlapped all-gather (AgPlan, sla_internal.hpp) walks the nv panels vis[v0 ..] instead of 0 .. P-1 and starts
    // from the running row sums a.yinit; vis == nullptr: all P panels ascending from zero (nv == P then).
    __shared__ double s_y[kTileWaves][kTileRows];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double coef;
    if (!spmv_prologue<EPI, RP>(a, s_red, coef)) return;
    double acc1 = 0.0, acc2 = 0.0;
    double *yl = s_y[wave < kTileWaves ? wave : 0];
    const uint32_t cmask = (1u << shift) - 1u;
    const int stride = (int)gridDim.x * kTileWaves;
    // Panel pacing.  The gathers only hit the L2 while the wavefronts of an XCD are on (nearly) the same panel, and left
    // alone they drift apart within a fraction of a slice (the SIMD arbiter favours the oldest wavefront; measured without
    // pacing: 78 % L2 misses, 33 GB through the fabric per launch).  So a wavefront does not START panel step q before every
    // workgroup of its XCD has finished step q - slack.  Progress is published WITHOUT atomics (device-scope atomics execute
    // on the memory side of the fabric: 512 wavefronts bumping one counter cost ~100 us per step, measured): each
    // workgroup keeps its wavefronts' step counts in LDS and plain-stores their minimum into its own slot of a per-XCD
    // table -- the store stays in the XCD's L2 -- and a waiting wavefront reads the <= 256 slots of its XCD with L1-bypassing
    // loads (up to four loads per poll, wavefront min).  The workgroups b with equal b % 8 share an XCD (probe_xcd_layout below)
    // and the grid is fully resident (kTileBlocksPerCu per CU).  Pacing is a throttle, never a correctness condition: a
    // wavefront that waits too long (grid not co-resident) stops pacing for the rest of the launch.
    __shared__ int s_prog[kBlock / 64];
    __shared__ int s_pf[kBlock / 64][64];   // where the x-panel prefetch lands (never read)
    __shared__ int s_poll[kBlock / 64][64 * kTilePollQ];   // the pacing slots of the XCD as the last look-ahead poll of each wavefront brought them
    if (tid < kBlock / 64) s_prog[tid] = tid < kTileWaves ? 0 : 0x7fffffff;
    for (int i = tid; i < (kBlock / 64) * 64 * kTilePollQ; i += kBlock) (&s_poll[0][0])[i] = 0;
    __syncthreads();
    const int xcd = (int)blockIdx.x & 7;
    const int nwg_xcd = ((int)gridDim.x - xcd + 7) >> 3;
    const int rounds = (S + stride - 1) / stride;
    if (vis == nullptr) nv = P;
    int *slots = prog ? (int *)prog + xcd * 256 : nullptr;      // <= 256 workgroups per XCD (kTileBlocksPerCu x 32 CUs)
    int *myslot = slots ? slots + ((int)blockIdx.x >> 3) : nullptr;
    bool pace = slack > 0 && slots != nullptr && nwg_xcd <= 256;
    int known = 0;                                               // steps every workgroup of the XCD is known to have finished
    auto publish = [&](int done) {                               // this wavefront has finished `done` steps
        if (!slots) return;
        if (lane == 0) {
            // Two wavefronts publishing at once may each miss the other's LDS update, and the staler minimum may reach the slot
            // last.  Stores of one wavefront to one address stay in order, so re-reading the minimum after the store and
            // storing again until it is stable leaves the slot at the true minimum.
            // (LDS accesses as relaxed workgroup-scope atomics, not volatile: the compiler puts s_waitcnt vmcnt(0) lgkmcnt(0) around every
            // volatile access -- round 4 found every publish, i.e. every tile, draining the wavefront's whole load pipeline there)
            // The store-then-read order the protocol needs is the LDS's own (one wavefront's LDS operations are served in issue order);
            // the empty asm statements keep the compiler from moving the reads over the stores.
            __hip_atomic_store(&s_prog[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int stored = -1;
            for (;;) {
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
    // The poll was a dependent L1-bypassing load (~1 us under this kernel's load) in front of every tile, and one the compiler had to put
    // s_waitcnt vmcnt(0) behind: every outstanding stream and gather of the wavefront drained there.  Round 4: the slots are polled one
    // step AHEAD and without a destination register -- a direct-to-LDS load (global_load_lds_dword, sc1 like the atomic load) issued at
    // step q drops the slots into s_poll, where step q + 1 reads them with a plain LDS read, a tile's worth of streams and gathers
    // later.  Progress only grows, so a stale (or not yet landed) value can only make a wavefront look again, with the loads below.
    auto poll = [&]() -> int {
        int v = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane + 64 * q < nwg_xcd) v = min(v, __hip_atomic_load(slots + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        return v;
    };
    int *pollw = s_poll[wave];
    const unsigned poll_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(&s_poll[wave][0]));
    if (apoll && nwg_xcd > 64 * kTilePollQ) apoll = 0;
    auto wait_for = [&](int need) {                              // until all workgroups of the XCD have finished `need` steps
        if (apoll) {
            if (!pace || need <= 0) return;
            int v = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < kTilePollQ; ++q)
                if (lane + 64 * q < nwg_xcd) v = min(v, __hip_atomic_load(pollw + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            int spins = 0;
            while (__ballot(v < need) != 0) {
                v = poll();
                if (__ballot(v < need) == 0) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > SLA_TILE_SPIN) { pace = false; return; }
            }
#pragma unroll
            for (int q = 0; q < kTilePollQ; ++q)
                if (64 * q < nwg_xcd) {
                    const int *src = slots + min(lane + 64 * q, nwg_xcd - 1);
                    const unsigned dst = poll_lds + 256u * q;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            return;
        }
        int spins = 0;
        while (pace && known < need) {
            int v = poll();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
            known = v;
            if (known >= need) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SLA_TILE_SPIN) pace = false;
        }
    };
    // x-panel prefetch (round 4).  The first touch of every line of a new panel was a demand miss of some gather -- 8192 lines per panel
    // and XCD, each holding up the chunk it belongs to for a trip to the memory side.  On entering visit step q a wavefront now pulls ITS
    // 1 / (wavefronts per XCD) of the panel of step q + pfd into the XCD's L2 with one direct-to-LDS load per 64 lines (no destination
    // register; the data lands in s_pf and is never read), one 128-byte line per lane.
    const int wix = ((int)blockIdx.x >> 3) * kTileWaves + wave, nwx = nwg_xcd * kTileWaves;
    const int plines = 1 << (shift - 4), lpw = (plines + nwx - 1) / nwx;
    const unsigned pf_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(&s_pf[wave][0]));   // (flat address of LDS: the low half is the LDS byte address)
    auto prefetch_panel = [&](int p) {
        for (int i = lane; i < lpw; i += 64) {
            const int64_t line = (int64_t)wix * lpw + i;
            const int64_t colx = ((int64_t)p << shift) + line * 16;
            if (line < plines && colx < ncols) {
                const double *src = xg + colx;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(pf_lds) : "memory");
            }
        }
    };
    int round = 0;
    for (int sv = (int)blockIdx.x * kTileWaves + wave; wave < kTileWaves; sv += stride, ++round) {   // (wavefronts beyond kTileWaves only join the reductions below)
        const int s = __builtin_amdgcn_readfirstlane(sv);   // everything per slice is wavefront-uniform: keep it in SGPRs
        if (round >= rounds || s >= S) {   // no (more) slices: count as finished with everything, the others must not wait for this one
            publish(0x7fffffff);
            break;
        }
        const int r0 = __builtin_amdgcn_readfirstlane(srow[s]), nr = __builtin_amdgcn_readfirstlane(srow[s + 1]) - r0;
        if (a.yinit) {
            for (int r = lane; r < nr; r += 64) yl[r] = a.yinit[r0 + r];
        } else {
            for (int r = lane; r < nr; r += 64) yl[r] = 0.0;
        }
        RP base = rowptr[r0];
        if constexpr (sizeof(RP) == 4) {
            base = (RP)__builtin_amdgcn_readfirstlane((int)base);
        } else {
            base = (RP)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)base >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)base));
        }
        const uint32_t *tp = toff + (size_t)s * (size_t)(P + 1);
        // walk the slice's tiles chunk by chunk (a chunk never crosses a tile); all of this is wavefront-uniform
        int j = -1, pj = 0, pjl = 0;             // visit step, its panel, the panels of the next 64 steps (one per lane)
        uint32_t k = 0, k1 = 0, plo = 0, phi = 0;
        auto advance = [&]() -> bool {
            while (k >= k1) {
                if (j >= 0) publish(round * nv + j + 1);   // done issuing the tile of step j
                ++j;
                if (j >= nv) return false;
                wait_for(round * nv + j - slack + 1);
                if (pfd > 0) {   // the panel of visit step j + pfd (of the wavefront's next slice past the end of this one)
                    int jj = j + pfd;
                    if (jj >= nv && round + 1 < rounds && s + stride < S) jj -= nv;
                    if (jj < nv) prefetch_panel(vis ? __builtin_amdgcn_readfirstlane(vis[v0 + jj]) : jj);
                }
                if ((j & 63) == 0) {   // the next 64 tile offsets, one per lane (no dependent load per tile)
                    pjl = vis ? vis[v0 + min(j + lane, nv - 1)] : min(j + lane, P - 1);
                    plo = tp[pjl];
                    phi = tp[pjl + 1];
                }
                pj = __builtin_amdgcn_readlane(pjl, j & 63);
                k = (uint32_t)__builtin_amdgcn_readlane((int)plo, j & 63);
                k1 = (uint32_t)__builtin_amdgcn_readlane((int)phi, j & 63);
            }
            return true;
        };
        auto issue = [&](TileChunk &c) {   // the chunk at (step j, k): its index / value streams (lanes past the end re-read the last entry)
            c.cnt = (int)min((uint32_t)(64 * kTileU), k1 - k);
            c.panel = pj;
            const uint32_t *ip = tidx + (base + (RP)k);
            const double *vp = tval + (base + (RP)k);
#pragma unroll
            for (int u = 0; u < kTileU; ++u) {
                const uint32_t i = (uint32_t)min(lane + 64 * u, c.cnt - 1);   // (cnt >= 1: advance() left k < k1)
                c.idx[u] = __builtin_nontemporal_load(ip + i);
                c.val[u] = __builtin_nontemporal_load(vp + i);
            }
            k += 64 * kTileU;
        };
        auto gather = [&](TileChunk &c) {
            const char *xb = (const char *)(c.panel < 0 ? dummy : xg + ((size_t)c.panel << shift));
#pragma unroll
            for (int u = 0; u < kTileU; ++u) c.xv[u] = *(const double *)(xb + (uint32_t)((c.idx[u] & cmask) << 3));
        };
        // three chunks in flight per wavefront: A is folded while B's gathers and C's index / value streams are outstanding
        TileChunk A, B, C;
#if SLA_TILE_UNIFORM
        // Round 4: the loop issues the SAME loads on every path.  With `if (more) issue(C)` the compiler's wait-count pass had to assume
        // the path on which C was not issued -- there B's streams are the youngest loads -- and put s_waitcnt vmcnt(0) in front of
        // gather(B): every chunk waited for the streams it had JUST issued (a full trip to HBM with one wavefront per SIMD and
        // nothing else to run).  Past the slice's last chunk the loop now issues empty chunks (cnt = 0: the loads re-read the first
        // entry of the arrays, the fold does nothing) until the pipeline has drained, so the waits are vmcnt(36): a chunk's streams
        // have one whole iteration -- the previous chunk's gathers and the chunk before's fold -- to arrive.
        bool live = true;
        auto next = [&](TileChunk &c) {
            if (live && !advance()) live = false;
            if (live) {
                issue(c);
            } else {
                c.cnt = 0;
                c.panel = -1;   // gathers from the matrix's all-zero dummy panel: whatever column bits the re-read entries carry, the address is inside
                                // an allocation (panel 0 may lie outside a window-mode slab's guard, a partial last panel ends early)
#pragma unroll
                for (int u = 0; u < kTileU; ++u) {   // (run-time indices: identical addresses would be merged into one load, and the count is the point)
                    const int i = min(lane + 64 * u, dlim);
                    c.idx[u] = __builtin_nontemporal_load(tidx + i);
                    c.val[u] = __builtin_nontemporal_load(tval + i);
                }
            }
        };
        next(A);
        next(B);
        gather(A);
        for (;;) {   // unrolled by three: the chunks rotate through the roles without register copies
            if (A.cnt == 0) break;
            next(C); gather(B); tile_fold_chunk(yl, A, shift, lane);
            if (B.cnt == 0) break;
            next(A); gather(C); tile_fold_chunk(yl, B, shift, lane);
            if (C.cnt == 0) break;
            next(B); gather(A); tile_fold_chunk(yl, C, shift, lane);
        }
#else
        bool hA = advance(), hB = false, hC = false;
        if (hA) issue(A);
        hB = hA && advance();
        if (hB) issue(B);
        if (hA) gather(A);
        for (;;) {   // unrolled by three: the chunks rotate through the roles without register copies
            if (!hA) break;
            hC = hB && advance(); if (hC) issue(C); if (hB) gather(B);
            tile_fold_chunk(yl, A, shift, lane);
            if (!hB) break;
            hA = hC && advance(); if (hA) issue(A); if (hC) gather(C);
            tile_fold_chunk(yl, B, shift, lane);
            if (!hC) break;
            hB = hA && advance(); if (hB) issue(B); if (hA) gather(A);
            tile_fold_chunk(yl, C, shift, lane);
        }
#endif
        for (int r = lane; r < nr; r += 64) spmv_epilogue<EPI, RP>(a, r0 + r, yl[r], coef, acc1, acc2);
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

// Panel pacing assumes the MI355X dispatch order: 8 XCDs, workgroups dealt out round-robin, so that the workgroups b with equal
// b % 8 share an XCD.  WHICH XCD gets workgroup 0 is not fixed: a dispatch usually starts at XCD 0 (tools/xcc_probe.cpp) but the
// first dispatch of a process was observed to start at XCD 7 (round 3: the probe below read 7 0 1 2 3 4 5 6 7 0 ... in the first
// context of every process and 0 1 2 ... afterwards) -- a rotation, which neither the pacing groups nor the XCD-contiguous walks
// (rb_walk) mind: they only need the GROUPS.  On another partition mode or part the slots a wavefront polls could belong to
// workgroups of another XCD, whose workgroup-scope stores may never become visible: every wavefront would spin to its limit on
// every launch.  So the layout is PROBED once per context (HW_REG_XCC_ID of the first 256 workgroups of a launch) and pacing is
// only used when workgroups b and b % 8 share an XCD for every b and the eight groups sit on eight different XCDs.
__global__ void xcc_probe_kernel(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID, bits [3:0]
}
int probe_xcd_layout(sla_ctx *c) {
    if (c->xcd8 >= 0) return SLA_OK;
    int *d = (int *)(c->d_result + 1024);   // 256 ints of the context's scratch
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(256), dim3(64), 0, stream_of(c), d);
    SLA_HIP_TRY(hipGetLastError());
    int h[256];
    SLA_HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream_of(c)));
    SLA_HIP_TRY(hipStreamSynchronize(stream_of(c)));
    int ok = 1;
    for (int b = 0; b < 256; ++b) ok &= h[b] == h[b & 7];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < i; ++j) ok &= h[i] != h[j];
    if (getenv("SLA_DEBUG_LOWER")) {
        fprintf(stderr, "[sla] xcd probe: ok=%d, first 24:", ok);
        for (int b = 0; b < 24; ++b) fprintf(stderr, " %d", h[b]);
        fprintf(stderr, "\n");
    }
    c->xcd8 = ok;
    return SLA_OK;
}

bool tiles_on(const sla_csr *A) { return A->use_tiles && A->ctx->tiles && A->ctx->spmv_algo == 0; }

int tiles_grid(const sla_csr *A) {
    if (A->tl_cu) return ctiles_grid(A);
    const int64_t blocks = ((int64_t)A->tl_S + kTileWaves - 1) / kTileWaves;
    return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)kTileBlocksPerCu * A->ctx->n_cu));
}

template <int EPI, typename RP>
static int launch_tiles_t(const sla_csr *A, const SpmvLaunch &l) {
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
    hipLaunchKernelGGL((spmv_tile_kernel<EPI, RP>), dim3(tiles_grid(A)), dim3(kBlock), 0, stream_of(c), a, a.rowptr, A->d_tlrow, A->d_tloff,
                       A->d_tlidx, A->d_tlval, l.x, A->tl_S, A->tl_P, A->tl_shift, A->d_tlprog, c->xcd8 == 1 ? c->tile_slack : 0, vis, l.tv0, nv,
                       c->xcd8 == 1 ? c->tile_prefetch : 0, c->tile_poll, A->n,
                       (int)std::min<int64_t>(64 * kTileU - 1, A->nnz - 1), A->d_tldummy);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

template <typename RP>
static int launch_tiles_rp(const sla_csr *A, const SpmvLaunch &l) {
    switch (l.epi) {
        case EPI_NONE: return launch_tiles_t<EPI_NONE, RP>(A, l);
        case EPI_DOT: return launch_tiles_t<EPI_DOT, RP>(A, l);
        case EPI_DOT2: return launch_tiles_t<EPI_DOT2, RP>(A, l);
        case EPI_DOT4: return launch_tiles_t<EPI_DOT4, RP>(A, l);
        case EPI_RES: return launch_tiles_t<EPI_RES, RP>(A, l);
        case EPI_AXPY_DOT: return launch_tiles_t<EPI_AXPY_DOT, RP>(A, l);
        case EPI_XPBY_NRM: return launch_tiles_t<EPI_XPBY_NRM, RP>(A, l);
        case EPI_SUB: return launch_tiles_t<EPI_SUB, RP>(A, l);
    }
    return fail(SLA_ERR_INVALID, "launch_spmv_tiles: unknown epilogue");
}

int launch_spmv_tiles(const sla_csr *A, const SpmvLaunch &l) {
    if (A->tl_cu) return launch_spmv_ctiles(A, l);   // CU-wide slices (round 5)
    return A->rp64 ? launch_tiles_rp<int64_t>(A, l) : launch_tiles_rp<int32_t>(A, l);
}

}  // namespace sla
