// sla_tri.hip -- triLowerSolve / triUpperSolve (Numeric/LinearAlgebra/Sparse.hs:750-811): one launch per dependency level of the
// level-scheduled triangle, one lane per row, ascending fold with separately rounded multiply / add, one IEEE division, then sparsifySV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

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

// ---------------------------------------------------------------------------------------------
// the same solve as ONE persistent launch (option tri_syncfree; round 5)
// ---------------------------------------------------------------------------------------------
// The level schedule costs a dependent launch per level (5.2 us x 646 levels at 216^3).  Here the workgroups of one resident grid walk
// the schedule slots in order (slot t in round t / (grid x 256): everything a slot depends on sits in an earlier slot, i.e. is held by a
// co-resident wavefront or is done), and a row waits for the rows it reads by POLLING x itself: x is pre-filled with a NaN payload no
// computation produces, a row's value is published with ONE 8-byte agent-scope store and read with agent-scope loads, so there is no
// flag beside the data and no fence.  The arithmetic is the level kernel's, entry by entry in ascending order: bit-identical results.
// A lane whose dependency does not show up within `spin_limit` polls raises *fail and leaves -- the host then runs the level schedule
// (a grid that is not co-resident, a part whose L2s do not forward agent-scope stores: never a hang).
constexpr unsigned long long kTriPending = 0x7ff8dead5a5a0001ull;

__global__ void __launch_bounds__(kBlock) tri_fill_pending_kernel(int64_t n, double *x) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        __hip_atomic_store((unsigned long long *)(x + i), kTriPending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(kBlock) tri_syncfree_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ tcol,
                                                               const double *__restrict__ tval, const double *__restrict__ tdiag,
                                                               const int32_t *__restrict__ order, int64_t nslots, const double *__restrict__ b,
                                                               double *x, int *fail, int spin_limit) {
#pragma clang fp contract(off)
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < nslots; base += (int64_t)gridDim.x * kBlock) {
        const int64_t t = base + threadIdx.x;
        bool done = t >= nslots;
        int64_t k = 0, e = 0;
        int i = 0;
        double d = 1.0, bi = 0.0, r = 0.0;
        if (!done) {
            k = tptr[t];
            e = tptr[t + 1];
            i = order[t];
            d = tdiag[t];
            bi = b[i];
        }
        int spins = 0;
        for (;;) {
            if (!done) {
                // up to four dependencies in flight, consumed in order as far as they have arrived
                unsigned long long v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = k + q < e ? __hip_atomic_load((const unsigned long long *)(x + tcol[k + q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTriPending;
                bool moved = false, chain = true;   // (in order: entry q of the batch is only taken if all before it were)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (chain && k < e && v[q] != kTriPending) {
                        const double prod = tval[k] * __longlong_as_double((long long)v[q]);
                        r = r + prod;
                        ++k;
                        moved = true;
                    } else {
                        chain = false;
                    }
                }
                if (k == e) {
                    const double xi = (bi - r) / d;
                    __hip_atomic_store((unsigned long long *)(x + i), (unsigned long long)__double_as_longlong(xi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                } else if (moved) {
                    spins = 0;
                } else if (++spins > spin_limit ||
                           ((spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {   // (a failure elsewhere: nobody waits for rows that will not come)
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                } else {
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (__ballot(!done) == 0) break;
        }
    }
}

int launch_tri_syncfree(const sla_csr *T, const sla_tri_plan *p, const double *b, double *x, int *d_fail) {
    sla_ctx *c = T->ctx;
    const int64_t n = T->m;
    hipStream_t st = stream_of(c);
    SLA_HIP_TRY(hipMemsetAsync(d_fail, 0, sizeof(int), st));
    hipLaunchKernelGGL(tri_fill_pending_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, x);
    // the grid must be co-resident: at most tri_grid workgroups, never more than 8 per CU
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(c->tri_grid ? c->tri_grid : 256, (int64_t)8 * c->n_cu), (n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(tri_syncfree_kernel, dim3(grid), dim3(kBlock), 0, st, p->d_tptr, p->d_tcol, p->d_tval, p->d_tdiag, p->d_order, n, b, x, d_fail,
                       c->tri_spin);
    SLA_HIP_TRY(hipGetLastError());
    return SLA_OK;
}

// ---------------------------------------------------------------------------------------------
// block-local persistent solve (option tri_syncfree = 2; round 5)
// ---------------------------------------------------------------------------------------------
// What the kernel above pays per dependency LEVEL is a trip through memory: a row's value is stored by one CU and polled by another
// (4.4 us x 646 levels at 216^3).  Here the sweep order is cut into blocks of <= 16384 consecutive rows and a workgroup owns a whole
// block with the block's x in LDS: a dependency inside the block (for a banded matrix: all but the farthest diagonals) costs an LDS
// round trip, only what a block reads from EARLIER blocks is polled in memory.  The plan (tri_blocks_build, sla_precond.cpp) orders a
// block's rows by their level inside the block, so the minimal unfinished slot of a workgroup never waits for a later slot; the blocks
// are taken in (block level, block) order by a co-resident grid (workgroup w: positions w, w + grid, ...), so the minimal unfinished
// block never waits for a block nobody holds.  A lane keeps a window of four entries: columns and values are loaded once, the x
// values polled until they arrive (LDS cell of the block, or x in memory for another block's row: both pre-filled with the pending
// payload), consumed strictly in ascending column order -- the reference's fold, bit for bit, for both triangles.
#ifndef SLA_TRI_THREADS
#define SLA_TRI_THREADS 512   // (216^3 in bricks: 1024: 0.77 ms, 512: 0.68, 256: 0.85; 1000^2 Poisson: 1.11 / 1.03 / 1.00)
#endif
constexpr int kTriBlockThreads = SLA_TRI_THREADS;
__device__ __forceinline__ bool t_last_slot(int tid, int64_t s0, int64_t s1) { return (s1 - 1 - s0) % kTriBlockThreads == tid; }

__global__ void __launch_bounds__(kTriBlockThreads) tri_blocks_kernel(const int64_t *__restrict__ slots, int64_t nb, const int32_t *__restrict__ row,
                                                                       const int64_t *__restrict__ tptr, const int32_t *__restrict__ tcol,
                                                                       const double *__restrict__ tval, const double *__restrict__ tdiag,
                                                                       const double *__restrict__ b, double *x, int ready, int *fail, int spin_limit, long long *trace) {
#pragma clang fp contract(off)
    extern __shared__ unsigned long long xl[];    // brows cells + one that is never pending (where unused window entries point)
    __shared__ int mark[kTriBlockThreads / 64];   // per wavefront: 2 bn + 1 once it has folded something of its bn-th batch, 2 bn + 2 once it is through with it
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid < kTriBlockThreads / 64) mark[tid] = 0;
    if (tid == 0) xl[ready] = 0;
    int bn = 0;                                   // batches this workgroup has started (over all its blocks)
    int seen = 0;                                 // the grid's progress count when this lane last looked
    for (int64_t p = blockIdx.x; p < nb; p += gridDim.x) {
        const int64_t s0 = slots[2 * p], s1 = slots[2 * p + 2];
        for (int64_t i = tid; i < s1 - s0; i += kTriBlockThreads) xl[i] = kTriPending;
        __syncthreads();
        if (trace && tid == 0) trace[4 * p] = wall_clock64();
        for (int64_t base = s0; base < s1; base += kTriBlockThreads, ++bn) {
            const int64_t t = base + tid;
            // the slots are in level order, so the chain through a batch runs wavefront after wavefront: one whose second predecessor (two
            // wavefronts before it, wrapping into the batch before) has not folded anything yet polls the LDS rarely --
            // sixteen wavefronts polling at full rate saturate the LDS and the one that has work waits behind their reads
            const int pred = (wave + kTriBlockThreads / 64 - 2) % (kTriBlockThreads / 64), pred_mark = wave < 2 ? 2 * bn - 1 : 2 * bn + 1;
            const int pred1 = (wave + kTriBlockThreads / 64 - 1) % (kTriBlockThreads / 64), pred1_mark = wave < 1 ? 2 * bn - 1 : 2 * bn + 1;
            bool marked = false;
            bool done = t >= s1;
            int64_t k = 0, e = 0;
            int i = 0;
            double d = 1.0, bi = 0.0, r = 0.0;
            if (!done) {
                k = tptr[t];
                e = tptr[t + 1];
                i = row[t];
                d = tdiag[t];
                bi = b[i];
            }
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) done = true;   // (a failure elsewhere: the host runs the level schedule)
            auto finish = [&]() {
                const unsigned long long xi = (unsigned long long)__double_as_longlong((bi - r) / d);
                __hip_atomic_store(&xl[t - s0], xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a row's cell is its slot inside the block)
                __hip_atomic_store((unsigned long long *)(x + i), xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            };
            if (trace && t == s0) trace[4 * p + 1] = wall_clock64();   // (the block's first row is about to be tried)
            if (!done && k == e) finish();
            // A lane holds a window of (up to) four consecutive entries of its row: columns and values are loaded once, the x values of
            // other blocks' rows polled in memory until each has arrived once (only while some lane of the wavefront still lacks one),
            // the block's own cells re-read every round; when the whole window is in, it is folded in ascending column order and the
            // next one loaded.  The round of a wavefront that is only WAITING must be a handful of instructions: sixteen wavefronts
            // share four SIMDs, and every instruction a waiting one issues is a slot the one that holds the chain does not get.
            int nw = 0, idle = 0, stall = 0;
            unsigned got = 0, glob = 0;          // per window entry: value in wx / lives in memory (another block's row)
            int cell[4] = {ready, ready, ready, ready};   // (the cell behind the block's: always there)
            const unsigned long long *gp[4] = {nullptr, nullptr, nullptr, nullptr};
            double wv[4] = {0.0, 0.0, 0.0, 0.0}, wx[4] = {0.0, 0.0, 0.0, 0.0};
            while (__ballot(!done) != 0) {
                if (!done && nw == 0) {   // the next window
                    nw = (int)min((int64_t)4, e - k);
                    got = 0;
                    glob = 0;
                    int32_t wc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // (all eight loads in flight: entries past the row's end re-read its last one)
                        const int64_t kk = k + min(q, nw - 1);
                        wc[q] = tcol[kk];
                        wv[q] = tval[kk];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        cell[q] = ready;
                        if (q < nw) {
                            if (wc[q] < 0) {
                                cell[q] = ~wc[q];
                            } else {
                                glob |= 1u << q;
                                gp[q] = (const unsigned long long *)(x + wc[q]);
                            }
                        }
                    }
                    // wait for the window HERE, not at its first use: the load counter also counts stores, so a wait in the fold would
                    // sit behind the write-through of the x value a neighbouring lane published a round ago -- a trip to memory per level
                    asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]));
                }
                const unsigned want = done ? 0u : (glob & ~got);
                const bool mem_wait = __ballot(want != 0) != 0;
                if (mem_wait) {   // (rare inside a block's chain: skipped by the whole wavefront)
                    unsigned long long gv[4] = {kTriPending, kTriPending, kTriPending, kTriPending};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((want >> q) & 1u) gv[q] = __hip_atomic_load(gp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gv[q] != kTriPending) {
                            wx[q] = __longlong_as_double((long long)gv[q]);
                            got |= 1u << q;
                        }
                }
                unsigned long long lv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) lv[q] = __hip_atomic_load(&xl[cell[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool all_in = !done && glob == got && lv[0] != kTriPending && lv[1] != kTriPending && lv[2] != kTriPending && lv[3] != kTriPending;
                if (__ballot(all_in) != 0) {
                    if (all_in) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double xq = ((glob >> q) & 1u) ? wx[q] : __longlong_as_double((long long)lv[q]);
                            const double prod = wv[q] * xq;
                            const double r2 = r + prod;
                            r = q < nw ? r2 : r;
                        }
                        k += nw;
                        nw = 0;
                        if (k == e) finish();
                    }
                    idle = 0;
                    stall = 0;
                    if (!marked) {
                        marked = true;
                        if ((tid & 63) == 0) __hip_atomic_store(&mark[wave], 2 * bn + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    // nothing came (wavefront-uniform from here).  A wavefront that waits for cells of its own block polls again at once when
                    // the chain is close (the wavefront two before it has begun to fold), rarely before that; one that waits for other
                    // blocks' rows sleeps longer the longer nothing has come -- every poll of it is a trip to memory.
                    ++idle;
                    ++stall;
                    if ((stall & 255) == 0) {   // somebody finished a block since the last look: the grid is alive, keep waiting
                        const int now = __hip_atomic_load(fail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (now != seen) { seen = now; stall = 0; }
                    }
                    if (stall > spin_limit || ((stall & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the host runs the level schedule instead)
                        done = true;
                    } else {
                        const bool near2 = __hip_atomic_load(&mark[pred], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= pred_mark;
                        const bool near1 = __hip_atomic_load(&mark[pred1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= pred1_mark;
                        if (near1 && !mem_wait) {
                        } else if (near2) {
                            __builtin_amdgcn_s_sleep(1);
                        } else if (mem_wait) {
                            if (idle > 64) __builtin_amdgcn_s_sleep(32);
                            else __builtin_amdgcn_s_sleep(8);
                        } else {
                            __builtin_amdgcn_s_sleep(16);
                        }
                    }
                }
            }
            if ((tid & 63) == 0) __hip_atomic_store(&mark[wave], 2 * bn + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (trace && t_last_slot(tid, s0, s1)) trace[4 * p + 2] = wall_clock64();
        __syncthreads();   // everybody is done reading the block's cells
        // the grid's progress count: what a waiting wavefront's patience is measured against (once per BLOCK: one counter for the whole
        // grid takes ~50 ns per update, per batch and wavefront that was 8 ms at 216^3)
        if (tid == 0) (void)__hip_atomic_fetch_add(fail + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (trace && tid == 0) { trace[4 * p + 3] = wall_clock64(); }
    }
}

int launch_tri_blocks(const sla_csr *T, const sla_tri_plan *p, int upper, const double *b, double *x, int *d_fail) {
    sla_ctx *c = T->ctx;
    const int64_t n = T->m;
    hipStream_t st = stream_of(c);
    // co-resident by construction: as many workgroups per CU as their LDS (the block's cells) leaves room for, never more than tri_grid
    const size_t lds = sizeof(unsigned long long) * ((size_t)p->brows + 1);
    // (checked BEFORE anything is enqueued: a device with less LDS per workgroup than a block's cells, or a runtime that refuses the attribute /
    // the occupancy query, sends the solve to the level schedule -- SLA_TRI_NO_FIT, no error)
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess || (size_t)lds_max < lds) { (void)hipGetLastError(); return SLA_TRI_NO_FIT; }
    if (hipFuncSetAttribute((const void *)tri_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned long long) * (kTriBlockRows + 1))) != hipSuccess) { (void)hipGetLastError(); return SLA_TRI_NO_FIT; }
    int per_cu = 0;   // (what the runtime says fits: registers, wavefront slots and this launch's LDS)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)tri_blocks_kernel, kTriBlockThreads, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return SLA_TRI_NO_FIT; }
    SLA_HIP_TRY(hipMemsetAsync(d_fail, 0, 2 * sizeof(int), st));   // [0] somebody gave up, [1] blocks finished (the grid's progress)
    hipLaunchKernelGGL(tri_fill_pending_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, x);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>({c->tri_grid ? (int64_t)c->tri_grid : (int64_t)1 << 30, (int64_t)per_cu * (int64_t)c->n_cu, p->nb}));
    // SLA_TRI_TRACE=<file>: per block taken (in the order taken) the 100 MHz clock at: cells initialised | first row tried | last row done | block left
    static const char *trace_path = getenv("SLA_TRI_TRACE");
    long long *d_trace = nullptr;
    if (trace_path) {
        SLA_HIP_TRY(hipMalloc((void **)&d_trace, sizeof(long long) * 4 * (size_t)p->nb));
        SLA_HIP_TRY(hipMemsetAsync(d_trace, 0, sizeof(long long) * 4 * (size_t)p->nb, st));
    }
    (void)upper;   // (the plan holds the triangle's side: the kernel does not care)
    hipLaunchKernelGGL(tri_blocks_kernel, dim3(grid), dim3(kTriBlockThreads), lds, st, p->d_bl_slots, p->nb, p->d_bl_row, p->d_bl_ptr, p->d_bl_col, p->d_bl_val,
                       p->d_bl_diag, b, x, (int)p->brows, d_fail, c->tri_spin, d_trace);
    SLA_HIP_TRY(hipGetLastError());
    if (d_trace) {
        std::vector<long long> t((size_t)(4 * p->nb));
        std::vector<int64_t> sl((size_t)(2 * p->nb + 1));
        SLA_HIP_TRY(hipStreamSynchronize(st));
        SLA_HIP_TRY(hipMemcpy(t.data(), d_trace, sizeof(long long) * t.size(), hipMemcpyDeviceToHost));
        SLA_HIP_TRY(hipMemcpy(sl.data(), p->d_bl_slots, sizeof(int64_t) * sl.size(), hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        if (FILE *f = fopen(trace_path, "w")) {
            long long t0 = t[0];
            for (int64_t q = 0; q < p->nb; ++q) t0 = std::min(t0, t[(size_t)(4 * q)]);
            fprintf(f, "# taken wg first_position rows start_us first_row_us last_row_us left_us\n");
            for (int64_t q = 0; q < p->nb; ++q)
                fprintf(f, "%lld %d %lld %lld %.2f %.2f %.2f %.2f\n", (long long)q, (int)(q % grid), (long long)sl[(size_t)(2 * q + 1)], (long long)(sl[(size_t)(2 * q + 2)] - sl[(size_t)(2 * q)]),
                        (t[(size_t)(4 * q)] - t0) * 0.01, (t[(size_t)(4 * q + 1)] - t0) * 0.01, (t[(size_t)(4 * q + 2)] - t0) * 0.01, (t[(size_t)(4 * q + 3)] - t0) * 0.01);
            fclose(f);
        }
    }
    return SLA_OK;
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

}  // namespace sla
