// sla_onchip.hip -- sla_solver_step(k) as ONE persistent launch with the solver state on chip (round 6).
//
// bicgstabStep (Numeric/LinearAlgebra/Sparse.hs:972-981) on matrices whose whole solver state fits the chip's register files and LDS:
// constant-coefficient stencil / banded matrices of <= 8 (offset, value) pairs and up to ~1.5 M rows (256 CUs x 12 x 512 rows).  The launch
// flow (K1 | K23 | K45, sla_solvers.cpp) is bound by launches in that regime -- fill, drain, three dependent kernels per step: 37.5 us per
// step at 1 M rows where the bytes would take 8 -- so here the step loop never leaves the chip:
//   * one 512-thread workgroup per CU owns a block of rows: consecutive rows (2-D stencils, bands) or a brick of a 3-D grid.  x, r0hat, r / s,
//     p, Ap, As of its rows live in REGISTERS (rows t, t + 512, ... of the block in thread t); p and s of own + halo cells in two LDS arrays
//     indexed by LOCAL cell -- own rows and the rows its entries reach into keep their relative positions, so (#>) reads its operands at
//     cell + local offset of the pair, with the row's pair mask deciding which entries exist: every row is the reference's ascending left
//     fold with separately rounded multiply and add (Data/Sparse/Common.hs:247-260), bit for bit the launch flow's (#>);
//   * ghost-row flow (the sharded solver's idea, one level down): r, p, s are kept valid on the halo cells as well, so a step needs TWO
//     grid-wide synchronisations, one behind each (#>): the first carries the partial sums of Ap . r0hat and the boundary rows of Ap, the second
//     the four sums of the fused K4+K5 flow (rho' through the identity, sla_vec_kernels.hip) and the boundary rows of As;
//   * what crosses workgroups is written with 8-byte agent-scope (write-through) stores, drained before the workgroup arrives at an
//     XCD-hierarchical counter barrier, and read with agent-scope loads behind it: no fence on the path, results independent of the
//     workgroup -> XCD placement.  Every workgroup sums the published partials in the same fixed order: identical scalars everywhere.
// The formulas are bicg_k2_kernel's and bicg_k45_kernel's term by term (same fused multiply-adds); only the grouping of the inner
// products differs from the launch flow (per wavefront, then per workgroup, then over workgroups), i.e. iterates agree to rounding.
// Probe and numbers: tools/onchip_probe.cpp, profiles/r06_onchip_probe.txt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>

#include "sla_internal.hpp"
#include "sla_device.hpp"

namespace sla {

namespace {

constexpr int T = kOcThreads;
constexpr int NW = T / 64;

struct OcArgs {
    const uint32_t *own_cm;
    const int32_t *own_row;
    const uint32_t *halo_cell;
    const int32_t *halo_row;
    const int32_t *halo_src;   // [G][HPT * 512]: slot (workgroup * RPT * 512 + position) of the row the cell mirrors: where its owner publishes it
    int L, np;
    int loff[kOcMaxPairs];
    double val[kOcMaxPairs];
    double *x, *r, *p;
    double *u;                 // CGS only
    const double *rhat;
    const double *b;           // RES instantiations (linSolve0): the right-hand side
    double *pubA, *pubS, *parts;
    unsigned *bar;
    SolverScalars *sc;
    int par, k;
    int sync;                  // 0: counter barrier (XCD-hierarchical), 1: epoch words polled by everybody
    int fault;                 // test hook (option onchip_fault): the last workgroup leaves at once -- the others' barrier times out (a lost CU, rehearsed)
};

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid-wide synchronisation that also carries K sums.  Every wavefront has left its K partial sums in red[8 k + wavefront] (LDS) and has
// issued its write-through stores (boundary rows).  All wavefronts drain their stores; wavefront 0 folds the per-wavefront sums in a
// fixed tree, lane 0 publishes them in parts[k * G + b], drains that store too and arrives: workgroups with equal b % 8 (observed: one
// XCD) share an arrival counter, the last arriver of a group bumps the top counter, the last group releases everybody through per-group
// generation words.  Returns false when the other workgroups did not show up (a co-residency failure: sc->flags gets SLA_FLAG_SYNC_TIMEOUT).
// flags (round 6, second half; option onchip_sync = 1; NOT the default -- measured slower, see below): ONE hop instead of three.  Every workgroup owns an epoch word; it publishes its
// partial sums, drains them, stores the epoch into its word -- and the first wavefront of every workgroup polls all G words (one 16-byte
// load per lane) until none is behind.  No atomic read-modify-write stands on the path (the counters cost two dependent atomics and a
// release store); the sums are read afterwards exactly as before, so the bits do not change.  MEASURED (config 2, same box): BiCGSTAB 17.4 us per step against
// 14.6 with the counters, CGS 15.1 against 12.05, linSolve0 22.3 against 18.8 -- every round of polling is 256 x 256 agent-scope loads on the memory side,
// which costs more than the two dependent atomics it removes.  The counters stay.
// `parts` is the buffer of THIS epoch's parity in both modes: a workgroup that is already past the barrier publishes its next partial sums
// into the other half, never into the one a slower workgroup may still be reading.
template <int K>
__device__ __forceinline__ bool oc_grid_sync(unsigned *bar, unsigned epoch, SolverScalars *sc, const double *red, double *parts, int *s_ok, int flags_mode) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int G = gridDim.x, l = threadIdx.x & 7;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = red[NW * k + l];
#pragma unroll
            for (int off = NW / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (threadIdx.x == 0) st_agent(parts + (size_t)k * G + blockIdx.x, s);
        }
        if (flags_mode) {
            unsigned *words = bar + 32 * 17;   // G <= 256 epoch words behind the counters' block
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(words + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int lane = threadIdx.x;
            int ok = 1;
            const long long t0 = wall_clock64();
            for (;;) {
                bool behind = false;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = lane + 64 * m;
                    if (j < G) behind |= __hip_atomic_load(words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch;
                }
                if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) {
                    if (lane == 0) __hip_atomic_fetch_or(&sc->flags, (int)SLA_FLAG_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            if (lane == 0) *s_ok = ok;
        } else if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned Gu = gridDim.x, g = blockIdx.x & 7, ng = Gu < 8 ? Gu : 8;
            const unsigned members = (Gu - g + 7) / 8;
            unsigned *cnt = bar + 32 * g, *top = bar + 32 * 8, *gen = bar + 32 * (9 + g);
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * members) {
                const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t + 1 == epoch * ng)
                    for (unsigned j = 0; j < ng; ++j) __hip_atomic_store(bar + 32 * (9 + j), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int ok = 1;
            const long long t0 = wall_clock64();   // (100 MHz: the watchdog is 2 s without the generation word moving)
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) {
                    __hip_atomic_fetch_or(&sc->flags, (int)SLA_FLAG_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            *s_ok = ok;
        }
    }
    __syncthreads();
    return *s_ok != 0;
}
// the sum of the G (<= 256) published partials of one quantity, formed by EVERY wavefront for itself in the same fixed order (lane j adds
// parts j, j + 64, j + 128, j + 192, then a butterfly): no LDS, no workgroup barrier, identical bits everywhere
__device__ __forceinline__ double oc_wave_total(const double *parts, int G) {
    const int l = threadIdx.x & 63;
    double v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = l + 64 * m < G ? ld_agent(parts + l + 64 * m) : 0.0;
    double s = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// RPT / HPT: own rows / halo cells per thread (instantiation classes; unused slots are harmless, see below).  NP: the matrix's pairs
// (0: any number <= 8 -- every row through the mask tests).
// RES (round 6, linSolve0 on chip): after every step the TRUE residual norm2 ((aa #> x) ^-^ b) is evaluated the way runIter does
// (Sparse.hs:1041-1052) -- x of own + halo cells in a third LDS array, b of the own rows in registers, the sum of squares riding on the NEXT pass's
// first synchronisation -- and every workgroup takes the same decision: stop at the first iterate with resnorm <= tol, or after a.k steps
// (silently, like the reference's nits).  Workgroup 0 keeps resnorm, the residual trace and the flags the way check_kernel / residual_converged do.
template <int RPT, int HPT, int NP, bool RES>
__global__ void __launch_bounds__(T) oc_bicgstab_kernel(OcArgs a) {
    extern __shared__ __attribute__((aligned(16))) double oc_lds[];
    __shared__ int s_ok;
    const int LA = (a.L + 2) & ~1;   // cells 0 .. L-1 + one dummy cell (L): what the slots of a short block write to
    double *P = oc_lds, *S = oc_lds + LA, *X = S + LA, *red = RES ? X + LA : S + LA;
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x, wave = t >> 6;
    SolverScalars *sc = a.sc;
    if (sc->done) return;   // (written by an earlier launch: every workgroup takes the same exit)
    if (a.fault && b == G - 1) return;
    // Branch-free slots: a slot without a row (short blocks) has pair mask 0, boundary 0, an in-range cell to read around and zeros
    // for its state, and writes to the dummy cell; a halo slot without a cell reads row 0 and writes the dummy cell.
    uint32_t cm[RPT];
    double x[RPT], rh[RPT], r[RPT], p[RPT], ap[RPT], as[RPT];
    const int np = NP ? NP : a.np;
    int fullbits = 0;   // bit i: every lane of this wavefront holds all np entries in its row i (interior rows: no mask tests)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const size_t j = ((size_t)b * RPT + i) * T + t;
        cm[i] = a.own_cm[j];
        if (NP && __builtin_amdgcn_ballot_w64(((cm[i] >> 16) & 0xff) != (1u << np) - 1) == 0) fullbits |= 1 << i;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const bool v = (cm[i] >> 25) & 1;
        const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
        const double x0 = a.x[g], r0 = a.r[g], p0 = a.p[g], h0 = a.rhat[g];
        x[i] = v ? x0 : 0.0;
        r[i] = v ? r0 : 0.0;
        p[i] = v ? p0 : 0.0;
        rh[i] = v ? h0 : 0.0;
        ap[i] = as[i] = 0.0;
    }
    double bb[RES ? RPT : 1];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
            const double b0 = a.b[g];
            bb[i] = ((cm[i] >> 25) & 1) ? b0 : 0.0;
        }
    }
    auto wcell = [&](uint32_t c) -> int { return ((c >> 25) & 1) ? (int)(c & 0xffff) : a.L; };
    uint32_t hc[HPT];
    int32_t hg[HPT];
    double hap[HPT];   // halo(Ap) of the step under way (a thread's halo slots are its own: registers, not LDS)
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const size_t j = ((size_t)b * HPT + i) * T + t;
        hc[i] = a.halo_cell[j];
        hg[i] = a.halo_src[j];
        const int32_t g = a.halo_row[j];
        P[hc[i]] = a.p[g];   // halo(p) and halo(r): the invariant every step starts from
        S[hc[i]] = a.r[g];
        if constexpr (RES) X[hc[i]] = a.x[g];
        hap[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) P[wcell(cm[i])] = p[i];
    __syncthreads();
    double rho = sc->rho2[a.par], alpha = 0.0, omega = 0.0, beta = 0.0;
    const double tol = RES ? sc->tol : 0.0;
    const int it0 = sc->iters;
    int steps_done = 0;
    bool conv = false;
    unsigned epoch = 0;
    // one row's left fold: a * x then +, two roundings like the reference's, never an FMA; ascending pair (= column) order
    auto fold = [&](const double *V, uint32_t c, bool full) -> double {
#pragma clang fp contract(off)
        const int cell = c & 0xffff;
        double y = 0.0;
        if (NP && full) {   // (wave-uniform)
#pragma unroll
            for (int k = 0; k < (NP ? NP : 1); ++k) {
                const double pk = a.val[k] * V[cell + a.loff[k]];
                y = y + pk;
            }
        } else {
            const uint32_t m = c >> 16;
#pragma unroll
            for (int k = 0; k < kOcMaxPairs; ++k) {
                if (k < np) {
                    const double pk = a.val[k] * V[cell + a.loff[k]];
                    y = ((m >> k) & 1) ? y + pk : y;
                }
            }
        }
        return y;
    };
    // RES: pass `step` first evaluates the true residual of the iterate after `step` steps -- in the K1 phase, its sum of squares riding on the first
    // synchronisation (the launch flow's dual-SpMV idea: two synchronisations per iteration, not three) --, tests it, and only then applies step
    // `step + 1`; one more pass than steps, the last one for the residual of the final iterate alone.
    for (int step = 0; step < a.k + (RES ? 1 : 0); ++step) {
        // (opaque per step: nothing derived from the slot words -- LDS addresses, publish addresses, mask tests -- is hoisted out of the loop
        // and kept in registers next to the state: 130 live registers otherwise)
#pragma unroll
        for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(cm[i]));
        int sl = b * RPT * T + t;   // this thread's first slot: a row is published in its own slot (no row numbers in the loop)
        asm volatile("" : "+v"(sl));
#pragma unroll
        for (int i = 0; i < HPT; ++i) asm volatile("" : "+v"(hc[i]), "+v"(hg[i]));
        // ---- K1: aap = aa #> p ; aap <.> r0hat.  Boundary rows of aap go out first (the plan puts them in the first slots: their
        //      write-through stores are in flight while the other rows are folded) ----
        {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                ap[i] = fold(P, cm[i], (fullbits >> i) & 1);
                acc += ap[i] * rh[i];
                if ((cm[i] >> 24) & 1) st_agent(a.pubA + (sl + i * T), ap[i]);
                __builtin_amdgcn_sched_barrier(0);   // one row's LDS operands at a time: the wavefronts hide the latency, the registers hold the state
            }
            acc = wave_sum(acc);
            if ((t & 63) == 0) red[wave] = acc;
            if constexpr (RES) {   // trueResidualNorm of the current x = norm2 ((aa #> x) ^-^ b)   (Sparse.hs:1041)
                double acr = 0.0;
                if (step > 0) {
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const double d = fold(X, cm[i], (fullbits >> i) & 1) - bb[i];
                        acr += d * d;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                acr = wave_sum(acr);
                if ((t & 63) == 0) red[NW + wave] = acr;
            }
        }
        ++epoch;
        if (!oc_grid_sync<RES ? 2 : 1>(a.bar, epoch, sc, red, a.parts + (size_t)(epoch & 1) * 4 * G, &s_ok, a.sync)) return;
        if constexpr (RES) {
            if (step > 0) {   // runIter's test (:1047-1050) on the iterate after `step` steps
                const double rn = sqrt(oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G + G, G));
                conv = rn <= tol;
                if (b == 0 && t == 0) {
                    const int it = it0 + step;
                    sc->resnorm = rn;
                    if (sc->hist && it >= 1 && it <= sc->hist_cap) sc->hist[it - 1] = rn;
                    if (conv) { sc->done = 1; sc->flags |= SLA_FLAG_CONVERGED; }
                    if (!is_finite(rn)) sc->flags |= SLA_FLAG_NONFINITE;
                }
            }
            if (conv || step == a.k) break;   // (the same decision in every workgroup: the total is formed from the same partials in the same order)
        }
        // ---- K2 on own + halo cells: alphaj = (r <.> r0hat) / (aap <.> r0hat) ; sj = r ^-^ (alphaj .* aap) ----
        {
#pragma unroll
            for (int i = 0; i < HPT; ++i) hap[i] = ld_agent(a.pubA + hg[i]);
            alpha = rho / oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G, G);
#pragma unroll
            for (int i = 0; i < HPT; ++i) S[hc[i]] = __builtin_fma(-alpha, hap[i], S[hc[i]]);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                r[i] = __builtin_fma(-alpha, ap[i], r[i]);   // s
                S[wcell(cm[i])] = r[i];
            }
        }
        __syncthreads();
        // ---- K3: aasj = aa #> sj ; aasj <.> sj, aasj <.> aasj, aasj <.> r0hat, sj <.> r0hat ----
        {
            double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                as[i] = fold(S, cm[i], (fullbits >> i) & 1);
                q[0] += as[i] * r[i];
                q[1] += as[i] * as[i];
                q[2] += as[i] * rh[i];
                q[3] += r[i] * rh[i];
                if ((cm[i] >> 24) & 1) st_agent(a.pubS + (sl + i * T), as[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = wave_sum(q[k]);
                if ((t & 63) == 0) red[NW * k + wave] = q[k];
            }
        }
        ++epoch;
        if (!oc_grid_sync<4>(a.bar, epoch, sc, red, a.parts + (size_t)(epoch & 1) * 4 * G, &s_ok, a.sync)) return;
        // ---- K4 + K5 on own + halo cells: omegaj ; xj1 = x ^+^ alphaj .* p ^+^ omegaj .* sj ; rj1 = sj ^-^ omegaj .* aasj ;
        //      betaj = rho' / rho * alphaj / omegaj with rho' = sj . r0hat - omegaj (aasj . r0hat) ; pj1 = rj1 ^+^ betaj .* (p ^-^ omegaj .* aap) ----
        {
            double has[HPT];
#pragma unroll
            for (int i = 0; i < HPT; ++i) has[i] = ld_agent(a.pubS + hg[i]);
            double q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G + (size_t)k * G, G);
            omega = q[0] / q[1];
            const double rn = q[3] - omega * q[2];
            beta = rn / rho * alpha / omega;
            rho = rn;
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                if constexpr (RES) X[hc[i]] = __builtin_fma(omega, S[hc[i]], __builtin_fma(alpha, P[hc[i]], X[hc[i]]));   // halo(x): the same update, from the old p and s
                const double rv = __builtin_fma(-omega, has[i], S[hc[i]]);
                S[hc[i]] = rv;   // halo(r) of the next step
                P[hc[i]] = __builtin_fma(beta, __builtin_fma(-omega, hap[i], P[hc[i]]), rv);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                x[i] = __builtin_fma(omega, r[i], __builtin_fma(alpha, p[i], x[i]));
                r[i] = __builtin_fma(-omega, as[i], r[i]);
                p[i] = __builtin_fma(beta, __builtin_fma(-omega, ap[i], p[i]), r[i]);
                P[wcell(cm[i])] = p[i];
                if constexpr (RES) X[wcell(cm[i])] = x[i];
            }
        }
        __syncthreads();
        steps_done = step + 1;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if ((cm[i] >> 25) & 1) {
            const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
            a.x[g] = x[i];
            a.r[g] = r[i];
            a.p[g] = p[i];
        }
    if (b == 0 && t == 0) {
        sc->rho2[(a.par + steps_done) & 1] = rho;   // where the launch flow's next step (parity par + steps) reads it
        sc->alpha = alpha;
        sc->omega = omega;
        sc->beta = beta;
        sc->iters = it0 + steps_done;
    }
}

// cgsStep (Numeric/LinearAlgebra/Sparse.hs:928-939) on chip, the same plan and the same two synchronisations per step:
//   C1  aap = aa #> p ; aap <.> rhat                                   | sync 1: the sum + the boundary rows of aap
//   C2  alpha ; q = u ^-^ alpha .* aap ; u ^+^ q ; x1 (own + halo cells: u is kept valid on the halo in registers, p and u + q in LDS)
//   C3  aa #> (u ^+^ q) ; r1 = r ^-^ alpha .* (...) ; r1 <.> rhat       | sync 2: the sum + the boundary rows of r1
//   C4  beta ; u1 = r1 ^+^ beta .* q ; p1 = u1 ^+^ beta .* (q ^+^ beta .* p)   (own + halo cells: r1 of a halo cell comes from its owner)
// Expressions are cgs_c24_kernel's and the SpMV epilogue's (EPI_AXPY_DOT) term by term; the grouping of the two inner products differs from
// the launch flow's, so the iterates agree with it to rounding.
// RES (linSolve0 CGS_ on chip): x1 is complete right after alpha, so its true residual is folded in the C3 phase and its sum of squares rides on
// the step's SECOND synchronisation -- no third one; the test (Sparse.hs:1047-1050) is taken by every workgroup alike after C4.
template <int RPT, int HPT, int NP, bool RES>
__global__ void __launch_bounds__(T) oc_cgs_kernel(OcArgs a) {
    extern __shared__ __attribute__((aligned(16))) double oc_lds[];
    __shared__ int s_ok;
    const int LA = (a.L + 2) & ~1;
    double *P = oc_lds, *S = oc_lds + LA, *X = S + LA, *red = RES ? X + LA : S + LA;
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x, wave = t >> 6;
    SolverScalars *sc = a.sc;
    if (sc->done) return;
    if (a.fault && b == G - 1) return;
    uint32_t cm[RPT];
    double x[RPT], rh[RPT], r[RPT], p[RPT], u[RPT], q[RPT];   // (q: aap until alpha is known, then q)
    const int np = NP ? NP : a.np;
    int fullbits = 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const size_t j = ((size_t)b * RPT + i) * T + t;
        cm[i] = a.own_cm[j];
        if (NP && __builtin_amdgcn_ballot_w64(((cm[i] >> 16) & 0xff) != (1u << np) - 1) == 0) fullbits |= 1 << i;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const bool v = (cm[i] >> 25) & 1;
        const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
        const double x0 = a.x[g], r0 = a.r[g], p0 = a.p[g], u0 = a.u[g], h0 = a.rhat[g];
        x[i] = v ? x0 : 0.0;
        r[i] = v ? r0 : 0.0;
        p[i] = v ? p0 : 0.0;
        u[i] = v ? u0 : 0.0;
        rh[i] = v ? h0 : 0.0;
        q[i] = 0.0;
    }
    double bb[RES ? RPT : 1];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
            const double b0 = a.b[g];
            bb[i] = ((cm[i] >> 25) & 1) ? b0 : 0.0;
        }
    }
    auto wcell = [&](uint32_t c) -> int { return ((c >> 25) & 1) ? (int)(c & 0xffff) : a.L; };
    uint32_t hc[HPT];
    int32_t hg[HPT];
    double hu[HPT], hq[HPT];   // u and q of this thread's halo cells (p and u + q of the halo live in LDS with the own cells; r1 of a halo cell is read
                               // from its owner, who publishes r1 -- not aa #> (u + q) -- for its boundary rows: one register array less)
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const size_t j = ((size_t)b * HPT + i) * T + t;
        hc[i] = a.halo_cell[j];
        hg[i] = a.halo_src[j];
        const int32_t g = a.halo_row[j];
        P[hc[i]] = a.p[g];
        if constexpr (RES) X[hc[i]] = a.x[g];
        hu[i] = a.u[g];
        hq[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) P[wcell(cm[i])] = p[i];
    __syncthreads();
    double rho = sc->rho2[a.par], alpha = 0.0, beta = 0.0;
    const double tol = RES ? sc->tol : 0.0;
    const int it0 = sc->iters;
    int steps_done = 0;
    bool conv = false;
    unsigned epoch = 0;
    auto fold = [&](const double *V, uint32_t c, bool full) -> double {
#pragma clang fp contract(off)
        const int cell = c & 0xffff;
        double y = 0.0;
        if (NP && full) {
#pragma unroll
            for (int k = 0; k < (NP ? NP : 1); ++k) {
                const double pk = a.val[k] * V[cell + a.loff[k]];
                y = y + pk;
            }
        } else {
            const uint32_t m = c >> 16;
#pragma unroll
            for (int k = 0; k < kOcMaxPairs; ++k) {
                if (k < np) {
                    const double pk = a.val[k] * V[cell + a.loff[k]];
                    y = ((m >> k) & 1) ? y + pk : y;
                }
            }
        }
        return y;
    };
    for (int step = 0; step < a.k && !conv; ++step) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(cm[i]));
        int sl = b * RPT * T + t;
        asm volatile("" : "+v"(sl));
#pragma unroll
        for (int i = 0; i < HPT; ++i) asm volatile("" : "+v"(hc[i]), "+v"(hg[i]));
        // ---- C1: aap = aa #> p ; aap <.> rhat ----
        {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                q[i] = fold(P, cm[i], (fullbits >> i) & 1);
                acc += q[i] * rh[i];
                if ((cm[i] >> 24) & 1) st_agent(a.pubA + (sl + i * T), q[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = wave_sum(acc);
            if ((t & 63) == 0) red[wave] = acc;
        }
        ++epoch;
        if (!oc_grid_sync<1>(a.bar, epoch, sc, red, a.parts + (size_t)(epoch & 1) * 4 * G, &s_ok, a.sync)) return;
        // ---- C2 on own + halo cells: alphaj ; q = u ^-^ alphaj .* aap ; uq = u ^+^ q ; xj1 = x ^+^ alphaj .* uq ----
        {
            double hap[HPT];
#pragma unroll
            for (int i = 0; i < HPT; ++i) hap[i] = ld_agent(a.pubA + hg[i]);
            alpha = rho / oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G, G);
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                hq[i] = __builtin_fma(-alpha, hap[i], hu[i]);
                const double sv = hu[i] + hq[i];
                S[hc[i]] = sv;
                if constexpr (RES) X[hc[i]] = __builtin_fma(alpha, sv, X[hc[i]]);   // halo(x): the same update
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                q[i] = __builtin_fma(-alpha, q[i], u[i]);
                const double sv = u[i] + q[i];
                x[i] = __builtin_fma(alpha, sv, x[i]);
                S[wcell(cm[i])] = sv;
                if constexpr (RES) X[wcell(cm[i])] = x[i];
            }
        }
        __syncthreads();
        // ---- C3: aa #> (u ^+^ q) ; rj1 = r ^-^ alphaj .* (...) ; rj1 <.> rhat ----
        {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const double auq = fold(S, cm[i], (fullbits >> i) & 1);
                r[i] = __builtin_fma(-alpha, auq, r[i]);
                if ((cm[i] >> 24) & 1) st_agent(a.pubS + (sl + i * T), r[i]);
                acc += r[i] * rh[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = wave_sum(acc);
            if ((t & 63) == 0) red[wave] = acc;
            if constexpr (RES) {   // trueResidualNorm of xj1 = norm2 ((aa #> x) ^-^ b)   (Sparse.hs:1041)
                double acr = 0.0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const double d = fold(X, cm[i], (fullbits >> i) & 1) - bb[i];
                    acr += d * d;
                    __builtin_amdgcn_sched_barrier(0);
                }
                acr = wave_sum(acr);
                if ((t & 63) == 0) red[NW + wave] = acr;
            }
        }
        ++epoch;
        if (!oc_grid_sync<RES ? 2 : 1>(a.bar, epoch, sc, red, a.parts + (size_t)(epoch & 1) * 4 * G, &s_ok, a.sync)) return;
        if constexpr (RES) {   // runIter's test on the iterate after step + 1 steps (the state record is completed below either way)
            const double rn = sqrt(oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G + G, G));
            conv = rn <= tol;
            if (b == 0 && t == 0) {
                const int it = it0 + step + 1;
                sc->resnorm = rn;
                if (sc->hist && it >= 1 && it <= sc->hist_cap) sc->hist[it - 1] = rn;
                if (conv) { sc->done = 1; sc->flags |= SLA_FLAG_CONVERGED; }
                if (!is_finite(rn)) sc->flags |= SLA_FLAG_NONFINITE;
            }
        }
        // ---- C4 on own + halo cells: betaj ; uj1 = rj1 ^+^ betaj .* q ; pj1 = uj1 ^+^ betaj .* (q ^+^ betaj .* p) ----
        {
            double hr[HPT];
#pragma unroll
            for (int i = 0; i < HPT; ++i) hr[i] = ld_agent(a.pubS + hg[i]);   // r1 of the halo cells, from their owners
            const double rn = oc_wave_total(a.parts + (size_t)(epoch & 1) * 4 * G, G);
            beta = rn / rho;
            rho = rn;
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                const double un = __builtin_fma(beta, hq[i], hr[i]);
                P[hc[i]] = __builtin_fma(beta, __builtin_fma(beta, P[hc[i]], hq[i]), un);
                hu[i] = un;
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const double un = __builtin_fma(beta, q[i], r[i]);
                p[i] = __builtin_fma(beta, __builtin_fma(beta, p[i], q[i]), un);
                u[i] = un;
                P[wcell(cm[i])] = p[i];
            }
        }
        __syncthreads();
        steps_done = step + 1;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if ((cm[i] >> 25) & 1) {
            const int32_t g = a.own_row[((size_t)b * RPT + i) * T + t];
            a.x[g] = x[i];
            a.r[g] = r[i];
            a.p[g] = p[i];
            a.u[g] = u[i];
        }
    if (b == 0 && t == 0) {
        sc->rho2[(a.par + steps_done) & 1] = rho;
        sc->alpha = alpha;
        sc->beta = beta;
        sc->iters = it0 + steps_done;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// the plan: rows -> workgroups -> local cells
// ------------------------------------------------------------------------------------------------------------------------------
struct Geo {                         // how rows are dealt: mode 0 consecutive rows, mode 1 bricks of an nx x ny x nz grid
    int mode = 0;
    int64_t R = 0;                   // mode 0: rows per workgroup
    int64_t nx = 0, ny = 0, nz = 0;  // mode 1
    int bx = 0, by = 0, bz = 0;
    int G = 0, L = 0;
    int loff[kOcMaxPairs] = {};
};

constexpr size_t kOcLdsMax = 160 * 1024 - 512;   // dynamic LDS a workgroup may ask for (the static part: a few words)
static size_t oc_lds_bytes(int L, bool res = false) { return sizeof(double) * ((size_t)(res ? 3 : 2) * ((L + 2) & ~1) + 4 * NW + 8); }
static int rpt_class(int64_t own) { return own <= 4 * T ? 4 : own <= 8 * T ? 8 : own <= 12 * T ? 12 : 0; }
static int hpt_class(int64_t halo) { return halo <= 4 * T ? 4 : halo <= 8 * T ? 8 : 0; }
static bool class_ok(int rpt, int hpt) { return rpt != 0 && hpt != 0 && !(rpt == 12 && hpt == 8); }   // (12 x 8 slots do not fit 256 registers: ~50 spilled)

template <int RPT, int HPT, bool RES = false>
static const void *oc_kernel_np(int np) {
    if (np == 5) return (const void *)oc_bicgstab_kernel<RPT, HPT, 5, RES>;
    if (np == 7) return (const void *)oc_bicgstab_kernel<RPT, HPT, 7, RES>;
    return (const void *)oc_bicgstab_kernel<RPT, HPT, 0, RES>;
}
// linSolve0 on chip: b of the own rows in registers on top of the step's state -- the classes that still fit 256 registers
static const void *oc_kernel_res(int rpt, int hpt, int np) {
    switch (rpt * 16 + hpt) {
        case 4 * 16 + 4: return oc_kernel_np<4, 4, true>(np);
        case 4 * 16 + 8: return oc_kernel_np<4, 8, true>(np);
        case 8 * 16 + 4: return oc_kernel_np<8, 4, true>(np);
        case 8 * 16 + 8: return oc_kernel_np<8, 8, true>(np);
    }
    return nullptr;
}
template <int RPT, int HPT, bool RES = false>
static const void *oc_cgs_kernel_np(int np) {
    if (np == 5) return (const void *)oc_cgs_kernel<RPT, HPT, 5, RES>;
    if (np == 7) return (const void *)oc_cgs_kernel<RPT, HPT, 7, RES>;
    return (const void *)oc_cgs_kernel<RPT, HPT, 0, RES>;
}
static const void *oc_kernel_cgs_res(int rpt, int hpt, int np) {   // (+ b of the own rows: 8 x 8 no longer fits)
    switch (rpt * 16 + hpt) {
        case 4 * 16 + 4: return oc_cgs_kernel_np<4, 4, true>(np);
        case 4 * 16 + 8: return oc_cgs_kernel_np<4, 8, true>(np);
        case 8 * 16 + 4: return oc_cgs_kernel_np<8, 4, true>(np);
    }
    return nullptr;
}
// (CGS keeps u, r and q of the halo cells in registers where BiCGSTAB keeps one array: the 12 x 4 class does not fit 256 registers and is declined)
static const void *oc_kernel_cgs(int rpt, int hpt, int np) {
    switch (rpt * 16 + hpt) {
        case 4 * 16 + 4: return oc_cgs_kernel_np<4, 4>(np);
        case 4 * 16 + 8: return oc_cgs_kernel_np<4, 8>(np);
        case 8 * 16 + 4: return oc_cgs_kernel_np<8, 4>(np);
        case 8 * 16 + 8: return oc_cgs_kernel_np<8, 8>(np);
    }
    return nullptr;   // (12 x 4: ~60 registers spilled -- declined, the launch flow runs; cgsStep on chip holds 8 x 512 rows per CU = 1.05 M rows)
}
static const void *oc_kernel(int rpt, int hpt, int np) {
    switch (rpt * 16 + hpt) {
        case 4 * 16 + 4: return oc_kernel_np<4, 4>(np);
        case 4 * 16 + 8: return oc_kernel_np<4, 8>(np);
        case 8 * 16 + 4: return oc_kernel_np<8, 4>(np);
        case 8 * 16 + 8: return oc_kernel_np<8, 8>(np);
        case 12 * 16 + 4: return oc_kernel_np<12, 4>(np);
        case 12 * 16 + 8: return oc_kernel_np<12, 8>(np);
    }
    return nullptr;
}

// the plan of matrix A (host side): nullptr-safe, never throws past its caller (no_throw around the entry point)
static void build_plan(sla_csr *A, OcPlan &pl) {
    sla_ctx *c = A->ctx;
    pl.ok = false;
    const int64_t n = A->n;
    if (c->collectives || c->nranks != 1) { pl.note = "row-sharded context (the persistent launch has no exchange with other devices)"; return; }
    if (A->m != A->n || A->rows != A->n || A->row_begin != 0) { pl.note = "not a whole square matrix"; return; }
    if (!A->use_wdia || A->wd_vv || !A->wd_lds || !A->d_wum || A->npairs < 1 || A->npairs > kOcMaxPairs || A->wd_uni.n != A->npairs) {
        pl.note = "not a constant-coefficient matrix of <= 8 (offset, value) pairs";
        return;
    }
    if (n >= ((int64_t)1 << 31) - 1) { pl.note = "too many rows"; return; }
    const int np = A->npairs;
    // pair table and per-row presence masks back from the device: 1 KB + 1 byte per row
    std::vector<int32_t> doff(256);
    const int64_t nsl = A->nslices;
    std::vector<uint64_t> wum((size_t)nsl * 16);
    if (hipMemcpy(doff.data(), A->d_vdoff, sizeof(int32_t) * 256, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wum.data(), A->d_wum, sizeof(uint64_t) * wum.size(), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        pl.note = "could not read the pair masks back";
        return;
    }
    std::vector<uint8_t> mask((size_t)n);
    par_rows(n, 128, [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t *w = &wum[(size_t)(i >> 7) * 16 + (size_t)(i & 1) * 8];
            const int lane = (int)((i & 127) >> 1);
            unsigned m = 0;
            for (int k = 0; k < np; ++k) m |= (unsigned)((w[k] >> lane) & 1) << k;
            mask[(size_t)i] = (uint8_t)m;
        }
    });
    int64_t off[kOcMaxPairs];
    int64_t lo = 0, hi = 0;
    for (int k = 0; k < np; ++k) {
        off[k] = doff[(size_t)k];
        pl.val[k] = A->wd_uni.val[k];
        lo = std::min(lo, off[k]);
        hi = std::max(hi, off[k]);
    }
    const int cus = std::max(1, std::min(c->n_cu, c->onchip_grid > 0 ? c->onchip_grid : c->n_cu));
    const int64_t rows_cap = std::min<int64_t>(12 * T, c->onchip_rows > 0 ? c->onchip_rows : 12 * T);
    if (n > (int64_t)cus * rows_cap) { pl.note = "more rows than the workgroups' registers hold (" + std::to_string((long long)cus * rows_cap) + ")"; return; }
    // ---- geometry: consecutive rows where the band fits the LDS, bricks of the stencil's grid otherwise ----
    Geo g;
    bool have = false;
    std::string why0;
    {
        int64_t R = std::max<int64_t>((n + cus - 1) / cus, std::min<int64_t>(64, n));
        R = std::min(R, rows_cap);
        const int64_t L = -lo + R + hi;
        const int G = (int)((n + R - 1) / R);
        if (G <= cus && L < 60000 && oc_lds_bytes((int)L) <= kOcLdsMax && class_ok(rpt_class(R), hpt_class(-lo + hi)) && c->onchip_bricks != 2) {
            g.mode = 0;
            g.R = R;
            g.G = G;
            g.L = (int)L;
            for (int k = 0; k < np; ++k) g.loff[k] = (int)off[k];
            have = true;
        } else {
            why0 = "the band (" + std::to_string((long long)(-lo + hi)) + " rows of halo) does not fit the LDS with " + std::to_string((long long)R) + " rows per workgroup";
        }
    }
    if (!have && c->onchip_bricks != 0) {
        // 5- / 7-point-like: offsets {0, +-1, +-s2 (, +-s3)} with s2 | s3: an nx x ny x nz grid in row-major order
        int64_t s2 = 0, s3 = 0;
        bool sten = true;
        std::vector<int64_t> mags;
        for (int k = 0; k < np; ++k)
            if (off[k] != 0 && std::find(mags.begin(), mags.end(), std::llabs((long long)off[k])) == mags.end()) mags.push_back(std::llabs((long long)off[k]));
        std::sort(mags.begin(), mags.end());
        if (mags.empty() || mags[0] != 1 || mags.size() < 2 || mags.size() > 3) sten = false;
        if (sten) {
            s2 = mags[1];
            s3 = mags.size() == 3 ? mags[2] : 0;
            if (s2 < 2 || (s3 && (s3 % s2 != 0 || s3 / s2 < 2))) sten = false;
        }
        if (!sten) {
            pl.note = why0 + "; offsets are not those of a 5- / 7-point stencil on a grid (no bricks)";
            return;
        }
        g.nx = s2;
        g.ny = s3 ? s3 / s2 : (n + s2 - 1) / s2;
        g.nz = s3 ? (n + s3 - 1) / s3 : 1;
        // bricks: pieces per dimension with at most `cus` bricks; cheapest = fewest rows per thread, then fewest halo cells
        double best = 1e300;
        for (int64_t pz = 1; pz <= std::min<int64_t>(g.nz, cus); ++pz)
            for (int64_t py = 1; py <= std::min<int64_t>(g.ny, cus / pz); ++py)
                for (int64_t px = 1; px <= std::min<int64_t>(g.nx, cus / (pz * py)); ++px) {
                    const int64_t bx = (g.nx + px - 1) / px, by = (g.ny + py - 1) / py, bz = (g.nz + pz - 1) / pz;
                    if (bx < 2) continue;   // (a cell's +-1 neighbours must be distinct local cells)
                    const int64_t own = bx * by * bz, halo = 2 * (bx * by * (g.nz > 1) + by * bz + bx * bz * (g.ny > 1));
                    const int64_t L = (bx + 2) * (by + 2) * (g.nz > 1 ? bz + 2 : 1);
                    if (own > rows_cap || !class_ok(rpt_class(own), hpt_class(halo)) || L >= 60000 || oc_lds_bytes((int)L) > kOcLdsMax) continue;
                    const double cost = rpt_class(own) * 1e9 + hpt_class(halo) * 1e7 + (double)std::max<int64_t>(own, 2 * T) * 100 + (double)halo;   // (no point in bricks below two rows per thread)
                    if (cost < best) {
                        best = cost;
                        g.bx = (int)bx; g.by = (int)by; g.bz = (int)bz;
                    }
                }
        if (best == 1e300) { pl.note = why0 + "; no brick shape fits either"; return; }
        g.mode = 1;
        const int LX = g.bx + 2, LY = g.by + 2;
        g.L = LX * LY * (g.nz > 1 ? g.bz + 2 : 1);
        for (int k = 0; k < np; ++k) {
            int64_t o = off[k];
            int dz = 0;
            if (s3) { dz = (int)std::llround((double)o / (double)s3); o -= (int64_t)dz * s3; }
            const int dy = (int)std::llround((double)o / (double)s2);
            o -= (int64_t)dy * s2;
            if (std::llabs((long long)o) > 1 || std::abs(dy) > 1 || std::abs(dz) > 1) { pl.note = "offset outside the brick padding"; return; }
            g.loff[k] = (dz * LY + dy) * LX + (int)o;
        }
        have = true;
    }
    if (!have) { pl.note = why0; return; }
    // ---- blocks: (row, local cell) lists ----
    // (everything per block below runs on the library's host threads -- par_rows over the blocks: the plan of a 1 M-row matrix took 48 ms on one
    // thread, five times the 200 iterations of the linSolve0 call that asked for it)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> blocks;
    if (g.mode == 0) {
        blocks.resize((size_t)g.G);
        par_rows(g.G, 1, [&](int, int64_t b0, int64_t b1) {
            for (int64_t b = b0; b < b1; ++b) {
                const int64_t r0 = b * g.R, r1 = std::min(n, r0 + g.R);
                auto &blk = blocks[(size_t)b];
                blk.resize((size_t)(r1 - r0));
                for (int64_t i = r0; i < r1; ++i) blk[(size_t)(i - r0)] = {(int32_t)i, (int32_t)(-lo + (i - r0))};
            }
        }, 2);
    } else {
        const int LX = g.bx + 2, LY = g.by + 2, zpad = g.nz > 1 ? 1 : 0;
        struct Org { int64_t x0, y0, z0; };
        std::vector<Org> orgs;
        for (int64_t z0 = 0; z0 < g.nz; z0 += g.bz)
            for (int64_t y0 = 0; y0 < g.ny; y0 += g.by)
                for (int64_t x0 = 0; x0 < g.nx; x0 += g.bx)
                    if ((z0 * g.ny + y0) * g.nx + x0 < n) orgs.push_back({x0, y0, z0});   // (a brick's first cell is its lowest row: the brick holds a row iff that one exists)
        blocks.resize(orgs.size());
        par_rows((int64_t)orgs.size(), 1, [&](int, int64_t q0, int64_t q1) {
            for (int64_t q = q0; q < q1; ++q) {
                const Org o = orgs[(size_t)q];
                auto &blk = blocks[(size_t)q];
                for (int64_t z = o.z0; z < std::min<int64_t>(g.nz, o.z0 + g.bz); ++z)
                    for (int64_t y = o.y0; y < std::min<int64_t>(g.ny, o.y0 + g.by); ++y)
                        for (int64_t x = o.x0; x < std::min<int64_t>(g.nx, o.x0 + g.bx); ++x) {
                            const int64_t row = (z * g.ny + y) * g.nx + x;
                            if (row < n) blk.push_back({(int32_t)row, (int32_t)(((z - o.z0 + zpad) * LY + (y - o.y0 + 1)) * LX + (x - o.x0 + 1))});
                        }
            }
        }, 2);
        g.G = (int)blocks.size();
        if (g.G > cus) { pl.note = "brick count exceeds the workgroups"; return; }
    }
    // ---- halo cells, boundary rows; every entry must land on the cell that mirrors its column ----
    const int G = g.G, L = g.L;
    std::vector<uint8_t> needed((size_t)n, 0);   // (set from several threads: a byte that only ever becomes 1 -- relaxed atomic stores)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> halos((size_t)G);
    int64_t own_max = 0, halo_max = 0;
    std::atomic<int> any_bad{0};
    par_rows(G, 1, [&](int, int64_t b0, int64_t b1) {
        std::vector<int32_t> cell_row((size_t)L, -1);   // this thread's scratch: which row a local cell mirrors in the block under way
        for (int64_t b = b0; b < b1; ++b) {
            auto &blk = blocks[(size_t)b];
            for (auto &rc : blk) cell_row[(size_t)rc.second] = rc.first;
            bool bad = false;
            for (auto &rc : blk) {
                const unsigned m = mask[(size_t)rc.first];
                for (int k = 0; k < np && !bad; ++k)
                    if ((m >> k) & 1) {
                        const int64_t cc = (int64_t)rc.second + g.loff[k], col = (int64_t)rc.first + off[k];
                        if (cc < 0 || cc >= L || col < 0 || col >= n) { bad = true; break; }
                        if (cell_row[(size_t)cc] < 0) {
                            cell_row[(size_t)cc] = (int32_t)col;
                            halos[(size_t)b].push_back({(int32_t)cc, (int32_t)col});
                            __atomic_store_n(&needed[(size_t)col], (uint8_t)1, __ATOMIC_RELAXED);
                        } else if (cell_row[(size_t)cc] != col) {
                            bad = true;
                        }
                    }
                if (bad) break;
            }
            for (auto &rc : blk) cell_row[(size_t)rc.second] = -1;
            for (auto &h : halos[(size_t)b]) cell_row[(size_t)h.first] = -1;
            if (bad) any_bad.store(1, std::memory_order_relaxed);
        }
    }, 2);
    if (any_bad.load()) { pl.note = "an entry's column is not where the local cell layout expects it (a wrapped or irregular stencil)"; return; }
    for (int b = 0; b < G; ++b) {
        own_max = std::max<int64_t>(own_max, (int64_t)blocks[(size_t)b].size());
        halo_max = std::max<int64_t>(halo_max, (int64_t)halos[(size_t)b].size());
    }
    const int rpt = rpt_class(own_max), hpt = hpt_class(std::max<int64_t>(halo_max, 1));
    if (!class_ok(rpt, hpt)) { pl.note = "block or halo too large for the instantiated kernels"; return; }
    // ---- tables ----
    std::vector<uint32_t> own_cm((size_t)G * rpt * T), halo_cell((size_t)G * hpt * T, (uint32_t)L);
    std::vector<int32_t> own_row((size_t)G * rpt * T, 0), halo_row((size_t)G * hpt * T, 0), halo_src((size_t)G * hpt * T, 0);
    std::vector<int32_t> slot_of((size_t)n, 0);
    int64_t nbound = 0;
    const unsigned fullm = (1u << np) - 1;
    std::atomic<long long> nbound_a{0};
    par_rows(G, 1, [&](int, int64_t b0, int64_t b1) {
        long long nb = 0;
        for (int64_t b = b0; b < b1; ++b) {
            auto &blk = blocks[(size_t)b];
            size_t j = (size_t)b * rpt * T;
            for (size_t q = 0; q < (size_t)rpt * T; ++q) own_cm[j + q] = (uint32_t)blk[0].second;   // (no row: mask 0, a cell to read around)
            // boundary rows first (their write-through stores are in flight while the rest is folded), rows with all their entries last
            // (whole wavefronts of interior rows skip the mask tests): a stable three-way partition of the block's rows
            std::vector<std::pair<int32_t, int32_t>> ord;
            ord.reserve(blk.size());
            for (int cls = 0; cls < 3; ++cls)
                for (auto &rc : blk) {
                    const int kc = needed[(size_t)rc.first] ? 0 : (mask[(size_t)rc.first] == fullm ? 2 : 1);
                    if (kc == cls) ord.push_back(rc);
                }
            blk.swap(ord);
            for (auto &rc : blk) {
                own_cm[j] = (uint32_t)rc.second | ((uint32_t)mask[(size_t)rc.first] << 16) | ((uint32_t)needed[(size_t)rc.first] << 24) | (1u << 25);
                own_row[j] = rc.first;
                slot_of[(size_t)rc.first] = (int32_t)j;
                nb += needed[(size_t)rc.first];
                ++j;
            }
        }
        nbound_a.fetch_add(nb, std::memory_order_relaxed);
    }, 2);
    nbound = nbound_a.load();
    par_rows(G, 1, [&](int, int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            size_t j = (size_t)b * hpt * T;
            for (auto &h : halos[(size_t)b]) {
                halo_cell[j] = (uint32_t)h.first;
                halo_row[j] = h.second;
                halo_src[j] = slot_of[(size_t)h.second];
                ++j;
            }
        }
    }, 2);
    // ---- co-residency: one workgroup per CU must be resident for the counter barrier to complete ----
    const void *kern = oc_kernel(rpt, hpt, np);
    const size_t lds = oc_lds_bytes(L);
    if (!kern) { pl.note = "no kernel instantiation"; return; }
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOcLdsMax) != hipSuccess) { (void)hipGetLastError(); pl.note = "the device does not grant the kernel its LDS"; return; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); pl.note = "the occupancy query does not grant one workgroup per CU"; return; }
    if (G > c->n_cu * per_cu) { pl.note = "more workgroups than can be resident"; return; }
    // ---- device side ----
    auto up = [&](void **dst, const void *src, size_t bytes) {
        return dev_malloc(c, dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool okd = up((void **)&pl.d_own_cm, own_cm.data(), own_cm.size() * 4) && up((void **)&pl.d_own_row, own_row.data(), own_row.size() * 4) &&
               up((void **)&pl.d_halo_cell, halo_cell.data(), halo_cell.size() * 4) && up((void **)&pl.d_halo_row, halo_row.data(), halo_row.size() * 4) &&
               up((void **)&pl.d_halo_src, halo_src.data(), halo_src.size() * 4);
    okd = okd && dev_malloc(c, (void **)&pl.d_pubA, sizeof(double) * own_cm.size()) == hipSuccess && dev_malloc(c, (void **)&pl.d_pubS, sizeof(double) * own_cm.size()) == hipSuccess &&
          dev_malloc(c, (void **)&pl.d_parts, sizeof(double) * 8 * (size_t)G) == hipSuccess && dev_malloc(c, (void **)&pl.d_bar, sizeof(unsigned) * (32 * 17 + 256)) == hipSuccess;
    if (!okd) { (void)hipGetLastError(); pl.note = "device allocation of the plan failed"; return; }
    pl.G = G; pl.L = L; pl.rpt = rpt; pl.hpt = hpt; pl.np = np; pl.mode = g.mode;
    pl.bx = g.bx; pl.by = g.by; pl.bz = g.bz;
    pl.own_max = own_max; pl.halo_max = halo_max; pl.nbound = nbound;
    for (int k = 0; k < np; ++k) pl.loff[k] = g.loff[k];
    pl.lds_bytes = lds;
    pl.ok = true;
    char buf[320];
    if (g.mode == 0)
        snprintf(buf, sizeof buf, "onchip: %d workgroups x %d threads, %lld consecutive rows each (+ %lld halo cells), %d x %d slots per thread, %zu KiB of LDS, %.1f %% boundary rows",
                 G, T, (long long)own_max, (long long)halo_max, rpt, hpt, lds >> 10, 100.0 * (double)nbound / (double)n);
    else
        snprintf(buf, sizeof buf, "onchip: %d workgroups x %d threads, bricks %d x %d x %d of a %lld x %lld x %lld grid (%lld rows + %lld halo cells at most), %d x %d slots per thread, %zu KiB of LDS, %.1f %% boundary rows",
                 G, T, g.bx, g.by, g.bz, (long long)g.nx, (long long)g.ny, (long long)g.nz, (long long)own_max, (long long)halo_max, rpt, hpt, lds >> 10, 100.0 * (double)nbound / (double)n);
    pl.note = buf;
}

}  // namespace

void onchip_plan_free(OcPlan *p) {
    if (!p) return;
    void *ptrs[] = {p->d_own_cm, p->d_own_row, p->d_halo_cell, p->d_halo_row, p->d_halo_src, p->d_pubA, p->d_pubS, p->d_parts, p->d_bar};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
}

// Can steps of S run on chip?  BiCGSTAB (with the fused flows on: the on-chip step IS the fused K4+K5 flow) or CGS, a single-rank context, no
// residual evaluation pending, and a plan for the matrix.  res: linSolve0's loop -- step, true residual, test -- inside the launch.
static const void *oc_pick(int method, bool res, const OcPlan &pl) {
    if (method == SLA_CGS_) return res ? oc_kernel_cgs_res(pl.rpt, pl.hpt, pl.np) : oc_kernel_cgs(pl.rpt, pl.hpt, pl.np);
    return res ? oc_kernel_res(pl.rpt, pl.hpt, pl.np) : oc_kernel(pl.rpt, pl.hpt, pl.np);
}
bool onchip_usable(sla_solver *S, bool res) {
    sla_ctx *c = S->ctx;
    if (c->onchip == 0 || (S->method != SLA_BICGSTAB_ && S->method != SLA_CGS_) || c->collectives || S->ghost || S->have_res) return false;
    if (S->method == SLA_BICGSTAB_ && !c->bicg_fuse45) return false;
    sla_csr *A = S->A;
    if (!A->oc) {
        A->oc = new OcPlan();
        const auto t0 = std::chrono::steady_clock::now();
        try {
            build_plan(A, *A->oc);
        } catch (...) {
            A->oc->ok = false;
            A->oc->note = "host allocation failed while planning";
        }
        A->oc->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("SLA_DEBUG_ONCHIP")) fprintf(stderr, "[sla] %s (planned in %.1f ms)\n", A->oc->note.c_str(), A->oc->build_ms);
        c->onchip_plan_ms = A->oc->build_ms;
    }
    c->onchip_note = A->oc->note;
    if (!A->oc->ok) return false;
    OcPlan &pl = *A->oc;
    const int slot = (S->method == SLA_CGS_ ? 2 : 0) + (res ? 1 : 0);
    if (slot != 0 && pl.kstate[slot] == 0) {   // every kernel is its own function: LDS grant and co-residency are asked once per plan and kernel
        const char *what = S->method == SLA_CGS_ ? (res ? "linSolve0 CGS_" : "cgsStep") : "linSolve0 BICGSTAB_";
        const void *kern = oc_pick(S->method, res, pl);
        const size_t lds = oc_lds_bytes(pl.L, res);
        int per_cu = 0;
        pl.kstate[slot] = -1;
        if (!kern) pl.knote[slot] = std::string("no ") + what + " instantiation for " + std::to_string(pl.rpt) + " x " + std::to_string(pl.hpt) + " slots per thread";
        else if (lds > kOcLdsMax) pl.knote[slot] = std::string(what) + ": x of own + halo cells does not fit the LDS next to p and s";
        else if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOcLdsMax) != hipSuccess) { (void)hipGetLastError(); pl.knote[slot] = std::string("the device does not grant the ") + what + " kernel its LDS"; }
        else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, lds) != hipSuccess || per_cu < 1 || pl.G > c->n_cu * per_cu) { (void)hipGetLastError(); pl.knote[slot] = std::string("the occupancy query does not grant the ") + what + " kernel one workgroup per CU"; }
        else pl.kstate[slot] = 1;
    }
    if (slot != 0 && pl.kstate[slot] < 0) {
        c->onchip_note += "; " + pl.knote[slot];
        return false;
    }
    return true;
}

int launch_onchip_steps(sla_solver *S, int par, int k, bool res) {
    sla_ctx *c = S->ctx;
    const OcPlan &pl = *S->A->oc;
    OcArgs a{};
    a.own_cm = pl.d_own_cm;
    a.own_row = pl.d_own_row;
    a.halo_cell = pl.d_halo_cell;
    a.halo_row = pl.d_halo_row;
    a.halo_src = pl.d_halo_src;
    a.L = pl.L;
    a.np = pl.np;
    for (int q = 0; q < pl.np; ++q) { a.loff[q] = pl.loff[q]; a.val[q] = pl.val[q]; }
    a.x = S->x->d;
    a.r = S->r->d;
    a.p = S->p->d;
    a.u = S->method == SLA_CGS_ ? S->u->d : nullptr;
    a.rhat = S->r0hat->d;
    a.b = S->b->d;
    a.pubA = pl.d_pubA;
    a.pubS = pl.d_pubS;
    a.parts = pl.d_parts;
    a.bar = pl.d_bar;
    a.sc = S->d_sc;
    a.par = par;
    a.k = k;
    a.sync = c->onchip_sync;
    a.fault = c->onchip_fault;
    const void *kern = oc_pick(S->method, res, pl);
    if (!kern) return fail(SLA_ERR_INVALID, "launch_onchip_steps: no instantiation for this plan");
    SLA_HIP_TRY(hipMemsetAsync(pl.d_bar, 0, sizeof(unsigned) * (32 * 17 + 256), stream_of(c)));
    ProfScope prof(c, SLA_KERNEL_ONCHIP, true);
    void *params[] = {&a};
    const size_t lds = oc_lds_bytes(pl.L, res);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof_take(c, &e0, &e1))
        SLA_HIP_TRY(hipExtLaunchKernel(kern, dim3(pl.G), dim3(T), params, lds, stream_of(c), e0, e1, 0));
    else
        SLA_HIP_TRY(hipLaunchKernel(kern, dim3(pl.G), dim3(T), params, lds, stream_of(c)));
    SLA_HIP_TRY(hipGetLastError());
    c->onchip_launches += 1;
    return SLA_OK;
}

}  // namespace sla
