// onchip_probe.cpp -- round 6: does a BiCGSTAB step with the solver state ON CHIP pay in the cache-resident regime?  (MI355X)
//
// BASELINE config 2 (1 M-row 5-point Poisson) runs at 37.5 us per step in three launches and one rank's N = 8 slab of
// config 4 (216 x 216 x 27) at 49.7 us -- fill / drain / launch gaps, not bytes.  This probe is the bare pattern of ONE persistent launch:
//   * one 1024-thread workgroup per CU owns a block of rows (2-D / banded: consecutive rows; 3-D: a brick) -- x, r0hat, r / s, p, Ap, As of
//     its rows live in REGISTERS, p and s of own + halo cells in two LDS arrays (the SpMV reads its operands from LDS);
//   * ghost-row flow: r, p, s are kept valid on the halo cells too, so a step needs TWO grid-wide synchronisations: (1) after Ap = A p
//     (carries the partial sums of Ap . r0hat and the boundary rows of Ap), (2) after As = A s (carries four sums and the boundary rows of As);
//   * boundary rows and partial sums are published with 8-byte agent-scope (write-through) stores and read back with agent-scope loads
//     after an XCD-hierarchical counter barrier: no release / acquire fence on the path.
// It runs REAL arithmetic: the iterates are compared with a host BiCGSTAB of the same formulas, so the hand-off protocol is checked
// word by word, every step.  Decision rule (VERDICT): if a step costs >= 30 us at 1 M rows, stop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/onchip_probe tools/onchip_probe.cpp
//   tools/onchip_probe [steps per launch = 20] [repeats = 5] [threads per workgroup = 1024 | 512]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

static int T = 1024;       // threads per workgroup (host side; the kernels carry it as TPB): 1024 = 16 wavefronts of 128 VGPRs, 512 = 8 of 256; one workgroup per CU
constexpr int kMaxPairs = 8;

struct OcArgs {
    const uint32_t *own_cm;    // [G][RPT * T]: local cell | pair mask << 16 | boundary << 24 | valid << 25
    const int32_t *own_row;    // [G][RPT * T]: global row
    const uint32_t *halo_cell; // [G][HPT * T]: local cell (slots without one: the dummy cell L)
    const int32_t *halo_row;   // [G][HPT * T]: global row the cell mirrors
    int L;                     // local cells (own + halo + padding) of a workgroup
    int np;
    int loff[kMaxPairs];       // local cell offset of pair k
    double val[kMaxPairs];
    double *x, *r, *p;
    const double *rhat;
    double *pubA, *pubS;       // boundary rows of Ap / As, by global row
    double *parts;             // [4][G]
    unsigned *bar;             // barrier words (zeroed before every launch)
    double *scal;              // [0] rho in, [1] rho out, [2] alpha, [3] omega, [4] beta
    int *fail;
    int k;
    int mode;                  // 0 full step, 1 barriers only (no arithmetic), 2 no barriers (arithmetic only: wrong results, phase cost)
    int fence;                 // 1: thread 0 runs an agent-scope acquire fence after every barrier (A/B)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid-wide synchronisation that also carries K sums.  Every wavefront has left its K partial sums in red[16 k + wavefront] (LDS) and has
// issued its write-through stores (boundary rows).  All wavefronts drain their stores; wavefront 0 folds the per-wavefront sums in a
// fixed tree, lane 0 publishes them in parts[k * G + b], drains that store too and arrives at an XCD-hierarchical counter barrier:
// workgroups with equal b % 8 (observed: one XCD) share an arrival counter, the last arriver of a group bumps the top counter, the last
// group releases everybody through per-group generation words.  Correctness does not depend on the placement.  No fence anywhere:
// everything another workgroup reads was stored and is loaded at agent scope (8-byte write-through stores, L1-bypassing loads).
template <int K, int NW>
__device__ __forceinline__ bool grid_sync(unsigned *bar, unsigned epoch, int *fail, int fence, const double *red, double *parts) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
        const int G = gridDim.x, l = threadIdx.x & 15;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = l < NW ? red[16 * k + l] : 0.0;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (threadIdx.x == 0) st_agent(parts + (size_t)k * G + blockIdx.x, s);
        }
        if (threadIdx.x == 0) {
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
            long spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { *fail = 1; ok = 0; break; }
            }
            if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_ok = ok;
        }
    }
    __syncthreads();
    return s_ok != 0;
}
// the sum of the G (<= 256) published partials of one quantity, formed by EVERY wavefront for itself in the same fixed order (lane j adds
// parts j, j + 64, j + 128, j + 192, then a butterfly): no LDS, no workgroup barrier, identical bits everywhere
__device__ __forceinline__ double wave_total(const double *parts, int G) {
    const int l = threadIdx.x & 63;
    double v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = l + 64 * m < G ? ld_agent(parts + l + 64 * m) : 0.0;
    double s = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// NP: the matrix's (offset, value) pairs (0: any number <= 8, masked path only)
template <int T, int RPT, int HPT, int NP>
__global__ void __launch_bounds__(T) onchip_bicgstab(OcArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int LA = (a.L + 2) & ~1;   // cells 0 .. L-1 + one dummy cell (L): what the slots of a short block write to
    double *P = lds, *S = lds + LA, *AH = S + LA, *red = AH + HPT * T;
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x, wave = t >> 6;
    // Branch-free slots: a slot without a row (short blocks) has pair mask 0, boundary 0, an in-range cell to read around and zeros
    // for its state, and writes to the dummy cell; a halo slot without a cell reads row 0 and writes the dummy cell.
    uint32_t cm[RPT];
    int32_t grow[RPT];
    double x[RPT], rh[RPT], r[RPT], p[RPT], ap[RPT], as[RPT];
    const int np = NP ? NP : a.np;
    int fullbits = 0;   // bit i: every lane of this wavefront holds all np entries in its row i (interior rows: no mask tests)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const size_t j = ((size_t)b * RPT + i) * T + t;
        cm[i] = a.own_cm[j];
        grow[i] = a.own_row[j];
        if (NP && __builtin_amdgcn_ballot_w64(((cm[i] >> 16) & 0xff) != (1u << np) - 1) == 0) fullbits |= 1 << i;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const bool v = (cm[i] >> 25) & 1;
        const double x0 = a.x[grow[i]], r0 = a.r[grow[i]], p0 = a.p[grow[i]], h0 = a.rhat[grow[i]];
        x[i] = v ? x0 : 0.0;
        r[i] = v ? r0 : 0.0;
        p[i] = v ? p0 : 0.0;
        rh[i] = v ? h0 : 0.0;
        ap[i] = as[i] = 0.0;
    }
    auto wcell = [&](uint32_t c) -> int { return ((c >> 25) & 1) ? (int)(c & 0xffff) : a.L; };
    uint32_t hc[HPT];
    int32_t hg[HPT];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const size_t j = ((size_t)b * HPT + i) * T + t;
        hc[i] = a.halo_cell[j];
        hg[i] = a.halo_row[j];
        P[hc[i]] = a.p[hg[i]];   // halo(p) and halo(r): the invariant every step starts from
        S[hc[i]] = a.r[hg[i]];
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) P[wcell(cm[i])] = p[i];
    __syncthreads();
    double rho = a.scal[0], alpha = 0.0, omega = 0.0, beta = 0.0;
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
            for (int k = 0; k < kMaxPairs; ++k) {
                if (k < np) {
                    const double pk = a.val[k] * V[cell + a.loff[k]];
                    y = ((m >> k) & 1) ? y + pk : y;
                }
            }
        }
        return y;
    };
    for (int step = 0; step < a.k; ++step) {
        // (opaque per step: nothing derived from the slot words -- LDS addresses, publish addresses, mask tests -- is hoisted out of the loop
        // and kept in registers next to the state)
#pragma unroll
        for (int i = 0; i < RPT; ++i) asm volatile("" : "+v"(cm[i]), "+v"(grow[i]));
#pragma unroll
        for (int i = 0; i < HPT; ++i) asm volatile("" : "+v"(hc[i]), "+v"(hg[i]));
        // ---- phase A: Ap = A p from LDS; Ap . r0hat; boundary rows of Ap out (the host puts them in the first slots: their stores are in
        //      flight while the other rows are folded) ----
        if (a.mode != 1) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                ap[i] = fold(P, cm[i], (fullbits >> i) & 1);
                acc += ap[i] * rh[i];
                if ((cm[i] >> 24) & 1) st_agent(a.pubA + grow[i], ap[i]);
                __builtin_amdgcn_sched_barrier(0);   // one row's LDS operands at a time: the wavefronts hide the latency, the registers hold the state
            }
            acc = wave_sum(acc);
            if ((t & 63) == 0) red[wave] = acc;
        }
        if (a.mode != 2 && !grid_sync<1, T / 64>(a.bar, ++epoch, a.fail, a.fence, red, a.parts)) return;
        // ---- phase B: alpha; s = r - alpha Ap on own + halo cells ----
        if (a.mode != 1) {
            double hv[HPT];
#pragma unroll
            for (int i = 0; i < HPT; ++i) hv[i] = ld_agent(a.pubA + hg[i]);
            alpha = rho / wave_total(a.parts, G);
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                AH[i * T + t] = hv[i];
                S[hc[i]] = __builtin_fma(-alpha, hv[i], S[hc[i]]);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                r[i] = __builtin_fma(-alpha, ap[i], r[i]);   // s
                S[wcell(cm[i])] = r[i];
            }
            __syncthreads();
            // ---- phase C: As = A s; four sums; boundary rows of As out ----
            double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                as[i] = fold(S, cm[i], (fullbits >> i) & 1);
                q[0] += as[i] * r[i];
                q[1] += as[i] * as[i];
                q[2] += as[i] * rh[i];
                q[3] += r[i] * rh[i];
                if ((cm[i] >> 24) & 1) st_agent(a.pubS + grow[i], as[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = wave_sum(q[k]);
                if ((t & 63) == 0) red[16 * k + wave] = q[k];
            }
        }
        if (a.mode != 2 && !grid_sync<4, T / 64>(a.bar, ++epoch, a.fail, a.fence, red, a.parts)) return;
        // ---- phase D: omega, rho', beta; x, r, p on own cells; r, p on halo cells ----
        if (a.mode != 1) {
            double hv[HPT];
#pragma unroll
            for (int i = 0; i < HPT; ++i) hv[i] = ld_agent(a.pubS + hg[i]);
            double q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = wave_total(a.parts + (size_t)k * G, G);
            omega = q[0] / q[1];
            const double rn = q[3] - omega * q[2];
            beta = rn / rho * alpha / omega;
            rho = rn;
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                const double rv = __builtin_fma(-omega, hv[i], S[hc[i]]);
                S[hc[i]] = rv;   // halo(r) of the next step
                P[hc[i]] = __builtin_fma(beta, __builtin_fma(-omega, AH[i * T + t], P[hc[i]]), rv);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                x[i] = __builtin_fma(omega, r[i], __builtin_fma(alpha, p[i], x[i]));
                r[i] = __builtin_fma(-omega, as[i], r[i]);
                p[i] = __builtin_fma(beta, __builtin_fma(-omega, ap[i], p[i]), r[i]);
                P[wcell(cm[i])] = p[i];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if ((cm[i] >> 25) & 1) {
            int g = grow[i];
            asm volatile("" : "+v"(g));   // (the store addresses are formed here, not kept from the loads at the head of the kernel)
            a.x[g] = x[i];
            a.r[g] = r[i];
            a.p[g] = p[i];
        }
    if (b == 0 && t == 0) { a.scal[1] = rho; a.scal[2] = alpha; a.scal[3] = omega; a.scal[4] = beta; }
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side: a constant-coefficient stencil as (offset, value) pairs + per-row presence, a partition into workgroup blocks, the tables
// ------------------------------------------------------------------------------------------------------------------------------
struct Stencil {
    int64_t n = 0;
    int np = 0;
    int64_t off[kMaxPairs];
    double val[kMaxPairs];
    std::vector<uint8_t> mask;   // per row: which pairs are present
};
static Stencil poisson2d(int nx, int ny) {
    Stencil s;
    s.n = (int64_t)nx * ny;
    s.np = 5;
    const int64_t o[5] = {-nx, -1, 0, 1, nx};
    const double v[5] = {-1, -1, 4, -1, -1};
    for (int k = 0; k < 5; ++k) { s.off[k] = o[k]; s.val[k] = v[k]; }
    s.mask.resize(s.n);
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i)
            s.mask[(size_t)j * nx + i] = (j > 0 ? 1 : 0) | (i > 0 ? 2 : 0) | 4 | (i < nx - 1 ? 8 : 0) | (j < ny - 1 ? 16 : 0);
    return s;
}
static Stencil laplace3d(int nx, int ny, int nz) {
    Stencil s;
    s.n = (int64_t)nx * ny * nz;
    s.np = 7;
    const int64_t o[7] = {-(int64_t)nx * ny, -nx, -1, 0, 1, nx, (int64_t)nx * ny};
    const double v[7] = {-1, -1, -1, 6, -1, -1, -1};
    for (int k = 0; k < 7; ++k) { s.off[k] = o[k]; s.val[k] = v[k]; }
    s.mask.resize(s.n);
    for (int z = 0; z < nz; ++z)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i)
                s.mask[((size_t)z * ny + j) * nx + i] = (z > 0 ? 1 : 0) | (j > 0 ? 2 : 0) | (i > 0 ? 4 : 0) | 8 | (i < nx - 1 ? 16 : 0) | (j < ny - 1 ? 32 : 0) | (z < nz - 1 ? 64 : 0);
    return s;
}
static void spmv_host(const Stencil &s, const std::vector<double> &x, std::vector<double> &y) {
    for (int64_t i = 0; i < s.n; ++i) {
        double acc = 0.0;
        for (int k = 0; k < s.np; ++k)
            if ((s.mask[i] >> k) & 1) { volatile double pk = s.val[k] * x[i + s.off[k]]; acc += pk; }
        y[i] = acc;
    }
}

struct Plan {
    int G = 0, L = 0, rpt = 0, hpt = 0;
    int loff[kMaxPairs];
    std::vector<uint32_t> own_cm, halo_cell;
    std::vector<int32_t> own_row, halo_row;
    int64_t nhalo_max = 0, nbound = 0, nown_max = 0;
};
// blocks[b] = the own rows of workgroup b with their local cells; loff = local offsets of the pairs
static bool make_plan(const Stencil &s, const std::vector<std::vector<std::pair<int64_t, int>>> &blocks, int L, const int *loff, Plan &pl) {
    pl.G = (int)blocks.size();
    pl.L = L;
    for (int k = 0; k < s.np; ++k) pl.loff[k] = loff[k];
    std::vector<uint8_t> needed(s.n, 0);
    std::vector<std::map<int, int64_t>> halos(pl.G);
    for (int b = 0; b < pl.G; ++b) {
        std::vector<int64_t> cell_row(L, -1);
        for (auto &rc : blocks[b]) cell_row[rc.second] = rc.first;
        for (auto &rc : blocks[b])
            for (int k = 0; k < s.np; ++k)
                if ((s.mask[rc.first] >> k) & 1) {
                    const int c = rc.second + loff[k];
                    const int64_t g = rc.first + s.off[k];
                    if (c < 0 || c >= L) { printf("plan: cell out of the box\n"); return false; }
                    if (cell_row[c] >= 0) {
                        if (cell_row[c] != g) { printf("plan: own cell mismatch\n"); return false; }
                    } else {
                        auto it = halos[b].find(c);
                        if (it != halos[b].end() && it->second != g) { printf("plan: halo cell mismatch\n"); return false; }
                        halos[b][c] = g;
                        needed[g] = 1;
                    }
                }
        pl.nown_max = std::max<int64_t>(pl.nown_max, (int64_t)blocks[b].size());
        pl.nhalo_max = std::max<int64_t>(pl.nhalo_max, (int64_t)halos[b].size());
    }
    pl.rpt = (int)((pl.nown_max + T - 1) / T);
    pl.hpt = (int)((pl.nhalo_max + T - 1) / T);
    pl.own_cm.assign((size_t)pl.G * pl.rpt * T, 0);
    pl.own_row.assign((size_t)pl.G * pl.rpt * T, 0);
    pl.halo_cell.assign((size_t)pl.G * pl.hpt * T, (uint32_t)L);   // (no cell: the dummy cell, row 0)
    pl.halo_row.assign((size_t)pl.G * pl.hpt * T, 0);
    for (int b = 0; b < pl.G; ++b) {
        size_t j = (size_t)b * pl.rpt * T;
        for (size_t q = 0; q < (size_t)pl.rpt * T; ++q) pl.own_cm[j + q] = (uint32_t)blocks[b][0].second;   // (no row: mask 0, a cell to read around)
        // boundary rows first (their write-through stores are in flight while the rest is folded), then rows with all their entries together
        // (whole wavefronts of interior rows skip the mask tests)
        std::vector<std::pair<int64_t, int>> ord = blocks[b];
        std::stable_sort(ord.begin(), ord.end(), [&](const std::pair<int64_t, int> &u, const std::pair<int64_t, int> &v) {
            const int ku = needed[u.first] ? 0 : (s.mask[u.first] == (1u << s.np) - 1 ? 2 : 1), kv = needed[v.first] ? 0 : (s.mask[v.first] == (1u << s.np) - 1 ? 2 : 1);
            return ku < kv;
        });
        for (auto &rc : ord) {
            pl.own_cm[j] = (uint32_t)rc.second | ((uint32_t)s.mask[rc.first] << 16) | ((uint32_t)needed[rc.first] << 24) | (1u << 25);
            pl.own_row[j] = (int32_t)rc.first;
            pl.nbound += needed[rc.first];
            ++j;
        }
        j = (size_t)b * pl.hpt * T;
        for (auto &h : halos[b]) {
            pl.halo_cell[j] = (uint32_t)h.first;
            pl.halo_row[j] = (int32_t)h.second;
            ++j;
        }
    }
    return true;
}
// consecutive rows per workgroup (2-D / banded): local cell = halo_lo + (row - first own row)
static bool plan_contiguous(const Stencil &s, int G, Plan &pl) {
    int64_t lo = 0, hi = 0;
    for (int k = 0; k < s.np; ++k) { lo = std::min(lo, s.off[k]); hi = std::max(hi, s.off[k]); }
    const int64_t R = (s.n + G - 1) / G;
    std::vector<std::vector<std::pair<int64_t, int>>> blocks(G);
    for (int b = 0; b < G; ++b)
        for (int64_t i = b * R; i < std::min(s.n, (b + 1) * R); ++i) blocks[b].push_back({i, (int)(-lo + (i - b * R))});
    int loff[kMaxPairs];
    for (int k = 0; k < s.np; ++k) loff[k] = (int)s.off[k];
    return make_plan(s, blocks, (int)(-lo + R + hi), loff, pl);
}
// bricks of bx x by x bz cells of an nx x ny x nz grid: local cell = padded-brick index
static bool plan_bricks(const Stencil &s, int nx, int ny, int nz, int bx, int by, int bz, Plan &pl) {
    const int LX = bx + 2, LY = by + 2, LZ = bz + 2;
    std::vector<std::vector<std::pair<int64_t, int>>> blocks;
    for (int z0 = 0; z0 < nz; z0 += bz)
        for (int y0 = 0; y0 < ny; y0 += by)
            for (int x0 = 0; x0 < nx; x0 += bx) {
                blocks.emplace_back();
                for (int z = z0; z < std::min(nz, z0 + bz); ++z)
                    for (int y = y0; y < std::min(ny, y0 + by); ++y)
                        for (int x = x0; x < std::min(nx, x0 + bx); ++x)
                            blocks.back().push_back({((int64_t)z * ny + y) * nx + x, ((z - z0 + 1) * LY + (y - y0 + 1)) * LX + (x - x0 + 1)});
            }
    int loff[kMaxPairs];
    for (int k = 0; k < s.np; ++k) {
        int64_t o = s.off[k];
        const int64_t pz = (int64_t)nx * ny;
        const int dz = (int)std::llround((double)o / pz);
        o -= dz * pz;
        const int dy = (int)std::llround((double)o / nx);
        o -= (int64_t)dy * nx;
        loff[k] = (dz * LY + dy) * LX + (int)o;
    }
    return make_plan(s, blocks, LX * LY * LZ, loff, pl);
}

struct HostState { std::vector<double> x, r, p, rhat; double rho; };
static void host_steps(const Stencil &s, HostState &h, int k) {
    const int64_t n = s.n;
    std::vector<double> ap(n), as(n);
    for (int it = 0; it < k; ++it) {
        spmv_host(s, h.p, ap);
        double d = 0.0;
        for (int64_t i = 0; i < n; ++i) d += ap[i] * h.rhat[i];
        const double alpha = h.rho / d;
        for (int64_t i = 0; i < n; ++i) h.r[i] = std::fma(-alpha, ap[i], h.r[i]);
        spmv_host(s, h.r, as);
        double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        for (int64_t i = 0; i < n; ++i) { q0 += as[i] * h.r[i]; q1 += as[i] * as[i]; q2 += as[i] * h.rhat[i]; q3 += h.r[i] * h.rhat[i]; }
        const double omega = q0 / q1, rn = q3 - omega * q2, beta = rn / h.rho * alpha / omega;
        h.rho = rn;
        for (int64_t i = 0; i < n; ++i) {
            h.x[i] = std::fma(omega, h.r[i], std::fma(alpha, h.p[i], h.x[i]));
            h.r[i] = std::fma(-omega, as[i], h.r[i]);
            h.p[i] = std::fma(beta, std::fma(-omega, ap[i], h.p[i]), h.r[i]);
        }
    }
}

static int g_generic = 0;   // 1: the masked any-pair-count instantiation (A/B)
template <int TPB, int RPT, int HPT, int NP>
static void launch_n(const OcArgs &a, int G, size_t lds) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void *)onchip_bicgstab<TPB, RPT, HPT, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); attr = true; }
    hipLaunchKernelGGL((onchip_bicgstab<TPB, RPT, HPT, NP>), dim3(G), dim3(TPB), lds, 0, a);
}
template <int TPB, int RPT, int HPT>
static void launch_t(const OcArgs &a, int G, size_t lds) {
    if (a.np == 5 && !g_generic) launch_n<TPB, RPT, HPT, 5>(a, G, lds);
    else if (a.np == 7 && !g_generic) launch_n<TPB, RPT, HPT, 7>(a, G, lds);
    else launch_n<TPB, RPT, HPT, 0>(a, G, lds);
}
static void launch(const OcArgs &a, const Plan &pl) {
    const size_t lds = sizeof(double) * ((size_t)2 * ((pl.L + 2) & ~1) + (size_t)pl.hpt * T + 64);
    const int key = (T == 1024 ? 1000 : 0) + pl.rpt * 16 + pl.hpt;
    switch (key) {
        case 1000 + 4 * 16 + 2: launch_t<1024, 4, 2>(a, pl.G, lds); break;
        case 1000 + 6 * 16 + 2: launch_t<1024, 6, 2>(a, pl.G, lds); break;
        case 8 * 16 + 4: launch_t<512, 8, 4>(a, pl.G, lds); break;
        case 11 * 16 + 4: launch_t<512, 11, 4>(a, pl.G, lds); break;
        default: printf("no instantiation for T=%d rpt=%d hpt=%d\n", T, pl.rpt, pl.hpt); exit(1);
    }
    CK(hipGetLastError());
}

static void run_case(const char *name, const Stencil &s, const Plan &pl, int k, int reps) {
    const int64_t n = s.n;
    printf("\n== %s: n = %lld, %d workgroups x %d threads, %d rows + %d halo cells per thread, L = %d cells (%.1f KiB of LDS), own <= %lld, halo <= %lld, boundary rows %.1f %%\n",
           name, (long long)n, pl.G, T, pl.rpt, pl.hpt, pl.L, (2.0 * pl.L + pl.hpt * T + 64) * 8 / 1024, (long long)pl.nown_max, (long long)pl.nhalo_max, 100.0 * pl.nbound / n);
    // b = A * 1, x0 = 0: r0 = b, p0 = r0, rhat = r0
    HostState h;
    h.x.assign(n, 0.0);
    std::vector<double> ones(n, 1.0), bvec(n);
    // (pad for the host SpMV's out-of-range reads: none -- masks keep it inside)
    spmv_host(s, ones, bvec);
    srand(7);
    for (int64_t i = 0; i < n; ++i) bvec[i] += 0.01 * ((rand() % 2001) - 1000) / 1000.0;
    h.r = bvec; h.p = bvec; h.rhat = bvec;
    h.rho = 0.0;
    for (int64_t i = 0; i < n; ++i) h.rho += bvec[i] * bvec[i];
    const double rho0 = h.rho;
    OcArgs a{};
    uint32_t *d_cm, *d_hc; int32_t *d_or, *d_hr;
    CK(hipMalloc(&d_cm, pl.own_cm.size() * 4)); CK(hipMemcpy(d_cm, pl.own_cm.data(), pl.own_cm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_or, pl.own_row.size() * 4)); CK(hipMemcpy(d_or, pl.own_row.data(), pl.own_row.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hc, pl.halo_cell.size() * 4)); CK(hipMemcpy(d_hc, pl.halo_cell.data(), pl.halo_cell.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hr, pl.halo_row.size() * 4)); CK(hipMemcpy(d_hr, pl.halo_row.data(), pl.halo_row.size() * 4, hipMemcpyHostToDevice));
    double *dx, *dr, *dp, *drh, *pubA, *pubS, *parts, *scal; unsigned *bar; int *fail;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dr, n * 8)); CK(hipMalloc(&dp, n * 8)); CK(hipMalloc(&drh, n * 8));
    CK(hipMalloc(&pubA, n * 8)); CK(hipMalloc(&pubS, n * 8)); CK(hipMalloc(&parts, 4 * 256 * 8)); CK(hipMalloc(&scal, 64));
    CK(hipMalloc(&bar, 32 * 17 * 4)); CK(hipMalloc(&fail, 4));
    CK(hipMemset(pubA, 0xff, n * 8)); CK(hipMemset(pubS, 0xff, n * 8));   // NaN: a word read before it was published shows
    a.own_cm = d_cm; a.own_row = d_or; a.halo_cell = d_hc; a.halo_row = d_hr;
    a.L = pl.L; a.np = s.np;
    for (int k2 = 0; k2 < s.np; ++k2) { a.loff[k2] = pl.loff[k2]; a.val[k2] = s.val[k2]; }
    a.x = dx; a.r = dr; a.p = dp; a.rhat = drh; a.pubA = pubA; a.pubS = pubS; a.parts = parts; a.bar = bar; a.scal = scal; a.fail = fail;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() {
        CK(hipMemcpy(dx, h.x.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, bvec.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dp, bvec.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drh, bvec.data(), n * 8, hipMemcpyHostToDevice));
        const double sc[5] = {rho0, 0, 0, 0, 0};
        CK(hipMemcpy(scal, sc, sizeof(sc), hipMemcpyHostToDevice));
    };
    auto timed = [&](int mode, int fence, int kk, bool check) {
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            reset();
            a.mode = mode; a.fence = fence; a.k = kk;
            CK(hipMemsetAsync(bar, 0, 32 * 17 * 4, 0)); CK(hipMemsetAsync(fail, 0, 4, 0));
            CK(hipEventRecord(e0, 0));
            launch(a, pl);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) { best = std::min(best, ms); sum += ms; }
            int hf; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            if (hf) { printf("   BARRIER TIMEOUT (mode %d)\n", mode); return; }
        }
        printf("   mode %d fence %d  k = %3d : %8.2f us per launch (best), %7.2f us per step (best), %7.2f (mean)\n", mode, fence, kk, best * 1e3, best * 1e3 / kk, sum / reps * 1e3 / kk);
        if (check) {
            std::vector<double> gx(n), gr(n), gp(n);
            CK(hipMemcpy(gx.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gr.data(), dr, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gp.data(), dp, n * 8, hipMemcpyDeviceToHost));
            HostState hh = h;
            hh.r = bvec; hh.p = bvec; hh.rhat = bvec; hh.rho = rho0;
            host_steps(s, hh, kk);
            auto rel = [&](const std::vector<double> &u, const std::vector<double> &v) {
                double d = 0, nv = 0; int bad = 0;
                for (int64_t i = 0; i < n; ++i) { if (!(u[i] == u[i])) ++bad; d += (u[i] - v[i]) * (u[i] - v[i]); nv += v[i] * v[i]; }
                if (bad) printf("   %d NaNs!\n", bad);
                return std::sqrt(d / nv);
            };
            printf("   vs the host BiCGSTAB after %d steps: |dx|/|x| = %.2e  |dr|/|r| = %.2e  |dp|/|p| = %.2e   (|r| = %.3e)\n", kk, rel(gx, hh.x), rel(gr, hh.r), rel(gp, hh.p),
                   std::sqrt([&] { double q = 0; for (double v : hh.r) q += v * v; return q; }()));
        }
    };
    timed(0, 0, 3, true);
    timed(0, 0, k, true);
    timed(0, 1, k, true);
    timed(0, 0, 5 * k, false);
    timed(1, 0, 5 * k, false);
    timed(1, 1, 5 * k, false);
    timed(2, 0, 5 * k, false);
    timed(0, 0, 1, false);
    for (void *q : {(void *)d_cm, (void *)d_or, (void *)d_hc, (void *)d_hr, (void *)dx, (void *)dr, (void *)dp, (void *)drh, (void *)pubA, (void *)pubS, (void *)parts, (void *)scal, (void *)bar, (void *)fail}) CK(hipFree(q));
}

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 5;
    T = argc > 3 ? atoi(argv[3]) : 1024;
    g_generic = argc > 4 ? atoi(argv[4]) : 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, cus);
    {
        Stencil s = poisson2d(1000, 1000);
        Plan pl;
        if (plan_contiguous(s, cus, pl)) run_case("BASELINE config 2: 1000 x 1000 5-point Poisson, consecutive rows per workgroup", s, pl, k, reps);
    }
    {
        Stencil s = laplace3d(216, 216, 27);
        Plan pl;
        if (plan_bricks(s, 216, 216, 27, 24, 24, 9, pl)) run_case("one N = 8 slab of BASELINE config 4: 216 x 216 x 27 7-point Laplacian, 24 x 24 x 9 bricks", s, pl, k, reps);
    }
    {
        Stencil s = laplace3d(100, 100, 100);
        Plan pl;
        if (plan_bricks(s, 100, 100, 100, 20, 20, 10, pl)) run_case("100^3 7-point Laplacian, 20 x 20 x 10 bricks (250 workgroups)", s, pl, k, reps);
    }
    return 0;
}
