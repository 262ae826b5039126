// gather_share_probe.cpp -- round 5: does a wavefront whose 64 lanes gather CONSECUTIVE entries of a column-sorted run (so that
// neighbouring LANES of one instruction share 128-byte lines of x) need fewer trips to the L2 than 64 scattered lanes?
//
// profiles/r04_gather_locality_probe.txt dealt a CU-wide sorted run LANE BY LANE to the four wavefronts (entries that share a
// line sit in the same lane of different wavefronts) and saw nothing: 185 vs 179 G gathers/s.  The case it did not measure is the
// one a CU-wide tile sorted by column gives for free: 64 consecutive sorted entries per instruction.  At d entries per line the
// instruction touches ~64 (1 - e^-d) / d distinct lines.  Modes:
//   random          : indices uniform in the window
//   run d=...       : every workgroup walks one sorted run of density d entries per 128-byte line, its wavefronts taking alternate
//                     64-entry groups (group g of a 256-entry step goes to wavefront g)
// with the 12 B per entry stream of the SpMV (index + value), U groups in flight per wavefront, OCC workgroups of 256 per CU.
// Second part: the same with the products added into a CU-wide LDS array of row sums (ds_add_f64 / read-modify-write), rows random.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_share_probe tools/gather_share_probe.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

// MODE 0: products into a register; 1: LDS atomic add (row = second index stream); 2: LDS read-modify-write (no atomics)
template <int U, int OCC, int MODE, int WG>
__global__ void __launch_bounds__(WG, OCC) gather_kernel(const uint32_t *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x,
                                                          double *out, size_t n, int rows) {
    extern __shared__ double ys[];
    if (MODE) {
        for (int i = threadIdx.x; i < rows; i += WG) ys[i] = 0.0;
        __syncthreads();
    }
    double acc = 0.0;
    const size_t chunk = (size_t)WG * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        uint32_t c[U];
        double a[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { c[j] = __builtin_nontemporal_load(idx + base + threadIdx.x + j * WG); a[j] = __builtin_nontemporal_load(val + base + threadIdx.x + j * WG); }
        double v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = x[c[j] & 0x3ffffu];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (MODE == 0) acc += a[j] * v[j];
            else if (MODE == 1) unsafeAtomicAdd(&ys[(c[j] >> 18) % (unsigned)rows], a[j] * v[j]);
            else { double *p = &ys[(c[j] >> 18) % (unsigned)rows]; *p = *p + a[j] * v[j]; }
        }
    }
    if (MODE) {
        __syncthreads();
        for (int i = threadIdx.x; i < rows; i += WG) acc += ys[i];
    }
    out[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t xorshift() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t N = (size_t)1 << 27;             // 134 M gathers per launch
    const size_t wcols = (size_t)1 << 18;         // 2 MiB window of x (one panel of the tile form)
    const size_t wlines = wcols / 16;
    std::vector<uint32_t> h(N);
    uint32_t *d_idx; double *d_x, *d_val, *d_out;
    CK(hipMalloc(&d_idx, N * 4)); CK(hipMalloc(&d_x, wcols * 8)); CK(hipMalloc(&d_val, N * 8)); CK(hipMalloc(&d_out, 4096 * 1024 * 8));
    CK(hipMemset(d_x, 0, wcols * 8)); CK(hipMemset(d_val, 0, N * 8));
    const int ROWS = 19584;
    CK(hipFuncSetAttribute((const void *)gather_kernel<8, 1, 1, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)gather_kernel<8, 1, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)gather_kernel<8, 1, 1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)gather_kernel<8, 1, 1, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)gather_kernel<4, 1, 1, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("# %zu M gathers per launch from a 2 MiB window (%zu lines), 12 B per entry streamed; G gathers/s (best of 5)\n", N >> 20, wlines);
    const double dens[] = {0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 16.0};
    for (double d : dens) {
        // a run = one sweep of the window with d entries per line: E = d * wlines entries, sorted; runs follow each other
        if (d == 0.0) {
            for (size_t i = 0; i < N; ++i) h[i] = (uint32_t)(xorshift() % wcols) | (uint32_t)((xorshift() % ROWS) << 18);
        } else {
            const size_t E = (size_t)(d * wlines);
            std::vector<uint32_t> run(E);
            for (size_t base = 0; base < N; base += E) {
                for (size_t i = 0; i < E; ++i) run[i] = (uint32_t)(xorshift() % wcols);
                std::sort(run.begin(), run.end());
                for (size_t i = 0; i < E && base + i < N; ++i) h[base + i] = run[i] | (uint32_t)((xorshift() % ROWS) << 18);
            }
        }
        // NB the kernel's workgroup walks chunks of WG * U entries, wavefront w of the chunk's group j reading entries j * WG + w * 64 + lane:
        // consecutive entries per instruction, the workgroup's wavefronts side by side in the run -- but different workgroups are on different
        // stretches of the run (a run much longer than a chunk is shared by many workgroups, like a panel is shared by the XCD's CUs).
        CK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        auto rate = [&](double ms) { return (double)N / ms * 1e-6; };
#define RUNV(UU, OCC, MODE, WG, LDS) rate(time_ms([&] { hipLaunchKernelGGL((gather_kernel<UU, OCC, MODE, WG>), dim3(256 * OCC), dim3(WG), LDS, 0, d_idx, d_val, d_x, d_out, N, ROWS); }))
        char name[64];
        if (d == 0.0) snprintf(name, sizeof name, "random"); else snprintf(name, sizeof name, "run d=%-5.2f", d);
        printf("%-12s | reg: U8 occ1 %6.1f  U8 occ2 %6.1f  U8 wg1024 %6.1f | lds atomic: wg256 %6.1f  wg512 %6.1f  wg1024 %6.1f  wg1024 U4 %6.1f | lds rmw wg256 %6.1f\n", name,
               RUNV(8, 1, 0, 256, 0), RUNV(8, 2, 0, 256, 0), RUNV(8, 1, 0, 1024, 0),
               RUNV(8, 1, 1, 256, ROWS * 8), RUNV(8, 1, 1, 512, ROWS * 8), RUNV(8, 1, 1, 1024, ROWS * 8), RUNV(4, 1, 1, 1024, ROWS * 8),
               RUNV(8, 1, 2, 256, ROWS * 8));
        fflush(stdout);
    }
    return 0;
}
