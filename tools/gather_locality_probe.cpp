// gather_locality_probe.cpp -- would config 3a's tile kernel gain from sharing x lines between the four wavefronts of a CU?  (round 4)
//
// spmv_tile_kernel gives every wavefront a private slice of 4896 rows; a (slice, 1 MiB panel) tile holds ~2090 entries on 8192 lines of
// x: 0.26 entries per line, every gather its own 128-byte line from the L2 (171 G gathers/s measured, the L2's line rate).  A CU-wide
// slice (4 x 4896 rows) would hold 1.02 entries per line; walked in COLUMN order by the four wavefronts side by side, entries that share a
// line would meet in the L1 (same instruction or a neighbouring wavefront's at the same moment): (1 - e^-1.02) / 1.02 = 0.63 line requests
// per gather.  This probe measures what the memory system makes of that on the bare pattern, before anyone rewrites the tile form:
//   random      : indices uniform in the window (what the tile kernel's (layer, row) order amounts to)
//   sorted/wave : every wavefront walks its own column-sorted run of E entries over the window (a tile sorted by column: 0.26 per line)
//   sorted/cu   : the four wavefronts of a workgroup walk ONE column-sorted run of 4 E entries, dealt lane by lane (1.02 per line)
// one workgroup of 4 wavefronts per CU (OCC 1, as the tile kernel) and two; U gathers in flight per lane; 12 B per entry streamed.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_locality_probe tools/gather_locality_probe.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int U, int OCC>
__global__ void __launch_bounds__(256, OCC) gatherv_kernel(const uint32_t *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x, double *out, size_t n) {
    double acc = 0.0;
    const size_t chunk = 256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        uint32_t c[U];
        double a[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { c[j] = __builtin_nontemporal_load(idx + base + threadIdx.x + j * 256); a[j] = __builtin_nontemporal_load(val + base + threadIdx.x + j * 256); }
        double v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = x[c[j]];
#pragma unroll
        for (int j = 0; j < U; ++j) acc += a[j] * v[j];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
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
    const size_t wcols = (size_t)1 << 17;         // 1 MiB window of x (one panel)
    const size_t E = 2048;                        // entries of one wavefront's tile
    std::vector<uint32_t> h(N), run(4 * E);
    uint32_t *d_idx; double *d_x, *d_val, *d_out;
    CK(hipMalloc(&d_idx, N * 4)); CK(hipMalloc(&d_x, wcols * 8 * 4)); CK(hipMalloc(&d_val, N * 8)); CK(hipMalloc(&d_out, 4096 * 256 * 8));
    CK(hipMemset(d_x, 0, wcols * 8 * 4)); CK(hipMemset(d_val, 0, N * 8));
    printf("# %zu M gathers per launch from a 1 MiB window, 12 B per entry streamed; G gathers/s (best of 5)\n", N >> 20);
    for (int mode = 0; mode < 3; ++mode) {
        // a workgroup's chunk j holds entries [j * 256, j * 256 + 256): thread t = wave * 64 + lane
        const size_t U = 8, chunk = 256 * U;       // layout below assumes chunks of 2048 = E entries per workgroup step of U = 8
        for (size_t base = 0; base + 4 * E <= N; base += 4 * E) {
            if (mode == 0) {
                for (size_t i = 0; i < 4 * E; ++i) h[base + i] = (uint32_t)(xorshift() % wcols);
            } else if (mode == 1) {                // per wavefront: its own sorted run of E entries
                for (int w = 0; w < 4; ++w) {
                    for (size_t i = 0; i < E; ++i) run[i] = (uint32_t)(xorshift() % wcols);
                    std::sort(run.begin(), run.begin() + E);
                    // wave w reads, in chunk q (of 4 chunks of 2048 per 4 E entries), group j: its entries q * (E / 4) + j * 64 + lane
                    for (size_t i = 0; i < E; ++i) {
                        const size_t q = i / (E / 4), r = i % (E / 4), j = r / 64, lane = r % 64;
                        h[base + q * chunk + j * 256 + (size_t)w * 64 + lane] = run[i];
                    }
                }
            } else {                               // the workgroup: ONE sorted run of 4 E entries, dealt lane by lane to the four wavefronts
                for (size_t i = 0; i < 4 * E; ++i) run[i] = (uint32_t)(xorshift() % wcols);
                std::sort(run.begin(), run.end());
                for (size_t i = 0; i < 4 * E; ++i) {
                    const size_t g = i / 256, p = i % 256, lane = p / 4, w = p % 4;   // 256 consecutive sorted entries per (chunk, group)
                    h[base + g * 256 + w * 64 + lane] = run[i];
                }
            }
        }
        CK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        auto rate = [&](double ms) { return (double)N / ms * 1e-6; };
#define RUNV(UU, OCC) rate(time_ms([&] { hipLaunchKernelGGL((gatherv_kernel<UU, OCC>), dim3(256 * OCC), dim3(256), 0, 0, d_idx, d_val, d_x, d_out, N); }))
        printf("%-12s | U=8 occ1 %6.1f | U=8 occ2 %6.1f | U=8 occ4 %6.1f | U=8 occ8 %6.1f\n", mode == 0 ? "random" : mode == 1 ? "sorted/wave" : "sorted/cu",
               RUNV(8, 1), RUNV(8, 2), RUNV(8, 4), RUNV(8, 8));
        fflush(stdout);
    }
    return 0;
}
