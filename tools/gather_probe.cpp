// gather_probe.cpp -- what does the memory system give a random 8-byte gather?  (MI355X, round 2)
//
// The SpMV of BASELINE config 3a (10 M rows, 33 random columns per row) is 330 M gathers x[col] per launch.  This probe
// measures the ceiling of that access pattern alone: a coalesced stream of 4-byte indices (uniform in a window of W
// bytes of x) + one 8-byte gather each, summed per lane.  Variants: window size (inside one XCD's 4 MiB L2 ... the
// 256 MiB memory-side cache ... HBM), loads in flight per lane, cache policy of the gather (plain / nt / agent scope
// = L1 bypass), 16-byte gathers, gathers served from an LDS copy of the window.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

enum { PLAIN = 0, NT = 1, AGENT = 2, SYS = 3 };

template <int MODE>
__device__ __forceinline__ double ld(const double *p) {
    if constexpr (MODE == NT) return __builtin_nontemporal_load(p);
    else if constexpr (MODE == AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (MODE == SYS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}

// U gathers in flight per lane; persistent grid; each workgroup walks chunks of 256 * U indices
template <int U, int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) gather_kernel(const uint32_t *__restrict__ idx, const double *__restrict__ x, double *out, size_t n) {
    double acc = 0.0;
    const size_t chunk = 256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        uint32_t c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) c[j] = __builtin_nontemporal_load(idx + base + threadIdx.x + j * 256);
        double v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ld<MODE>(x + c[j]);
#pragma unroll
        for (int j = 0; j < U; ++j) acc += v[j];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// the SpMV's real stream mix: 4-byte index + 8-byte value per gather (12 B per entry through the same L1)
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

// the same with 16-byte gathers (pairs: two vectors interleaved, or a complex)
template <int U, int OCC>
__global__ void __launch_bounds__(256, OCC) gather16_kernel(const uint32_t *__restrict__ idx, const double2 *__restrict__ x, double *out, size_t n) {
    double acc = 0.0;
    const size_t chunk = 256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        uint32_t c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) c[j] = __builtin_nontemporal_load(idx + base + threadIdx.x + j * 256);
        double2 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = x[c[j] >> 1];
#pragma unroll
        for (int j = 0; j < U; ++j) acc += v[j].x + v[j].y;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// gathers served from LDS: the workgroup copies a 128 KiB window (16384 doubles) once, then streams 2-byte indices
template <int U>
__global__ void __launch_bounds__(1024) gather_lds_kernel(const uint16_t *__restrict__ idx, const double *__restrict__ x, double *out, size_t n) {
    extern __shared__ double sx[];
    for (int i = threadIdx.x; i < 16384; i += 1024) sx[i] = x[i];
    __syncthreads();
    double acc = 0.0;
    const size_t chunk = 1024 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        uint16_t c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) c[j] = __builtin_nontemporal_load(idx + base + threadIdx.x + j * 1024);
#pragma unroll
        for (int j = 0; j < U; ++j) acc += sx[c[j] & 16383];
    }
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
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

int main(int argc, char **argv) {
    const bool quick = argc > 1;   // one window, one variant: for counter passes
    const size_t N = (size_t)1 << 28;             // 268 M gathers per launch (config 3a: 330 M)
    const size_t XMAX = (size_t)160 << 20;        // doubles: up to 1.25 GiB of x
    std::vector<uint32_t> h(N);
    uint32_t *d_idx; uint16_t *d_idx16; double *d_x, *d_out;
    CK(hipMalloc(&d_idx, N * 4)); CK(hipMalloc(&d_idx16, N * 2)); CK(hipMalloc(&d_x, XMAX * 8)); CK(hipMalloc(&d_out, 8192 * 1024 * 8));
    CK(hipMemset(d_x, 0, XMAX * 8));
    {
        std::vector<uint16_t> h16(N);
        for (size_t i = 0; i < N; ++i) h16[i] = (uint16_t)(xorshift() & 16383);
        CK(hipMemcpy(d_idx16, h16.data(), N * 2, hipMemcpyHostToDevice));
    }
    const size_t windows_kb[] = {512, 1024, 2048, 3072, 4096, 8192, 81920, 1048576};
    printf("# %zu M gathers per launch; rates in G gathers/s (best of 5); grid = OCC x 256 CUs\n", N >> 20);
    for (size_t wkb : windows_kb) {
        if (quick && wkb != 2048) continue;
        const size_t wcols = wkb * 1024 / 8;
        for (size_t i = 0; i < N; ++i) h[i] = (uint32_t)(xorshift() % wcols);
        CK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        auto rate = [&](double ms) { return (double)N / ms * 1e-6; };
        if (quick) {
            printf("window 2048 KiB plain U=8 occ8 %6.1f G gathers/s\n", rate(time_ms([&] { hipLaunchKernelGGL((gather_kernel<8, PLAIN, 8>), dim3(2048), dim3(256), 0, 0, d_idx, d_x, d_out, N); })));
            return 0;
        }
#define RUN(U, MODE, OCC) rate(time_ms([&] { hipLaunchKernelGGL((gather_kernel<U, MODE, OCC>), dim3(256 * OCC), dim3(256), 0, 0, d_idx, d_x, d_out, N); }))
        printf("window %7zu KiB | plain U=4 occ8 %6.1f | U=8 occ8 %6.1f | U=8 occ4 %6.1f | U=16 occ4 %6.1f | U=16 occ2 %6.1f | nt U=8 %6.1f | agent U=8 %6.1f | sys U=8 %6.1f",
               wkb, RUN(4, PLAIN, 8), RUN(8, PLAIN, 8), RUN(8, PLAIN, 4), RUN(16, PLAIN, 4), RUN(16, PLAIN, 2), RUN(8, NT, 8), RUN(8, AGENT, 8), RUN(8, SYS, 8));
        if (wkb == 2048) {
            double *d_val; CK(hipMalloc(&d_val, N * 8)); CK(hipMemset(d_val, 0, N * 8));
#define RUNV(U, OCC) rate(time_ms([&] { hipLaunchKernelGGL((gatherv_kernel<U, OCC>), dim3(256 * OCC), dim3(256), 0, 0, d_idx, d_val, d_x, d_out, N); }))
            printf("\n   + 8-byte value stream (12 B per entry), window 2048 KiB: U=4 occ4 %6.1f | U=4 occ8 %6.1f | U=8 occ4 %6.1f | U=8 occ8 %6.1f | U=16 occ4 %6.1f\n   ",
                   RUNV(4, 4), RUNV(4, 8), RUNV(8, 4), RUNV(8, 8), RUNV(16, 4));
            CK(hipFree(d_val));
        }
        printf(" | 16B U=8 %6.1f\n",
               rate(time_ms([&] { hipLaunchKernelGGL((gather16_kernel<8, 8>), dim3(2048), dim3(256), 0, 0, d_idx, (const double2 *)d_x, d_out, N); })));
        fflush(stdout);
    }
    CK(hipFuncSetAttribute((const void *)gather_lds_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)gather_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    double m8 = time_ms([&] { hipLaunchKernelGGL((gather_lds_kernel<8>), dim3(256), dim3(1024), 131072, 0, d_idx16, d_x, d_out, N); });
    double m16 = time_ms([&] { hipLaunchKernelGGL((gather_lds_kernel<16>), dim3(256), dim3(1024), 131072, 0, d_idx16, d_x, d_out, N); });
    printf("LDS window 128 KiB, 2-byte indices | U=8 %6.1f | U=16 %6.1f  G gathers/s\n", (double)N / m8 * 1e-6, (double)N / m16 * 1e-6);
    return 0;
}
