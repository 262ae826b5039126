// barrier_probe.cpp -- what does a grid-wide barrier cost on MI355X?  (round 2)
//
// A persistent whole-step BiCGSTAB kernel for launch-bound sizes (1 M rows: 42 us per step for 4-5 dependent launches) would
// replace every kernel boundary by a grid barrier.  This probe times the barrier alone: G co-resident workgroups, K rounds of
//   ticket   lane 0 of each workgroup: atomicAdd(counter, 1) (device scope), then spin (L1-bypassing loads) until the counter
//            reaches round * G; __syncthreads() on both sides
//   flags    no read-modify-write at all: workgroup b plain-stores the round number into its own slot, workgroup 0 polls the G
//            slots (one wavefront, coalesced) and then plain-stores the round into a release flag that everybody polls
// against the cost of a kernel boundary (K empty launches back to back on one stream).
//   hipcc --offload-arch=gfx950 -O3 -o tools/barrier_probe tools/barrier_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) ticket_kernel(unsigned *counter, int rounds, int *timeout) {
    const unsigned G = gridDim.x;
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * G) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { *timeout = 1; break; }
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) flags_kernel(int *slots, int *release, int rounds, int *timeout) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (tid == 0) __hip_atomic_store(slots + b, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (b == 0 && tid < 64) {   // the collector: one wavefront reads all slots until every one shows round r
            long spins = 0;
            for (;;) {
                int m = r;
                for (int i = tid; i < G; i += 64) m = min(m, __hip_atomic_load(slots + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
                for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off, 64));
                if (m >= r) break;
                if (++spins > 20000000) { *timeout = 1; break; }
            }
            if (tid == 0) __hip_atomic_store(release, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            long spins = 0;
            while (__hip_atomic_load(release, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { *timeout = 1; break; }
            }
        }
        __syncthreads();
    }
}

__global__ void empty_kernel(int *p) { if (p == nullptr && threadIdx.x == 12345) *p = 0; }

int main() {
    unsigned *counter;
    int *slots, *release, *timeout, h_timeout = 0;
    CK(hipMalloc(&counter, 256));
    CK(hipMalloc(&slots, sizeof(int) * 4096));
    CK(hipMalloc(&release, 256));
    CK(hipMalloc(&timeout, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = 200;
    auto timed = [&](auto launch) {
        CK(hipMemset(counter, 0, 256));
        CK(hipMemset(slots, 0, sizeof(int) * 4096));
        CK(hipMemset(release, 0, 256));
        CK(hipMemset(timeout, 0, 256));
        launch();   // warm
        CK(hipDeviceSynchronize());
        CK(hipMemset(counter, 0, 256));
        CK(hipMemset(slots, 0, sizeof(int) * 4096));
        CK(hipMemset(release, 0, 256));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h_timeout, timeout, sizeof(int), hipMemcpyDeviceToHost));
        return ms;
    };
    {
        const float ms = timed([&] { for (int i = 0; i < rounds; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int *)slots); });
        printf("kernel boundary (empty 256-workgroup launches back to back): %.2f us each\n", ms * 1e3 / rounds);
    }
    for (int G : {64, 256, 512, 1024}) {   // all co-resident: <= 4 workgroups of 256 threads per CU on 256 CUs
        float ms = timed([&] { hipLaunchKernelGGL(ticket_kernel, dim3(G), dim3(256), 0, 0, counter, rounds, timeout); });
        printf("G = %4d workgroups: ticket barrier %7.2f us%s", G, ms * 1e3 / rounds, h_timeout ? " (TIMED OUT)" : "");
        ms = timed([&] { hipLaunchKernelGGL(flags_kernel, dim3(G), dim3(256), 0, 0, slots, release, rounds, timeout); });
        printf("   flag barrier %7.2f us%s\n", ms * 1e3 / rounds, h_timeout ? " (TIMED OUT)" : "");
    }
    return 0;
}
