// How many bytes per clock can one CU pull through its vector-memory path with 16-byte-per-lane loads?
//   L2-resident working set (every load misses the 32 KB L1, hits the XCD's 4 MiB L2) vs L1-resident.
// Prints GB/s and B/clk/CU at the measured kernel time (clock from hipDeviceProp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int INFLIGHT>
__global__ void __launch_bounds__(256) probe(const d2 *__restrict__ base, size_t elems_per_block, int iters, double *sink) {
    const d2 *p = base + (size_t)blockIdx.x * elems_per_block;
    const int tid = threadIdx.x;
    d2 acc[INFLIGHT];
    for (int j = 0; j < INFLIGHT; ++j) acc[j] = d2{0.0, 0.0};
    const size_t span = elems_per_block;  // d2 elements
    size_t off = tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            const d2 v = p[(off + (size_t)j * 256) & (span - 1)];  // span is a power of two
            acc[j] += v;
        }
        off = (off + (size_t)INFLIGHT * 256) & (span - 1);
    }
    d2 s = acc[0];
    for (int j = 1; j < INFLIGHT; ++j) s += acc[j];
    if (s.x + s.y == 12345.678) sink[0] = s.x;
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate / 1e6;
    const size_t bytes = 1u << 30;
    d2 *buf;
    double *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 8);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("%d CUs, %.2f GHz (nominal)\n", cus, ghz);
    for (int blocks_per_cu : {4, 6, 8}) {
        for (size_t kb_per_block : {4, 16, 32, 64, 1024}) {      // 4 KB/block: L1-resident; 64 KB/block x 8 x 32 CUs/XCD = 16 MB/XCD > L2 ... see below
            const int grid = cus * blocks_per_cu;
            const size_t elems = kb_per_block * 1024 / 16;
            if ((size_t)grid * elems * 16 > bytes) continue;
            const int iters = 2000 / 8 * (kb_per_block >= 1024 ? 1 : 4);
            hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(256), 0, 0, buf, elems, 10, sink);
            hipEventRecord(a);
            hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(256), 0, 0, buf, elems, iters, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double total = (double)grid * 256 * 16 * 8 * iters;
            printf("blocks/CU %d  %5zu KB/block (%7.1f MB total, %6.2f MB per XCD): %8.1f GB/s = %5.1f B/clk/CU\n", blocks_per_cu, kb_per_block,
                   grid * elems * 16 / 1e6, grid * elems * 16 / 8e6, total / ms / 1e6, total / (ms * 1e-3) / cus / (ghz * 1e9));
        }
    }
    return 0;
}
