// Which XCD does workgroup b run on?  Does the assignment of a dispatch continue where the previous dispatch of the queue ended?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // XCC_ID, bits [3:0]
}
__global__ void filler(int *out) { if (threadIdx.x == 0 && out) out[0] = 1; }
int main() {
    const int G = 256;
    int *d, *e;
    hipMalloc(&d, G * 4);
    hipMalloc(&e, 1 << 20);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<int> h(G);
    auto probe = [&](const char *what) {
        hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, s, d);
        hipMemcpyAsync(h.data(), d, G * 4, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        int ok = 0;
        for (int b = 0; b < G; ++b) ok += (h[b] == (b & 7));
        printf("%-44s %3d of %d on XCD b %% 8; first 12:", what, ok, G);
        for (int b = 0; b < 12; ++b) printf(" %d", h[b]);
        printf("\n");
    };
    probe("fresh stream");
    for (int f : {1, 3, 5, 8, 13}) {
        hipLaunchKernelGGL(filler, dim3(f), dim3(64), 0, s, (int *)nullptr);
        char nm[64];
        snprintf(nm, sizeof nm, "after a %d-workgroup kernel", f);
        probe(nm);
    }
    hipMemsetAsync(e, 0, 8192, s);
    probe("after hipMemsetAsync 8 KiB");
    hipMemsetAsync(e, 0, 1 << 20, s);
    probe("after hipMemsetAsync 1 MiB");
    hipMemsetAsync(e, 0, 40, s);
    probe("after hipMemsetAsync 40 B");
    hipMemcpyAsync(e, d, 1024, hipMemcpyDeviceToDevice, s);
    probe("after hipMemcpyAsync D2D 1 KiB");
    return 0;
}
