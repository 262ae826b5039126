// Which XCD does workgroup b run on?  Prints HW_REG_XCC_ID for the first workgroups of a 2048-block launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // XCC_ID, bits [3:0]
}
int main() {
    const int G = 2048;
    int *d;
    hipMalloc(&d, G * 4);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d);
    std::vector<int> h(G);
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < G; ++b) ok += (h[b] == (b & 7));
    printf("blocks with xcc == b %% 8: %d of %d\nfirst 32:", ok, G);
    for (int b = 0; b < 32; ++b) printf(" %d", h[b]);
    printf("\n");
    return 0;
}
