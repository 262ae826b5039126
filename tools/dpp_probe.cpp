// What does v_mov_b32_dpp wave_shr:1 do on gfx950?  Prints the source lane every lane received (lane 0: -1 = kept `old`).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xf, 0xf, false);   // row_shr:1
    out[128 + lane] = __shfl_up(lane, 1, 64);
}
int main() {
    int *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; ++r) { printf("%s:", r == 0 ? "wave_shr:1" : r == 1 ? "row_shr:1 " : "shfl_up 1 "); for (int i = 0; i < 64; ++i) printf(" %d", h[64 * r + i]); printf("\n"); }
    return 0;
}
