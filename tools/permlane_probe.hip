// probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950 (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned x = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane32_swap(x, x + 100, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(x, x + 100, false, false);
    out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: p32 (%3u,%3u) p16 (%3u,%3u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
