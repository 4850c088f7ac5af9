// Bare MFMA loops, operands in registers, two waves per SIMD (512-thread workgroups, one per CU), random f16 data:
// v_mfma_f32_32x32x16_f16 (8 tiles x 16 acc regs) against v_mfma_f32_16x16x32_f16 (32 tiles x 4 acc regs), same FLOP per wave.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_bench.hip -o tools/mfma_shape_bench && tools/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k32(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[(blockIdx.x * 512 + threadIdx.x) * 6 + i];
    for (int i = 0; i < 2; ++i) b[i] = in[(blockIdx.x * 512 + threadIdx.x) * 6 + 4 + i];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 2) void k16(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    f16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = in[(blockIdx.x * 512 + threadIdx.x) * 12 + i];
    for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 512 + threadIdx.x) * 12 + 8 + i];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 8, iters = 2000;
    std::vector<_Float16> h((size_t)blocks * 512 * 12 * 8);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    f16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, (size_t)blocks * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave and iteration: 16 MFMAs x 32768 flop (32x32x16) = 32 x 16384 (16x16x32) = 524288 flop
            const double flop = (double)blocks * 8 * iters * 524288.0;
            printf("%s rep %d: %.3f ms  %.1f TFLOP/s\n", shape == 0 ? "32x32x16" : "16x16x32", rep, ms, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
