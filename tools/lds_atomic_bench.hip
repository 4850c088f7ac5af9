// micro-benchmark: LDS atomic throughput on gfx950 (development tool). hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int SPREAD>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, float* out, int iters) {
    __shared__ float hf[4][2048];
    __shared__ unsigned hu[4][2048];
    __shared__ unsigned long long hl[4][1024];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 2048; i += 64) { hf[w][i] = 0; hu[w][i] = 0; if (i < 1024) hl[w][i] = 0; }
    int a = idx[(blockIdx.x * 256 + threadIdx.x) % 4096] % SPREAD;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) atomicAdd(&hf[w][a], v);
        else if (MODE == 1) atomicAdd(&hu[w][a], (unsigned)lane);
        else if (MODE == 2) atomicAdd(&hl[w][a & 1023], (unsigned long long)lane);
        else if (MODE == 3) hf[w][a] = v;                       // plain store for reference
        a = (a * 5 + 1) % SPREAD;
    }
    __syncthreads();
    float s = 0;
    for (int i = lane; i < 2048; i += 64) s += hf[w][i] + hu[w][i] + (i < 1024 ? (float)hl[w][i] : 0.f);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int SPREAD> void run(const char* name, const int* idx, float* out) {
    const int iters = 2000, blocks = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(blocks), dim3(256), 0, 0, idx, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(blocks), dim3(256), 0, 0, idx, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waveinstr = (double)blocks * 4 * iters;
    // cycles per wave-instruction per CU (256 CUs, 2.4 GHz)
    printf("%-28s spread=%4d: %.3f ms  -> %.1f CU-cycles per wave-instruction\n", name, SPREAD, ms, ms * 1e-3 * 2.4e9 * 256 / waveinstr);
}
int main() {
    std::vector<int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = rand();
    int* idx; float* out; hipMalloc(&idx, 4096 * 4); hipMalloc(&out, 2048 * 256 * 4);
    hipMemcpy(idx, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 2048>("ds_add_f32 random", idx, out); run<0, 352>("ds_add_f32 random", idx, out); run<0, 32>("ds_add_f32 random", idx, out); run<0, 4>("ds_add_f32 random", idx, out); run<0, 1>("ds_add_f32 same", idx, out);
    run<1, 2048>("ds_add_u32 random", idx, out); run<1, 352>("ds_add_u32 random", idx, out); run<1, 32>("ds_add_u32 random", idx, out); run<1, 1>("ds_add_u32 same", idx, out);
    run<2, 1024>("ds_add_u64 random", idx, out); run<2, 352>("ds_add_u64 random", idx, out); run<2, 32>("ds_add_u64 random", idx, out);
    run<3, 2048>("ds_write_b32 random", idx, out); run<3, 32>("ds_write_b32 random", idx, out);
    return 0;
}
