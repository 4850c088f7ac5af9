// Micro-benchmark behind DESIGN.md §5 "FPFH: each pair once": what do the symmetric deposits cost?
// Model of the pair-once k_spfh: a wave owns point p of object o and, for every later neighbour q (64 per wave instruction, M/2 per
// point), adds 1 to three of the 33 bin counters of q's row plus its neighbour count (u32, 136-byte rows, order-free).
// Rows of one object are contiguous (16384 x 136 B = 2.2 MB); neighbours are spatially close, modelled as a window of +-W rows.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rate.hip -o tools/micro/atomic_rate && tools/micro/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_deposit(uint32_t* rows, int n_per_obj, int half_m, int window, int n_atomics) {
    const int o = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= n_per_obj) return;
    uint32_t* base = rows + (size_t)o * n_per_obj * 34;
    uint32_t s = (uint32_t)(o * 7919 + p) * 2654435761u + lane * 40503u;
    for (int t = lane; t < half_m; t += 64) {
        s = s * 1664525u + 1013904223u;
        int q = p + 1 + (int)((s >> 8) % (uint32_t)window);
        if (q >= n_per_obj) q -= n_per_obj;
        uint32_t* r = base + (size_t)q * 34;
        const uint32_t h = s >> 4;
        if (n_atomics > 0) atomicAdd(r + (h % 11u), 1u);
        if (n_atomics > 1) atomicAdd(r + 11 + ((h >> 8) % 11u), 1u);
        if (n_atomics > 2) atomicAdd(r + 22 + ((h >> 16) % 11u), 1u);
        if (n_atomics > 3) atomicAdd(r + 33, 1u);
    }
}

int main(int argc, char** argv) {
    const int n_obj = argc > 1 ? atoi(argv[1]) : 128, n_per = 16384, m = argc > 2 ? atoi(argv[2]) : 900;
    uint32_t* rows;
    const size_t bytes = (size_t)n_obj * n_per * 34 * 4;
    if (hipMalloc(&rows, bytes) != hipSuccess) return 1;
    hipMemset(rows, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int window : {512, 2048, 8192})
        for (int na : {1, 3, 4}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                k_deposit<<<dim3(n_per / 4, n_obj), 256>>>(rows, n_per, m / 2, window, na);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double pairs = (double)n_obj * n_per * (m / 2);
            printf("objects %d M %d window %d atomics/pair %d: %.2f ms  (%.1f G atomics/s)\n", n_obj, m, window, na, best, pairs * na / best * 1e-6);
        }
    return 0;
}
