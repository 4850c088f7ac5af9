// kmeans.hip — k-means codebook clustering on the device.
// Reference seam: ClusteringKMeans::cluster<DistanceType> (clustering/clustering_kmeans.h:53-131) with its callers KMeansCount /
// KMeansFactor / KMeansThumbRule / KMeansHartigan: flann::hierarchicalClustering with branching == the desired cluster count (so the
// cut through the tree is the root's children: ONE level of Lloyd k-means), then a 1-NN search of every feature in the centres.
// FLANN 1.9.1 is not in /root/reference (EXTERNAL, restated from knowledge of kmeans_index.h / center_chooser.h): centre choosers
// RANDOM / GONZALES / KMEANSPP (one local try), Lloyd loop [means of the members -> reassign, strict '>' so ties go to the lowest
// centre -> hand a point of a populated cluster to every empty one], at most `Iterations` rounds, stop when nothing moved.
// Build-defined where FLANN is not reproducible or not exact (parity unpinned, see DESIGN):
//   * randomness: FLANN draws from rand(); here splitmix64(seed + draw counter), first centre = draw % n
//   * k-means++ sampling: FLANN walks the float potentials sequentially; here the potentials are turned into integers
//     (floor(d_i / dmax0 * 2^40), dmax0 = the largest distance to the first centre) so that the draw r < sum picks the first index
//     whose running integer sum exceeds r whatever the summation order
//   * means: FLANN adds the members in double in index order; here every element is rounded to a multiple of 2^-40 of the data's
//     largest power of two and added as an integer (order-free), mean = (sum / scale) / count rounded to float; the centres are
//     float (FLANN keeps them double inside the loop), distances are the float functor the activation uses (bit-exact, exact search)
//   * the final assignment is the exact 1-NN (FLANN: an approximate tree search with 32 checks)
// Every distance goes through wave_functor / ismhip_knn, so a CPU restatement with the same integers reproduces the result bit by bit.
#include "common.h"
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

int ism_knn_only_codebook(ismhip_ctx* ctx, int n_words, int dim, const float* words_d, ismhip_codebook** out, bool light);   // train.hip

namespace {

#include "functor.h"

__host__ __device__ inline unsigned long long km_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
#define KM_INVALID 0xffffffffu
#define KM_POT_BITS 40

// closest[i] = min(closest[i], d(point i, centre chosen[step - 1])): one wave per point, the functor's own summation order.
// slot[step] collects max over i of (closest bits << 32 | ~i): the farthest point, lowest index on ties (GONZALES; step 1 also
// gives dmax0 for the k-means++ integers).
__global__ __launch_bounds__(256) void k_km_closest(int n, int dim, int metric, const float* __restrict__ desc, const uint32_t* __restrict__ chosen, int step,
                                                    float* __restrict__ closest, unsigned long long* __restrict__ slot) {
    __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
    __shared__ unsigned long long s_best[4];
    const uint32_t c = chosen[step - 1];
    if (c == KM_INVALID) return;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    unsigned long long best = 0ull;
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        const float d = wave_functor(metric, desc + (size_t)i * dim, desc + (size_t)c * dim, dim, lane, s_terms[wv]);
        float cur = closest[i];
        if (d < cur) { cur = d; if (lane == 0) closest[i] = d; }
        if (cur == cur) {                                             // NaN distances never lead
            const unsigned long long key = ((unsigned long long)__float_as_uint(cur) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
            best = key > best ? key : best;
        }
    }
    if (lane == 0) s_best[wv] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s_best[0];
        for (int w = 1; w < 4; ++w) b = s_best[w] > b ? s_best[w] : b;
        if (b) atomicMax(&slot[step], b);
    }
}
__device__ __forceinline__ unsigned long long km_pot(float d, float dmax0) {
    if (!(d > 0.f) || !(dmax0 > 0.f)) return 0ull;
    const double q = (double)d / (double)dmax0 * (double)(1ull << KM_POT_BITS);
    return (unsigned long long)q;
}
// k-means++: integer potentials summed per block of 1024 points
__global__ __launch_bounds__(256) void k_km_block_sums(int n, const float* __restrict__ closest, const unsigned long long* __restrict__ slot,
                                                       const uint32_t* __restrict__ chosen, int step, unsigned long long* __restrict__ bsum) {
    __shared__ unsigned long long s_p[4];
    if (chosen[step - 1] == KM_INVALID) return;
    const float dmax0 = __uint_as_float((uint32_t)(slot[1] >> 32));
    unsigned long long s = 0ull;
    for (int t = 0; t < 4; ++t) { const int i = blockIdx.x * 1024 + t * 256 + threadIdx.x; if (i < n) s += km_pot(closest[i], dmax0); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane_id() == 0) s_p[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_p[0] + s_p[1] + s_p[2] + s_p[3];
}
// one workgroup: total potential T, draw r = floor(rng * T / 2^64), first index whose running sum exceeds r
__global__ __launch_bounds__(1024) void k_km_select(int n, int nb, const unsigned long long* __restrict__ bsum, const float* __restrict__ closest,
                                                    const unsigned long long* __restrict__ slot, unsigned long long seed, int step, uint32_t* __restrict__ chosen) {
    __shared__ unsigned long long s_w[16], s_total, s_r, s_base;
    __shared__ int s_blk;
    __shared__ uint32_t s_pick;
    if (chosen[step - 1] == KM_INVALID) { if (threadIdx.x == 0) chosen[step] = KM_INVALID; return; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long part = 0ull;
    for (int b = tid; b < nb; b += 1024) part += bsum[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_w[wv] = part;
    __syncthreads();
    if (tid == 0) {
        unsigned long long T = 0ull;
        for (int w = 0; w < 16; ++w) T += s_w[w];
        s_total = T;
        s_r = T ? __umul64hi(km_splitmix64(seed + (unsigned long long)step), T) : 0ull;
        // the block holding r (sequential over the block sums: nb <= n / 1024)
        unsigned long long run = 0ull; int blk = nb - 1;
        for (int b = 0; b < nb; ++b) { if (run + bsum[b] > s_r) { blk = b; break; } run += bsum[b]; }
        s_blk = blk; s_base = run; s_pick = KM_INVALID;
    }
    __syncthreads();
    if (s_total == 0ull) { if (tid == 0) chosen[step] = KM_INVALID; return; }       // every point coincides with a chosen centre
    const float dmax0 = __uint_as_float((uint32_t)(slot[1] >> 32));
    const int i = s_blk * 1024 + tid;
    const unsigned long long q = i < n ? km_pot(closest[i], dmax0) : 0ull;
    unsigned long long inc = q;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    unsigned long long off = s_base;
    for (int w = 0; w < wv; ++w) off += s_w[w];
    if (q && off + inc > s_r && off + inc - q <= s_r) atomicMin(&s_pick, (uint32_t)i);
    __syncthreads();
    if (tid == 0) chosen[step] = s_pick;
}
__global__ void k_km_select_farthest(const unsigned long long* __restrict__ slot, int step, uint32_t* __restrict__ chosen) {
    if (chosen[step - 1] == KM_INVALID) { chosen[step] = KM_INVALID; return; }
    const unsigned long long key = slot[step];
    const float d = __uint_as_float((uint32_t)(key >> 32));
    chosen[step] = d > 0.f ? 0xffffffffu - (uint32_t)(key & 0xffffffffull) : KM_INVALID;          // FLANN: only a distance > 0 is taken
}
// RANDOM: the next entry of the seeded permutation that does not coincide with a chosen centre (distance >= 1e-16)
__global__ __launch_bounds__(64) void k_km_select_random(int n, const uint32_t* __restrict__ perm, uint32_t* __restrict__ cursor, const float* __restrict__ closest,
                                                         int step, uint32_t* __restrict__ chosen) {
    if (chosen[step - 1] == KM_INVALID) { if (threadIdx.x == 0) chosen[step] = KM_INVALID; return; }
    uint32_t c = *cursor;
    const int lane = threadIdx.x;
    uint32_t pick = KM_INVALID;
    while (c < (uint32_t)n) {
        const uint32_t t = c + lane;
        const bool ok = t < (uint32_t)n && !(closest[perm[t]] < 1e-16f);
        const unsigned long long m = __ballot(ok);
        if (m) { const int first = __ffsll((long long)m) - 1; pick = perm[c + first]; c += first + 1; break; }
        c += 64;
    }
    if (lane == 0) { chosen[step] = pick; *cursor = c; }
}
__global__ void k_km_count_chosen(int k, const uint32_t* __restrict__ chosen, uint32_t* __restrict__ n_chosen) {
    uint32_t c = 0;
    while (c < (uint32_t)k && chosen[c] != KM_INVALID) ++c;
    *n_chosen = c;
}
__global__ void k_km_gather(int k, int dim, const uint32_t* __restrict__ chosen, const float* __restrict__ desc, float* __restrict__ centers) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)k * dim) return;
    centers[t] = desc[(size_t)chosen[t / dim] * dim + t % dim];
}
__global__ void k_km_absmax(size_t tot, const float* __restrict__ desc, uint32_t* __restrict__ out_bits) {
    uint32_t m = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(desc[t]) & 0x7fffffffu;
        if (b <= 0x7f800000u) m = b > m ? b : m;                      // NaN payloads do not count
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(m, o, 64); m = x > m ? x : m; }
    if (lane_id() == 0 && m) atomicMax(out_bits, m);
}
// integer sums of the members of every cluster; counts
__global__ void k_km_accumulate(int n, int dim, const float* __restrict__ desc, const int32_t* __restrict__ belongs, double scale,
                                long long* __restrict__ sums, uint32_t* __restrict__ counts) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * dim) return;
    const int i = (int)(t / dim), j = (int)(t % dim);
    const int c = belongs[i];
    if (c < 0) return;
    const double v = (double)desc[t] * scale;
    if (v == v) atomicAdd((unsigned long long*)&sums[(size_t)c * dim + j], (unsigned long long)(long long)llrint(v));
    if (j == 0) atomicAdd(&counts[c], 1u);
}
__global__ void k_km_means(int k, int dim, const long long* __restrict__ sums, const uint32_t* __restrict__ counts, double scale, float* __restrict__ centers) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)k * dim) return;
    const uint32_t cnt = counts[t / dim];
    if (cnt) centers[t] = (float)(((double)sums[t] / scale) / (double)cnt);
}
__global__ void k_km_reassign(int n, const int32_t* __restrict__ nearest, int32_t* __restrict__ belongs, uint32_t* __restrict__ counts, uint32_t* __restrict__ changed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nw = nearest[i], old = belongs[i];
    if (nw >= 0 && nw != old) {
        if (old >= 0) atomicSub(&counts[old], 1u);
        atomicAdd(&counts[nw], 1u);
        belongs[i] = nw;
        atomicOr(changed, 1u);
    }
}
// FLANN: an empty cluster i takes the FIRST point of the next cluster (cyclically) that has more than one; sequential over i
__global__ __launch_bounds__(1024) void k_km_fix_empty(int n, int k, int32_t* __restrict__ belongs, uint32_t* __restrict__ counts, uint32_t* __restrict__ changed) {
    __shared__ int s_j;
    __shared__ uint32_t s_min;
    for (int i = 0; i < k; ++i) {
        if (threadIdx.x == 0) {
            s_j = -1; s_min = 0xffffffffu;
            if (counts[i] == 0) {
                int j = (i + 1) % k;
                for (int t = 0; t < k && counts[j] <= 1; ++t) j = (j + 1) % k;
                if (counts[j] > 1) s_j = j;
            }
        }
        __syncthreads();
        const int j = s_j;
        if (j >= 0) {
            for (int p = threadIdx.x; p < n; p += 1024) if (belongs[p] == j) { atomicMin(&s_min, (uint32_t)p); break; }
            __syncthreads();
            if (threadIdx.x == 0) { belongs[s_min] = i; counts[j] -= 1; counts[i] += 1; *changed = 1u; }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int ismhip_kmeans(ismhip_ctx* ctx, int metric, int n, int dim, const float* desc, int n_clusters, int max_iterations,
                             int centers_init, unsigned long long seed, float* centers_out, int32_t* assign_out, float* dist_out,
                             int32_t* n_clusters_out, int32_t* iterations_out) {
    if (!ctx || n <= 0 || dim <= 0 || !desc || n_clusters <= 0 || max_iterations < 0 || !centers_out || !assign_out || !n_clusters_out ||
        (metric != ISMHIP_METRIC_L2SQ && metric != ISMHIP_METRIC_CHI2) ||
        (centers_init != ISMHIP_CENTERS_RANDOM && centers_init != ISMHIP_CENTERS_GONZALES && centers_init != ISMHIP_CENTERS_KMEANSPP))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "kmeans: bad argument");
    if (dim > 1344) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "kmeans: descriptor longer than 1344 not built");
    if (n_clusters > n) n_clusters = n;                               // clustering_kmeans.h:69-73
    hipStream_t st = ctx->stream;
    const int k = n_clusters;
    const int nb = (n + 1023) / 1024;
    // scratch: chosen[k] | n_chosen, cursor, changed, absmax | slot[k + 1] | bsum[nb] | closest[n] | belongs[n] | nearest[n] | ndist[n] | counts[k] | sums[k * dim] | perm[n]
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_chosen = take((size_t)k * 4), o_misc = take(64), o_slot = take(((size_t)k + 1) * 8), o_bsum = take((size_t)nb * 8), o_closest = take((size_t)n * 4),
                 o_belongs = take((size_t)n * 4), o_nearest = take((size_t)n * 4), o_ndist = take((size_t)n * 4), o_counts = take((size_t)k * 4),
                 o_sums = take((size_t)k * dim * 8), o_perm = take((size_t)n * 4);
    char* base = (char*)ism_scratch(ctx, SCR_KMEANS, off);
    if (!base) return ISMHIP_ERR_NOMEM;
    uint32_t* chosen = (uint32_t*)(base + o_chosen); uint32_t* misc = (uint32_t*)(base + o_misc);
    unsigned long long* slot = (unsigned long long*)(base + o_slot); unsigned long long* bsum = (unsigned long long*)(base + o_bsum);
    float* closest = (float*)(base + o_closest); int32_t* belongs = (int32_t*)(base + o_belongs); int32_t* nearest = (int32_t*)(base + o_nearest);
    float* ndist = (float*)(base + o_ndist); uint32_t* counts = (uint32_t*)(base + o_counts); long long* sums = (long long*)(base + o_sums);
    uint32_t* perm = (uint32_t*)(base + o_perm);
    uint32_t* n_chosen_d = misc; uint32_t* cursor = misc + 1; uint32_t* changed = misc + 2; uint32_t* absmax = misc + 3;
    TimerScope ts(ctx, "kmeans");
    // ---- centres
    ISM_HIP(ctx, hipMemsetAsync(misc, 0, 64, st));
    ISM_HIP(ctx, hipMemsetAsync(slot, 0, ((size_t)k + 1) * 8, st));
    ISM_HIP(ctx, hipMemsetAsync(closest, 0x7f, (size_t)n * 4, st));   // 0x7f7f7f7f = 3.39e38: above every finite distance of interest
    uint32_t first;
    if (centers_init == ISMHIP_CENTERS_RANDOM) {
        std::vector<uint32_t> p((size_t)n); std::iota(p.begin(), p.end(), 0u);
        for (int i = n - 1; i > 0; --i) {                              // Fisher-Yates, draw i uses splitmix64(seed + (n - i))
            const uint32_t j = (uint32_t)(km_splitmix64(seed + (unsigned long long)(n - i)) % (unsigned long long)(i + 1));
            std::swap(p[(size_t)i], p[j]);
        }
        ISM_HIP(ctx, hipMemcpyAsync(perm, p.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        ISM_HIP(ctx, hipStreamSynchronize(st));                        // p leaves scope
        first = p[0];
        static const uint32_t one = 1u;
        ISM_HIP(ctx, hipMemcpyAsync(cursor, &one, 4, hipMemcpyHostToDevice, st));
    } else {
        first = (uint32_t)(km_splitmix64(seed) % (unsigned long long)n);
    }
    ISM_HIP(ctx, hipMemcpyAsync(chosen, &first, 4, hipMemcpyHostToDevice, st));
    for (int step = 1; step < k; ++step) {
        hipLaunchKernelGGL(k_km_closest, dim3(std::min(2048, (n + 3) / 4)), dim3(256), 0, st, n, dim, metric, desc, chosen, step, closest, slot);
        if (centers_init == ISMHIP_CENTERS_KMEANSPP) {
            hipLaunchKernelGGL(k_km_block_sums, dim3(nb), dim3(256), 0, st, n, closest, slot, chosen, step, bsum);
            hipLaunchKernelGGL(k_km_select, dim3(1), dim3(1024), 0, st, n, nb, bsum, closest, slot, seed, step, chosen);
        } else if (centers_init == ISMHIP_CENTERS_GONZALES) {
            hipLaunchKernelGGL(k_km_select_farthest, dim3(1), dim3(1), 0, st, slot, step, chosen);
        } else {
            hipLaunchKernelGGL(k_km_select_random, dim3(1), dim3(64), 0, st, n, perm, cursor, closest, step, chosen);
        }
    }
    ISM_CHECK_LAUNCH(ctx, "kmeans centre choosers");
    hipLaunchKernelGGL(k_km_count_chosen, dim3(1), dim3(1), 0, st, k, chosen, n_chosen_d);
    hipLaunchKernelGGL(k_km_absmax, dim3(1024), dim3(256), 0, st, (size_t)n * dim, desc, absmax);
    uint32_t hv[4] = {0, 0, 0, 0};
    ISM_HIP(ctx, hipMemcpyAsync(hv, misc, 16, hipMemcpyDeviceToHost, st));
    ISM_HIP(ctx, hipStreamSynchronize(st));
    const int kc = (int)hv[0];                                          // centres found (fewer than asked when the points run out)
    if (kc <= 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "kmeans: no centre could be chosen");
    float amax; std::memcpy(&amax, &hv[3], 4);
    int e = 0;
    if (amax > 0.f && std::isfinite(amax)) std::frexp(amax, &e);        // amax < 2^e
    const double scale = std::ldexp(1.0, KM_POT_BITS - e);
    hipLaunchKernelGGL(k_km_gather, dim3((unsigned)(((size_t)kc * dim + 255) / 256)), dim3(256), 0, st, kc, dim, chosen, desc, centers_out);
    ISM_CHECK_LAUNCH(ctx, "k_km_gather");
    // ---- Lloyd
    auto assign = [&](int32_t* idx, float* dist) -> int {
        ismhip_codebook* cb = nullptr;
        int rc = ism_knn_only_codebook(ctx, kc, dim, centers_out, &cb, true);      // rebuilt every iteration: no stage-1 image
        if (rc != ISMHIP_OK) return rc;
        rc = ismhip_knn(ctx, cb, metric, n, desc, 1, idx, dist);
        if (rc == ISMHIP_OK && hipStreamSynchronize(st) != hipSuccess) rc = ism_set_err(ctx, ISMHIP_ERR_HIP, "kmeans: sync");
        ismhip_codebook_destroy(ctx, cb);
        return rc;
    };
    int rc = assign(belongs, ndist);
    if (rc != ISMHIP_OK) return rc;
    int it = 0;
    bool converged = false;
    while (!converged && it < max_iterations) {
        ++it;
        ISM_HIP(ctx, hipMemsetAsync(sums, 0, (size_t)kc * dim * 8, st));
        ISM_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)kc * 4, st));
        ISM_HIP(ctx, hipMemsetAsync(changed, 0, 4, st));
        hipLaunchKernelGGL(k_km_accumulate, dim3((unsigned)(((size_t)n * dim + 255) / 256)), dim3(256), 0, st, n, dim, desc, belongs, scale, sums, counts);
        hipLaunchKernelGGL(k_km_means, dim3((unsigned)(((size_t)kc * dim + 255) / 256)), dim3(256), 0, st, kc, dim, sums, counts, scale, centers_out);
        ISM_CHECK_LAUNCH(ctx, "k_km_means");
        rc = assign(nearest, ndist);
        if (rc != ISMHIP_OK) return rc;
        hipLaunchKernelGGL(k_km_reassign, dim3((n + 255) / 256), dim3(256), 0, st, n, nearest, belongs, counts, changed);
        if (n > kc) hipLaunchKernelGGL(k_km_fix_empty, dim3(1), dim3(1024), 0, st, n, kc, belongs, counts, changed);
        ISM_CHECK_LAUNCH(ctx, "k_km_reassign");
        uint32_t ch = 0;
        ISM_HIP(ctx, hipMemcpyAsync(&ch, changed, 4, hipMemcpyDeviceToHost, st));
        ISM_HIP(ctx, hipStreamSynchronize(st));
        converged = ch == 0;
    }
    // ---- the reference's own pass: nearest centre of every feature (clustering_kmeans.h:109-127), exact here
    rc = assign(assign_out, dist_out ? dist_out : ndist);
    if (rc != ISMHIP_OK) return rc;
    *n_clusters_out = kc;
    if (iterations_out) *iterations_out = it;
    return ISMHIP_OK;
}
