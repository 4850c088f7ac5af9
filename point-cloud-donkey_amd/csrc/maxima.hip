// maxima.hip — continuous Hough space: mean-shift mode search (one workgroup per (object, class)) + maxima post-processing
// (one workgroup per object).
// Reference seams: Voting::findMaxima (voting/voting.cpp:79-328, 436-462), VotingMeanShift::iFindMaxima and helpers
// (voting/voting_mean_shift.cpp:33-37, 39-177, 201-481), MaximaHandler::averageNeighborMaxima / suppressNeighborMaxima
// (voting/maxima_handler.cpp:51-157).
//
// Latency/LDS-bound and tiny (V <= a few thousand votes per object): all votes of the current class live in LDS,
// seeds come from a 64-bit (z,y,x) cell-key bitonic sort (the reference's std::map order), every wave shifts one
// seed at a time with its 64 lanes striding the votes, and the order-dependent greedy passes (average, suppress)
// run on one lane exactly as the reference's loops do. Votes of a class are taken in slot order.
// Round 3: single-object max types (one density estimate at the cloud centroid), MaxFilterType "Merge", AverageRotation (quaternion
// scatter matrix -> dominant eigenvector). Not built (as in the oracle): RANSAC vote filter, global features.
#include "common.h"
#include <cstring>

namespace {

#define MX_MAXC 256        // classes
#define MX_MAXM 1024       // maxima kept per object before sort/normalise
#define MX_BIAS (1 << 20)
#define MX_MAXM_C 128      // maxima kept per (object, class)
#define MX_REC 16          // pos[3], weight, instance (bits), instance weight, bbox[3], n_votes (bits), bbox quaternion (w,x,y,z), 2 spare

struct MaxArgs {
    const uint32_t* slot_off; const float* vpos; const float* vw; const int32_t* vcls; const int32_t* vinst; const float* vbs;
    int n_classes; const float* class_bw; float bandwidth, threshold; int max_iter, kernel, suppression, min_votes;
    float min_threshold; int best_k, max_maxima, cap; int max_filter; float filter_radius;
    int32_t* n_max; float* mpos; float* mw; int32_t* mcls; int32_t* minst; float* miw; float* mbs; int32_t* mnv; float* class_score;
    float* rec; int32_t* rec_count;      // per (object, class): up to MX_MAXM_C records of MX_REC floats
    unsigned char* work; const uint32_t* work_off; const uint32_t* class_count;   // big objects: per-(object, class) vote arrays in HBM
    uint32_t* truncated;                 // ctx counter: maxima dropped by a cap (MX_MAXM_C per class, MX_MAXM per object); ismhip_sync reports it
    const float* vbq; float* mbq;        // Voting.AverageRotation: bbox quaternions of the votes (w,x,y,z) in, of the maxima out (both or neither)
    int som_type; const float* obj_centroid; const float* obj_radius;   // single-object max types (voting_mean_shift.cpp:124-157)
};
#define MX_LDS_SLOTS 2048  // vote slots per object that fit the LDS-resident kernels
#define MX_WORK_STRIDE 72  // bytes of workspace per vote slot (65 used by k_find_maxima, 37 by k_hough3d)
__device__ __forceinline__ int pow2_cap(uint32_t n) { int c = 64; while ((uint32_t)c < n) c <<= 1; return c; }

// Objects with more than MX_LDS_SLOTS vote slots: the votes of one class no longer fit LDS with their work arrays, so every
// (object, class) workgroup gets a private region of a global workspace instead, sized from its class's vote count:
// k_class_counts tallies the counts, k_work_offsets lays the regions out (offsets in slots, power-of-two sizes, prefix sum).
__global__ __launch_bounds__(256) void k_class_counts(int n_classes, const uint32_t* __restrict__ slot_off, const int32_t* __restrict__ vcls, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_c[MX_MAXC];
    const int o = blockIdx.x;
    for (int c = threadIdx.x; c < n_classes; c += 256) s_c[c] = 0;
    __syncthreads();
    for (uint32_t s = slot_off[o] + threadIdx.x; s < slot_off[o + 1]; s += 256) { const int c = vcls[s]; if (c >= 0 && c < n_classes) atomicAdd(&s_c[c], 1u); }
    __syncthreads();
    for (int c = threadIdx.x; c < n_classes; c += 256) counts[(size_t)o * n_classes + c] = s_c[c];
}
__global__ __launch_bounds__(1024) void k_work_offsets(uint32_t n_oc, const uint32_t* __restrict__ counts, uint32_t* __restrict__ work_off) {
    __shared__ uint32_t s_w[16], s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_oc; c0 += 1024) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < n_oc && counts[i] ? (uint32_t)pow2_cap(counts[i]) : 0u;
        uint32_t inc = v;                                         // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane_id() >= d) inc += t; }
        if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += s_w[w];
        if (i < n_oc) work_off[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_base = off + inc;
        __syncthreads();
    }
}

__device__ __forceinline__ float ms_kernel(int kernel, float u) {          // voting_mean_shift.cpp:378-417
    // the reference evaluates exp(-0.5 * x) in double and rounds to float; expf differs by <= 2 ulp (2e-7 relative), far inside
    // the 1e-4 tolerance on vote weights, and is ~10x cheaper than the double-precision exp on the device
    if (kernel == ISMHIP_KERNEL_GAUSSIAN) return expf(-0.5f * u);
    if (kernel == ISMHIP_KERNEL_UNIFORM) return 1.f;
    return 0.f;
}
__device__ __forceinline__ float ms_neg_kernel_derivative(int kernel, float u) {
    if (kernel == ISMHIP_KERNEL_GAUSSIAN) { const float profile = expf(-0.5f * u); return -(-0.5f * profile); }
    if (kernel == ISMHIP_KERNEL_UNIFORM) return -1.f;
    return 0.f;
}
__device__ __forceinline__ float dist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

// Utils::quatWeightedAverage (utils/utils.cpp:617-665): the "mean" of unit quaternions = dominant eigenvector of the scatter matrix
// sum_k w_k q_k q_k^T. The reference solves it with Eigen::EigenSolver and then looks at eigenvalue 0 only (its loop runs to
// eigenvalues.cols() == 1), i.e. it returns whichever eigenvector the unordered general solver lists first: not reproducible. Here:
// cyclic Jacobi in double on the symmetric 4x4 matrix, the eigenvector of the LARGEST eigenvalue, sign chosen so that its first
// non-zero component (w first) is positive (q and -q are the same rotation). Stated deviation, DESIGN.md §7.
// S = upper triangle row by row: (00 01 02 03 11 12 13 22 23 33).
__device__ inline void quat_from_scatter(const float* S, float* q) {
    double A[4][4] = {{S[0], S[1], S[2], S[3]}, {S[1], S[4], S[5], S[6]}, {S[2], S[5], S[7], S[8]}, {S[3], S[6], S[8], S[9]}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        if (!(off > 1e-30)) break;
        for (int p = 0; p < 3; ++p) for (int r = p + 1; r < 4; ++r) {
            if (fabs(A[p][r]) < 1e-300) continue;
            const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 4; ++k) { const double x = A[k][p], y = A[k][r]; A[k][p] = c * x - sn * y; A[k][r] = sn * x + c * y; }
            for (int k = 0; k < 4; ++k) { const double x = A[p][k], y = A[r][k]; A[p][k] = c * x - sn * y; A[r][k] = sn * x + c * y; }
            for (int k = 0; k < 4; ++k) { const double x = V[k][p], y = V[k][r]; V[k][p] = c * x - sn * y; V[k][r] = sn * x + c * y; }
        }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[best][best]) best = i;
    double v[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
    double sgn = 1.0;
    for (int i = 0; i < 4; ++i) if (v[i] != 0.0) { sgn = v[i] < 0 ? -1.0 : 1.0; break; }
    for (int i = 0; i < 4; ++i) q[i] = (float)(sgn * v[i]);
}
__device__ __forceinline__ void quat_scatter_add(float* S, float w, const float* q) {
    S[0] += w * q[0] * q[0]; S[1] += w * q[0] * q[1]; S[2] += w * q[0] * q[2]; S[3] += w * q[0] * q[3];
    S[4] += w * q[1] * q[1]; S[5] += w * q[1] * q[2]; S[6] += w * q[1] * q[3];
    S[7] += w * q[2] * q[2]; S[8] += w * q[2] * q[3]; S[9] += w * q[3] * q[3];
}

__device__ __forceinline__ float block_sum_f(float v, float* s_red) {   // 256 threads
    v = wave_sum_f(v);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ int block_sum_i(int v, int* s_red) {
    v = wave_sum_i(v);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

template <bool GM>      // GM: the vote arrays live in the global workspace (objects with more than MX_LDS_SLOTS slots)
__global__ __launch_bounds__(256) void k_find_maxima(MaxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // power of two >= max votes of one object (LDS) / >= the votes of this (object, class) (workspace)
    const int cap = GM ? pow2_cap(a.class_count[(size_t)blockIdx.x * a.n_classes + blockIdx.y]) : a.cap;
    unsigned char* arrays = smem;
    if constexpr (GM) arrays = a.work + (size_t)a.work_off[(size_t)blockIdx.x * a.n_classes + blockIdx.y] * MX_WORK_STRIDE;
    float* vx = (float*)arrays;             float* vy = vx + cap; float* vz = vy + cap; float* vw = vz + cap;
    int* vinst = (int*)(vw + cap);          int* vslot = vinst + cap;
    unsigned long long* keys = (unsigned long long*)(vslot + cap);
    float4* ctr = (float4*)(keys + cap);    float4* ctr2 = ctr + cap;
    unsigned char* member = (unsigned char*)(ctr2 + cap);
    __shared__ int s_nmax, s_n, s_ns, s_nc, s_np;
    __shared__ int s_wcnt[4];
    __shared__ float s_redf[4];
    __shared__ int s_redi[4];
    __shared__ int s_bi[4];

    const int o = blockIdx.x;
    const int c = blockIdx.y;                                    // the reference visits the classes of m_votes one by one (voting.cpp:95)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t s0 = a.slot_off[o], s1 = a.slot_off[o + 1];
    const int C = a.n_classes;
    float* rec = a.rec + ((size_t)o * C + c) * MX_MAXM_C * MX_REC;
    if (tid == 0) { s_nmax = 0; a.rec_count[(size_t)o * C + c] = 0; }
    __syncthreads();
    {
        float h = a.class_bw ? a.class_bw[c] : a.bandwidth;                // voting_mean_shift.cpp:48-49
        float h2 = (float)((double)h * (double)h);
        float hh = h * h;
        // ---- ordered compaction of the class's votes into LDS
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (uint32_t base = s0; base < s1; base += 256) {
            const uint32_t s = base + tid;
            const bool f = s < s1 && a.vcls[s] == c;
            const unsigned long long mask = __ballot(f);
            if (lane == 0) s_wcnt[wv] = __popcll(mask);
            __syncthreads();
            int off = s_n;
            for (int k = 0; k < wv; ++k) off += s_wcnt[k];
            if (f) {
                const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cap) {
                    vx[pos] = a.vpos[(size_t)s * 3]; vy[pos] = a.vpos[(size_t)s * 3 + 1]; vz[pos] = a.vpos[(size_t)s * 3 + 2];
                    vw[pos] = a.vw[s]; vinst[pos] = a.vinst[s]; vslot[pos] = (int)s;
                }
            }
            __syncthreads();
            if (tid == 0) s_n += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            __syncthreads();
        }
        const int n = min(s_n, cap);
        if (n == 0) return;                                       // class absent from m_votes (uniform across the block)
        if (a.som_type != ISMHIP_SOM_MEANSHIFT) {
            // Single-object mode with SingleObjectMaxType BANDWIDTH / MODEL_RADIUS / COMPLETE_VOTING_SPACE (voting_mean_shift.cpp:124-157):
            // no mean shift; ONE maximum per class at the centroid of the object's cloud, density and reweighting with the bandwidth
            // of the type: the class's search distance | the farthest cloud point from the centroid (single_object_mode_helper.cpp:15-27)
            // | the farthest vote of the class from the centroid (:29-40, squaredNorm then sqrt)
            const float qx = a.obj_centroid[o * 3], qy = a.obj_centroid[o * 3 + 1], qz = a.obj_centroid[o * 3 + 2];
            if (a.som_type == ISMHIP_SOM_MODEL_RADIUS) h = a.obj_radius[o];
            else if (a.som_type == ISMHIP_SOM_COMPLETE_VOTING_SPACE) {
                float md = 0.f;
                for (int i = tid; i < n; i += 256) { const float dx = vx[i] - qx, dy = vy[i] - qy, dz = vz[i] - qz; md = fmaxf(md, dx * dx + dy * dy + dz * dz); }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) md = fmaxf(md, __shfl_xor(md, off, 64));
                __syncthreads();
                if (lane == 0) s_redf[wv] = md;
                __syncthreads();
                h = sqrtf(fmaxf(fmaxf(s_redf[0], s_redf[1]), fmaxf(s_redf[2], s_redf[3])));
                __syncthreads();
            }
            h2 = (float)((double)h * (double)h); hh = h * h;
            if (tid == 0) { ctr2[0] = make_float4(qx, qy, qz, 0.f); s_np = 1; }
            __syncthreads();
        } else {
        // ---- seeds: unique cells of edge 2h/sqrt(2), key floor(x/edge + 0.5), (z,y,x) order (:431-481)
        const float bin = (h * 2.0f) / sqrtf(2.0f);
        int P = 1; while (P < n) P <<= 1;
        for (int i = tid; i < P; i += 256) {
            unsigned long long key = ~0ull;
            if (i < n) {
                int kx = (int)floor((double)(vx[i] / bin) + 0.5), ky = (int)floor((double)(vy[i] / bin) + 0.5), kz = (int)floor((double)(vz[i] / bin) + 0.5);
                kx = max(-(MX_BIAS - 1), min(MX_BIAS - 1, kx)); ky = max(-(MX_BIAS - 1), min(MX_BIAS - 1, ky)); kz = max(-(MX_BIAS - 1), min(MX_BIAS - 1, kz));
                key = ((unsigned long long)(kz + MX_BIAS) << 42) | ((unsigned long long)(ky + MX_BIAS) << 21) | (unsigned long long)(kx + MX_BIAS);
            }
            keys[i] = key;
        }
        __syncthreads();
        for (int k2 = 2; k2 <= P; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long ka = keys[i], kb = keys[ixj];
                        const bool up = (i & k2) == 0;
                        if ((ka > kb) == up) { keys[i] = kb; keys[ixj] = ka; }
                    }
                }
                __syncthreads();
            }
        if (tid == 0) {
            int ns = 0;
            for (int i = 0; i < n; ++i) if (i == 0 || keys[i] != keys[i - 1]) keys[ns++] = keys[i];
            s_ns = ns;
        }
        __syncthreads();
        const int ns = s_ns;
        // ---- mean shift, one wave per seed (:201-244, 331-376)
        for (int si = wv; si < ns; si += 4) {
            const unsigned long long key = keys[si];
            float cx = (float)((int)(key & 0x1fffff) - MX_BIAS) * bin;
            float cy = (float)((int)((key >> 21) & 0x1fffff) - MX_BIAS) * bin;
            float cz = (float)((int)((key >> 42) & 0x1fffff) - MX_BIAS) * bin;
            int iter = 0; float diff = 0.f; bool skip = false;
            do {
                float sx = 0.f, sy = 0.f, sz = 0.f; double tw = 0.0; int cnt = 0;
                for (int i = lane; i < n; i += 64) {
                    const float d2 = sqdist3(vx[i], vy[i], vz[i], cx, cy, cz);
                    if (d2 < h2) {
                        const float g = ms_neg_kernel_derivative(a.kernel, d2 / hh) * vw[i];
                        sx += g * vx[i]; sy += g * vy[i]; sz += g * vz[i]; tw += (double)g; cnt++;
                    }
                }
                cnt = wave_sum_i(cnt);
                if (cnt == 0) { skip = true; break; }
                sx = wave_sum_f(sx); sy = wave_sum_f(sy); sz = wave_sum_f(sz); tw = wave_sum_d(tw);
                if (tw != 0.0) { const float twf = (float)tw; sx /= twf; sy /= twf; sz /= twf; }
                diff = dist3(cx, cy, cz, sx, sy, sz);
                cx = sx; cy = sy; cz = sz;
                iter++;
            } while (diff > a.threshold && iter <= a.max_iter);
            if (lane == 0) ctr[si] = make_float4(cx, cy, cz, skip ? -1.f : 1.f);
        }
        __syncthreads();
        if (tid == 0) {
            int nc = 0;
            for (int i = 0; i < ns; ++i) if (ctr[i].w > 0.f) ctr[nc++] = ctr[i];
            s_nc = nc;
        }
        __syncthreads();
        int nc = s_nc;
        // ---- densities (:247-285), one wave per centre
        for (int ci = wv; ci < nc; ci += 4) {
            const float4 p = ctr[ci];
            float dsum = 0.f;
            for (int i = lane; i < n; i += 64) {
                const float d2 = sqdist3(vx[i], vy[i], vz[i], p.x, p.y, p.z);
                if (d2 < h2) dsum += ms_kernel(a.kernel, d2 / hh) * vw[i];
            }
            dsum = wave_sum_f(dsum);
            if (lane == 0) ctr[ci].w = dsum;
        }
        __syncthreads();
        if (a.suppression == ISMHIP_SUPPRESS_AVERAGE) {
            // averageNeighborMaxima (maxima_handler.cpp:94-157): greedy, order dependent; duplicates stay in the list
            if (tid == 0) {
                for (int i = 0; i < nc; ++i) member[i] = 0;
                for (int k = 0; k < nc; ++k) {
                    if (member[k]) { ctr2[k] = ctr[k]; continue; }
                    const float4 pa = ctr[k];
                    float ax = 0.f, ay = 0.f, az = 0.f, sd = 0.f; int cnt = 1;
                    ax = pa.x * pa.w; ay = pa.y * pa.w; az = pa.z * pa.w; sd = pa.w;
                    for (int j = k + 1; j < nc; ++j) {
                        if (member[j]) continue;
                        const float4 pb = ctr[j];
                        if (dist3(pa.x, pa.y, pa.z, pb.x, pb.y, pb.z) < h) {
                            member[j] = 1; cnt++;
                            ax += pb.x * pb.w; ay += pb.y * pb.w; az += pb.z * pb.w; sd += pb.w;
                        }
                    }
                    if (cnt == 1) ctr2[k] = pa;
                    else {
                        // the reference accumulates from zero: (0 + c_k*d_k) + c_j*d_j ...; adding to 0 first is exact
                        ctr2[k] = make_float4(ax / sd, ay / sd, az / sd, 0.f);
                    }
                }
            }
            __syncthreads();
            for (int ci = wv; ci < nc; ci += 4) {
                const float4 p = ctr2[ci];
                float dsum = 0.f;
                for (int i = lane; i < n; i += 64) {
                    const float d2 = sqdist3(vx[i], vy[i], vz[i], p.x, p.y, p.z);
                    if (d2 < h2) dsum += ms_kernel(a.kernel, d2 / hh) * vw[i];
                }
                dsum = wave_sum_f(dsum);
                if (lane == 0) ctr[ci] = make_float4(p.x, p.y, p.z, dsum);
            }
            __syncthreads();
        }
        // ---- suppressNeighborMaxima (maxima_handler.cpp:51-92): greedy NMS by density
        if (tid == 0) {
            int np = 0;
            if (a.suppression == ISMHIP_SUPPRESS_AVERAGE || a.suppression == ISMHIP_SUPPRESS_SUPPRESS) {
                float* work = (float*)keys;
                for (int i = 0; i < nc; ++i) work[i] = ctr[i].w;
                for (;;) {
                    float mx = -1.f; int mi = -1;
                    for (int i = 0; i < nc; ++i) if (mi < 0 || work[i] > mx) { mx = work[i]; mi = i; }   // std::max_element: first largest
                    if (mi < 0 || mx == -1.f) break;
                    const float4 cpt = ctr[mi];
                    ctr2[np++] = cpt;
                    work[mi] = -1.f;
                    for (int i = 0; i < nc; ++i)
                        if (dist3(cpt.x, cpt.y, cpt.z, ctr[i].x, ctr[i].y, ctr[i].z) < h) work[i] = -1.f;
                }
            }
            s_np = np;
        }
        __syncthreads();
        }   // mean shift / single-object type
        const int np = s_np;
        // ---- per maximum: density + in-place reweighting (:289-328) and the Voting::findMaxima block (voting.cpp:131-236)
        for (int pi = 0; pi < np; ++pi) {
            const float4 p = ctr2[pi];
            int cnt = 0; float sw = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
            float qs[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < n; i += 256) {
                const float d2 = sqdist3(vx[i], vy[i], vz[i], p.x, p.y, p.z);
                const bool in = d2 < h2;
                member[i] = in ? 1 : 0;
                if (in) {
                    const float w = ms_kernel(a.kernel, d2 / hh) * vw[i];
                    vw[i] = w;
                    cnt++; sw += w;
                    if (a.vbs) { const size_t s = (size_t)vslot[i] * 3; b0 += w * a.vbs[s]; b1 += w * a.vbs[s + 1]; b2 += w * a.vbs[s + 2]; }
                    if (a.vbq) quat_scatter_add(qs, w, a.vbq + (size_t)vslot[i] * 4);
                }
            }
            cnt = block_sum_i(cnt, s_redi);
            if (cnt < a.min_votes || cnt == 0) continue;      // uniform across the block
            sw = block_sum_f(sw, s_redf);
            b0 = block_sum_f(b0, s_redf); b1 = block_sum_f(b1, s_redf); b2 = block_sum_f(b2, s_redf);
            if (a.vbq) {
#pragma unroll
                for (int e = 0; e < 10; ++e) qs[e] = block_sum_f(qs[e], s_redf);
            }
            __syncthreads();
            // instance id with the largest summed weight; ties -> smallest id; weights <= 0 never win (voting.cpp:139-165).
            // Per-instance sums in an LDS hash table (open addressing, keys = instance ids, values = 2^-32 fixed point updated with
            // ds_add_u64: O(n), order independent) instead of the O(n^2) pairwise tally.
            int* hkey = (int*)ctr;                                   // ctr / keys are free once the final positions sit in ctr2
            unsigned long long* hval = keys;
            const int HEMPTY = (int)0x80000000;
            for (int i = tid; i < cap; i += 256) { hkey[i] = HEMPTY; hval[i] = 0ull; }
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                if (!member[i]) continue;
                const int id = vinst[i];
                const float wv_ = vw[i];
                const unsigned long long fx = wv_ > 0.f ? (unsigned long long)((double)wv_ * 4294967296.0) : 0ull;
                unsigned slot = ((unsigned)id * 2654435761u) & (unsigned)(cap - 1);
                for (int probe = 0; probe < cap; ++probe) {
                    const int old = atomicCAS(&hkey[slot], HEMPTY, id);
                    if (old == HEMPTY || old == id) { atomicAdd(&hval[slot], fx); break; }
                    slot = (slot + 1) & (unsigned)(cap - 1);
                }
            }
            __syncthreads();
            unsigned long long bS = 0ull; int bI = 0x7fffffff;
            for (int i = tid; i < cap; i += 256) {
                const int id = hkey[i];
                if (id == HEMPTY) continue;
                const unsigned long long S = hval[i];
                if (S > bS || (S == bS && S > 0ull && (unsigned)id < (unsigned)bI)) { bS = S; bI = id; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long oS = __shfl_xor(bS, off, 64); const int oI = __shfl_xor(bI, off, 64);
                if (oS > bS || (oS == bS && oS > 0ull && (unsigned)oI < (unsigned)bI)) { bS = oS; bI = oI; }
            }
            float bestS = (float)((double)bS * 2.3283064365386963e-10); int bestI = bI;
            __shared__ unsigned long long s_bS[4];
            if (lane == 0) { s_bS[wv] = bS; s_bi[wv] = bI; }
            __syncthreads();
            if (tid == 0) {
                unsigned long long fS = s_bS[0]; bestI = s_bi[0];
                for (int k = 1; k < 4; ++k)
                    if (s_bS[k] > fS || (s_bS[k] == fS && fS > 0ull && (unsigned)s_bi[k] < (unsigned)bestI)) { fS = s_bS[k]; bestI = s_bi[k]; }
                bestS = (float)((double)fS * 2.3283064365386963e-10);
                const int m = s_nmax;
                if (m < MX_MAXM_C) {
                    float* r = rec + (size_t)m * MX_REC;
                    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = sw;
                    r[4] = __int_as_float(bestS > 0.f ? bestI : -1); r[5] = bestS > 0.f ? bestS : 0.f;
                    r[6] = b0 / sw; r[7] = b1 / sw; r[8] = b2 / sw; r[9] = __int_as_float(cnt);
                    r[10] = 1.f; r[11] = 0.f; r[12] = 0.f; r[13] = 0.f;
                    if (a.vbq) quat_from_scatter(qs, r + 10);          // voting.cpp:210-215
                    s_nmax = m + 1;
                } else atomicAdd(a.truncated, 1u);
            }
            __syncthreads();
        }
        __syncthreads();
    }
    if (tid == 0) a.rec_count[(size_t)o * C + c] = s_nmax;
}

// Voting::findMaxima tail (voting.cpp:272, 298-323, 441-462): gather the maxima of all classes (class order, then the order
// in which iFindMaxima produced them), stable sort by weight, normalise, MinThreshold, BestK, class scores.
__global__ __launch_bounds__(64) void k_finalize_maxima(MaxArgs a) {
    __shared__ float s_w[MX_MAXM], s_iw[MX_MAXM];
    __shared__ int s_src[MX_MAXM], s_cls[MX_MAXM], s_order[MX_MAXM];
    __shared__ float s_tw[MX_MAXM], s_tiw[MX_MAXM];                 // scratch of the inter-class filter
    __shared__ int s_tsrc[MX_MAXM], s_tcls[MX_MAXM], s_iw2[MX_MAXM];
    const int o = blockIdx.x;
    const int C = a.n_classes;
    if (threadIdx.x != 0) return;
    int nm = 0;
    {
        int total = 0;
        for (int c = 0; c < C; ++c) total += a.rec_count[(size_t)o * C + c];
        if (total > MX_MAXM && a.truncated) atomicAdd(a.truncated, 1u);          // the object keeps its first MX_MAXM maxima: reported by ismhip_sync
    }
    for (int c = 0; c < C; ++c) {
        const int cnt = a.rec_count[(size_t)o * C + c];
        for (int m = 0; m < cnt && nm < MX_MAXM; ++m) {
            const float* r = a.rec + (((size_t)o * C + c) * MX_MAXM_C + m) * MX_REC;
            s_w[nm] = r[3]; s_iw[nm] = r[5]; s_src[nm] = c * MX_MAXM_C + m; s_cls[nm] = c; s_order[nm] = nm; ++nm;
        }

    }
    if (a.max_filter == ISMHIP_MAXFILTER_SIMPLE && nm > 1) {
        // MaximaHandler::filterMaxima "Simple" -> suppressNeighborMaxima2 (maxima_handler.cpp:227-268): greedy non-maximum suppression
        // over the maxima of ALL classes: take the heaviest one left (std::max_element: the first largest), drop everything closer
        // than the search radius (itself included), repeat. Runs on the raw weights, before sorting / normalising (voting.cpp:262-272).
        // s_order doubles as the work list here: >= 0 = still a candidate, -1 = gone; the survivors are compacted in selection order.
        int kept = 0;
        for (;;) {
            int mi = -1;
            for (int i = 0; i < nm; ++i) if (s_order[i] >= 0 && (mi < 0 || s_w[i] > s_w[mi])) mi = i;
            if (mi < 0) break;
            const float* rm = a.rec + ((size_t)o * C * MX_MAXM_C + s_src[mi]) * MX_REC;
            const float cx = rm[0], cy = rm[1], cz = rm[2];
            for (int i = 0; i < nm; ++i) {
                if (s_order[i] < 0) continue;
                const float* ri = a.rec + ((size_t)o * C * MX_MAXM_C + s_src[i]) * MX_REC;
                if (i == mi || dist3(cx, cy, cz, ri[0], ri[1], ri[2]) < a.filter_radius) s_order[i] = -1;
            }
            s_iw2[kept++] = mi;
        }
        // move the survivors to the front (selection order = descending weight; the stable sort below keeps it)
        for (int r = 0; r < kept; ++r) { const int i = s_iw2[r]; s_tw[r] = s_w[i]; s_tiw[r] = s_iw[i]; s_tsrc[r] = s_src[i]; s_tcls[r] = s_cls[i]; }
        for (int r = 0; r < kept; ++r) { s_w[r] = s_tw[r]; s_iw[r] = s_tiw[r]; s_src[r] = s_tsrc[r]; s_cls[r] = s_tcls[r]; s_order[r] = r; }
        nm = kept;
    }
    if (a.max_filter == ISMHIP_MAXFILTER_MERGE && nm > 1) {
        // MaximaHandler::filterMaxima "Merge" -> mergeAndFilterMaxima(maxima, true) (maxima_handler.cpp:300-387) with mergeMaxima
        // (:390-440), on the raw weights and in the reference's list order (classes ascending, then the order iFindMaxima produced).
        // For every not yet consumed maximum i: the later maxima closer than i's search distance whose own search distance is not
        // larger are consumed; if there are any, they and i are grouped by class (ascending), every group is merged (running weighted
        // means of position and box size, weights added, instance weights tallied per instance id, quaternion average of the running
        // result and the member), and the heaviest merged maximum (the first one among equals) replaces the whole neighbourhood.
        float* recs = a.rec + (size_t)o * C * MX_MAXM_C * MX_REC;
        int kept = 0;
        for (int i = 0; i < nm; ++i) s_order[i] = 1;                      // 1 = not consumed ("dirty" is 0)
        for (int i = 0; i < nm; ++i) {
            if (!s_order[i]) continue;
            const float* ri = recs + (size_t)s_src[i] * MX_REC;
            const float sd = a.class_bw ? a.class_bw[s_cls[i]] : a.bandwidth;
            int nclose = 0;                                               // members in s_tsrc[] (list order: the consumed ones, then i)
            for (int j = i + 1; j < nm; ++j) {
                if (!s_order[j]) continue;
                const float* rj = recs + (size_t)s_src[j] * MX_REC;
                const float osd = a.class_bw ? a.class_bw[s_cls[j]] : a.bandwidth;
                if (dist3(rj[0], rj[1], rj[2], ri[0], ri[1], ri[2]) < sd && osd <= sd) { s_tsrc[nclose++] = j; s_order[j] = 0; }
            }
            if (nclose == 0) { s_iw2[kept++] = i; continue; }
            s_tsrc[nclose++] = i;
            float best[MX_REC]; float best_w = 0.f; int best_cls = -1; bool have = false;     // VotingMaximum(): weight 0
            for (int cc = 0; cc < C; ++cc) {                              // std::map<unsigned, ...>: ascending class id
                float m[MX_REC]; bool any = false;
                float mw = 0.f, px = 0.f, py = 0.f, pz = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f; int nv = 0;
                float mq[4] = {1.f, 0.f, 0.f, 0.f};
                int inst_id[64]; float inst_w[64]; int n_inst = 0;       // instance tally of the group (more than 64 distinct ids: the rest is ignored)
                int bi = -1;
                for (int t = 0; t < nclose; ++t) {
                    const int x = s_tsrc[t];
                    if (s_cls[x] != cc) continue;
                    const float* rx = recs + (size_t)s_src[x] * MX_REC;
                    const float w = s_w[x];
                    any = true;
                    px = (px * mw + rx[0] * w) / (mw + w); py = (py * mw + rx[1] * w) / (mw + w); pz = (pz * mw + rx[2] * w) / (mw + w);
                    b0 = (b0 * mw + rx[6] * w) / (mw + w); b1 = (b1 * mw + rx[7] * w) / (mw + w); b2 = (b2 * mw + rx[8] * w) / (mw + w);
                    { float S[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; quat_scatter_add(S, mw, mq); quat_scatter_add(S, w, rx + 10); quat_from_scatter(S, mq); }
                    mw += w; nv += __float_as_int(rx[9]);
                    const int iid = __float_as_int(rx[4]);
                    int k = 0; for (; k < n_inst; ++k) if (inst_id[k] == iid) break;
                    if (k == n_inst && n_inst < 64) { inst_id[n_inst] = iid; inst_w[n_inst] = 0.f; ++n_inst; }
                    if (k < n_inst) inst_w[k] += s_iw[x];
                    bi = -1;                                              // largest tally, lowest id among equals (std::map order), tallies <= 0 never win
                    for (int k2 = 0; k2 < n_inst; ++k2) {
                        const float w2 = inst_w[k2];
                        if (w2 > 0.f && (bi < 0 || w2 > inst_w[bi] || (w2 == inst_w[bi] && (unsigned)inst_id[k2] < (unsigned)inst_id[bi]))) bi = k2;
                    }
                }
                if (!any) continue;
                m[0] = px; m[1] = py; m[2] = pz; m[3] = mw; m[4] = __int_as_float(bi >= 0 ? inst_id[bi] : -1); m[5] = bi >= 0 ? inst_w[bi] : 0.f;
                m[6] = b0; m[7] = b1; m[8] = b2; m[9] = __int_as_float(nv); m[10] = mq[0]; m[11] = mq[1]; m[12] = mq[2]; m[13] = mq[3]; m[14] = 0.f; m[15] = 0.f;
                if (mw > best_w) { for (int e = 0; e < MX_REC; ++e) best[e] = m[e]; best_w = mw; best_cls = cc; have = true; }
            }
            if (have) {
                float* wr = recs + (size_t)s_src[i] * MX_REC;             // the merged maximum takes i's record
                for (int e = 0; e < MX_REC; ++e) wr[e] = best[e];
                s_w[i] = best_w; s_iw[i] = best[5]; s_cls[i] = best_cls;
                s_iw2[kept++] = i;
            }
        }
        for (int r = 0; r < kept; ++r) { const int i = s_iw2[r]; s_tw[r] = s_w[i]; s_tiw[r] = s_iw[i]; s_tsrc[r] = s_src[i]; s_tcls[r] = s_cls[i]; }
        for (int r = 0; r < kept; ++r) { s_w[r] = s_tw[r]; s_iw[r] = s_tiw[r]; s_src[r] = s_tsrc[r]; s_cls[r] = s_tcls[r]; }
        nm = kept;
    }
    for (int i = 0; i < nm; ++i) s_order[i] = i;
    for (int i = 1; i < nm; ++i) {
        const int v = s_order[i]; int j = i - 1;
        while (j >= 0 && s_w[s_order[j]] < s_w[v]) { s_order[j + 1] = s_order[j]; --j; }
        s_order[j + 1] = v;
    }
    float sum = 0.f, sum_inst = 0.f;
    for (int i = 0; i < nm; ++i) { sum += s_w[s_order[i]]; sum_inst += s_iw[s_order[i]]; }
    for (int i = 0; i < nm; ++i) {
        s_w[i] = sum != 0.f ? s_w[i] / sum : 0.f;
        s_iw[i] = sum_inst != 0.f ? s_iw[i] / sum_inst : 0.f;
    }
    float thr = a.min_threshold;
    if (thr < 0.f) { const float mxw = nm > 0 ? s_w[s_order[0]] : 0.0f; thr = -thr * mxw; }
    int kept = 0;
    for (int i = 0; i < nm; ++i) if (s_w[s_order[i]] >= thr) s_order[kept++] = s_order[i];
    if (a.best_k > 0 && kept >= a.best_k) kept = a.best_k;
    for (int cc = 0; cc < C; ++cc) a.class_score[(size_t)o * C + cc] = 0.f;
    for (int i = 0; i < kept; ++i) {
        float* cs = &a.class_score[(size_t)o * C + s_cls[s_order[i]]];
        if (s_w[s_order[i]] > *cs) *cs = s_w[s_order[i]];
    }
    const int nout = min(kept, a.max_maxima);
    a.n_max[o] = nout;
    for (int i = 0; i < a.max_maxima; ++i) {
        const size_t t = (size_t)o * a.max_maxima + i;
        const bool ok = i < nout; const int m = ok ? s_order[i] : 0;
        const float* r = a.rec + ((size_t)o * C * MX_MAXM_C + (ok ? s_src[m] : 0)) * MX_REC;
        a.mpos[t * 3] = ok ? r[0] : 0.f; a.mpos[t * 3 + 1] = ok ? r[1] : 0.f; a.mpos[t * 3 + 2] = ok ? r[2] : 0.f;
        a.mw[t] = ok ? s_w[m] : 0.f; a.mcls[t] = ok ? s_cls[m] : -1; a.minst[t] = ok ? __float_as_int(r[4]) : -1;
        a.miw[t] = ok ? s_iw[m] : 0.f; a.mnv[t] = ok ? __float_as_int(r[9]) : 0;
        if (a.mbs) { a.mbs[t * 3] = ok ? r[6] : 0.f; a.mbs[t * 3 + 1] = ok ? r[7] : 0.f; a.mbs[t * 3 + 2] = ok ? r[8] : 0.f; }
        if (a.mbq) { a.mbq[t * 4] = ok ? r[10] : 1.f; a.mbq[t * 4 + 1] = ok ? r[11] : 0.f; a.mbq[t * 4 + 2] = ok ? r[12] : 0.f; a.mbq[t * 4 + 3] = ok ? r[13] : 0.f; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Discrete Hough space: VotingHough3D::iFindMaxima (voting/voting_hough_3d.cpp:33-95) over pcl::recognition::HoughSpace3D
// (EXTERNAL, restated from SURVEY Appendix A.7 exactly as the oracle restates it).
// One workgroup per (object, class) owns that class's whole accumulator, so the scatter is PRIVATE to the workgroup: the bins
// the votes can reach (their bounding box in bin coordinates, +1 for the interpolation neighbours) are cut into tiles that fit
// LDS, every tile is zeroed, filled with ds_add_u64 (2^-40 fixed point: the sum does not depend on the order of the votes,
// where the reference's double accumulator does) and scanned for maxima in place. Nothing is shared between workgroups, so no
// global atomics are needed; a vote set whose bounding box does not fit one tile simply takes more passes over the (few
// hundred) votes, and one extra pass up front finds max(H) for the relative threshold.
// ---------------------------------------------------------------------------------------------------------------------------
#define HG_FIX 1099511627776.0            /* 2^40 */
struct HoughArgs {
    const float* vbq;                    // Voting.AverageRotation: bbox quaternions of the votes, or NULL
    const uint32_t* slot_off; const float* vpos; const float* vw; const int32_t* vcls; const int32_t* vinst; const float* vbs;
    int n_classes; const float* class_bin; float bin; float minc[3], maxc[3]; int use_int; float rel; int min_votes, cap, tile_edge;
    float* rec; int32_t* rec_count; int32_t* overflow;
    unsigned char* work; const uint32_t* work_off; const uint32_t* class_count;
    uint32_t* truncated;
};
struct HgBin { int c[3]; int dir[3]; float wc[3]; bool in; };
__device__ __forceinline__ HgBin hg_bin(const HoughArgs& a, double bin, const int cnt[3], float px, float py, float pz) {
    HgBin b; b.in = true;
    const float p[3] = {px, py, pz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double rel = (double)p[d] - (double)a.minc[d];
        const double fl = floor(rel / bin);
        if (!(fl >= 0.0 && fl < (double)cnt[d])) { b.in = false; b.c[d] = 0; b.dir[d] = 0; b.wc[d] = 0.f; continue; }
        b.c[d] = (int)fl;
        const float centre = (float)((2.0 * (double)b.c[d] * bin + bin) / 2.0);
        const double diff = rel - (double)centre;
        b.wc[d] = (float)(1.0 - fabs(diff) / bin);
        b.dir[d] = diff < 0 ? -1 : 1;
    }
    return b;
}

template <bool GM>
__global__ __launch_bounds__(256) void k_hough3d(HoughArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cap = GM ? pow2_cap(a.class_count[(size_t)blockIdx.x * a.n_classes + blockIdx.y]) : a.cap;
    unsigned char* arrays = smem;
    if constexpr (GM) arrays = a.work + (size_t)a.work_off[(size_t)blockIdx.x * a.n_classes + blockIdx.y] * MX_WORK_STRIDE;
    float* vx = (float*)arrays;             float* vy = vx + cap; float* vz = vy + cap; float* vw = vz + cap;
    int* vinst = (int*)(vw + cap);          int* vslot = vinst + cap;
    unsigned long long* hval = (unsigned long long*)(vslot + cap);      // instance tally (values)
    int* hkey = (int*)(hval + cap);                                       // instance tally (keys)
    unsigned char* member = (unsigned char*)(hkey + cap);
    unsigned long long* tile = GM ? (unsigned long long*)smem : (unsigned long long*)(member + cap);     // tile_edge^3 bins, always in LDS
    __shared__ int s_n, s_nmax, s_lo[3], s_hi[3], s_nm;
    __shared__ int s_wcnt[4];
    __shared__ float s_redf[4];
    __shared__ int s_redi[4];
    __shared__ int s_bi[4];
    __shared__ unsigned long long s_bS[4], s_hmax;
    __shared__ long long s_mbin[MX_MAXM_C];
    const int o = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t s0 = a.slot_off[o], s1 = a.slot_off[o + 1];
    const int C = a.n_classes;
    float* rec = a.rec + ((size_t)o * C + c) * MX_MAXM_C * MX_REC;
    if (tid == 0) { s_nmax = 0; s_n = 0; s_nm = 0; s_hmax = 0ull; a.rec_count[(size_t)o * C + c] = 0;
                    for (int d = 0; d < 3; ++d) { s_lo[d] = 0x7fffffff; s_hi[d] = -1; } }
    __syncthreads();
    // ---- ordered compaction of the class's votes into LDS (as k_find_maxima)
    for (uint32_t base = s0; base < s1; base += 256) {
        const uint32_t s = base + tid;
        const bool f = s < s1 && a.vcls[s] == c;
        const unsigned long long mask = __ballot(f);
        if (lane == 0) s_wcnt[wv] = __popcll(mask);
        __syncthreads();
        int off = s_n;
        for (int k = 0; k < wv; ++k) off += s_wcnt[k];
        if (f) {
            const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < cap) {
                vx[pos] = a.vpos[(size_t)s * 3]; vy[pos] = a.vpos[(size_t)s * 3 + 1]; vz[pos] = a.vpos[(size_t)s * 3 + 2];
                vw[pos] = a.vw[s]; vinst[pos] = a.vinst[s]; vslot[pos] = (int)s;
            }
        }
        __syncthreads();
        if (tid == 0) s_n += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n = min(s_n, cap);
    if (n == 0) return;                                       // class absent from m_votes (uniform across the block)
    const double bin = (double)(a.class_bin ? a.class_bin[c] : a.bin);
    int cnt[3];
    for (int d = 0; d < 3; ++d) {
        const double q = ceil(((double)a.maxc[d] - (double)a.minc[d]) / bin);
        cnt[d] = q > 0.0 ? (q < 2.0e9 ? (int)q : 2000000000) : 0;
    }
    if (cnt[0] == 0 || cnt[1] == 0 || cnt[2] == 0) return;
    // ---- bounding box of the reachable bins
    for (int i = tid; i < n; i += 256) {
        const HgBin b = hg_bin(a, bin, cnt, vx[i], vy[i], vz[i]);
        if (!b.in) continue;
        for (int d = 0; d < 3; ++d) { atomicMin(&s_lo[d], max(0, b.c[d] - 1)); atomicMax(&s_hi[d], min(cnt[d] - 1, b.c[d] + 1)); }
    }
    __syncthreads();
    if (s_hi[0] < 0) return;                                  // every vote fell outside the space
    const int E = a.tile_edge, EI = E - 2;                    // tile = interior EI^3 + a halo of one bin on every side
    const int lo[3] = {s_lo[0], s_lo[1], s_lo[2]}, hi[3] = {s_hi[0], s_hi[1], s_hi[2]};
    const int nt[3] = {(hi[0] - lo[0]) / EI + 1, (hi[1] - lo[1]) / EI + 1, (hi[2] - lo[2]) / EI + 1};
    const int n_tiles = nt[0] * nt[1] * nt[2];
    const int passes = a.rel > 0.f || true ? 2 : 1;          // pass 0: max(H); pass 1: maxima
    unsigned long long thr_fix = 0ull;
    for (int pass = 0; pass < passes; ++pass) {
        for (int t = 0; t < n_tiles; ++t) {
            // tile t covers interior bins [t0, t0 + EI) per axis, stored at local (b - t0 + 1)
            const int t0[3] = {lo[0] + (t % nt[0]) * EI, lo[1] + ((t / nt[0]) % nt[1]) * EI, lo[2] + (t / (nt[0] * nt[1])) * EI};
            for (int i = tid; i < E * E * E; i += 256) tile[i] = 0ull;
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                const HgBin b = hg_bin(a, bin, cnt, vx[i], vy[i], vz[i]);
                if (!b.in) continue;
                const double w = (double)vw[i];
                const int nn = a.use_int ? 8 : 1;
                for (int q = 0; q < nn; ++q) {
                    float iw = 1.0f; bool ok = true; int l[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const int side = (q >> d) & 1;                 // 0: central bin, 1: the neighbour on the vote's side
                        const int bb = b.c[d] + (side ? b.dir[d] : 0);
                        if (bb < 0 || bb >= cnt[d]) ok = false;
                        if (a.use_int) iw *= side ? 1.0f - b.wc[d] : b.wc[d];
                        l[d] = bb - t0[d] + 1;
                        if (l[d] < 0 || l[d] >= E) ok = false;
                    }
                    if (!ok || !(iw > 0.0f)) continue;
                    const double contrib = a.use_int ? w * (double)iw : w;
                    atomicAdd(&tile[(l[2] * E + l[1]) * E + l[0]], (unsigned long long)(contrib * HG_FIX + 0.5));
                }
            }
            __syncthreads();
            if (pass == 0) {
                unsigned long long m = 0ull;
                for (int i = tid; i < E * E * E; i += 256) m = tile[i] > m ? tile[i] : m;
                m = wave_min_u64(~m); m = ~m;                             // wave max through the min helper
                if (lane == 0) atomicMax(&s_hmax, m);
            } else {
                // interior bins >= threshold with no strictly greater 26-neighbour (halo bins are complete: every vote was scanned)
                for (int i = tid; i < EI * EI * EI; i += 256) {
                    const int lx = i % EI + 1, ly = (i / EI) % EI + 1, lz = i / (EI * EI) + 1;
                    const int gx = t0[0] + lx - 1, gy = t0[1] + ly - 1, gz = t0[2] + lz - 1;
                    if (gx > hi[0] || gy > hi[1] || gz > hi[2]) continue;
                    const unsigned long long v = tile[(lz * E + ly) * E + lx];
                    if (v < thr_fix || v == 0ull) continue;               // H = 0 bins have no voters: the reference drops them later (voting.cpp:131)
                    bool is_max = true;
                    for (int q = 0; q < 27 && is_max; ++q) {
                        if (q == 13) continue;
                        const int nx = lx + q % 3 - 1, ny = ly + (q / 3) % 3 - 1, nz = lz + q / 9 - 1;
                        if (tile[(nz * E + ny) * E + nx] > v) is_max = false;
                    }
                    if (!is_max) continue;
                    const int m = atomicAdd(&s_nm, 1);
                    if (m < MX_MAXM_C) s_mbin[m] = (long long)gx + (long long)cnt[0] * ((long long)gy + (long long)cnt[1] * (long long)gz);
                }
            }
            __syncthreads();
        }
        if (pass == 0) {
            // findMaxima(-rel): threshold = rel * max(H), or max(H) itself when rel > 1; in fixed point, rounded down so that the
            // bin holding the maximum always passes
            const double hmax = (double)s_hmax / HG_FIX, rel = (double)a.rel;
            const double thr = rel <= 1.0 ? rel * hmax : hmax;
            thr_fix = (unsigned long long)(thr * HG_FIX);
        }
    }
    int nm = s_nm;
    if (nm > MX_MAXM_C) { if (tid == 0) { atomicAdd(a.overflow, 1); atomicAdd(a.truncated, 1u); } nm = MX_MAXM_C; }
    if (tid == 0)                                               // ascending bin index = the order findMaxima reports them in
        for (int i = 1; i < nm; ++i) { const long long v = s_mbin[i]; int j = i - 1; while (j >= 0 && s_mbin[j] > v) { s_mbin[j + 1] = s_mbin[j]; --j; } s_mbin[j + 1] = v; }
    __syncthreads();
    // ---- per maximum: voters, weighted centre (voting_hough_3d.cpp:70-93) and the Voting::findMaxima block (voting.cpp:131-236)
    for (int pi = 0; pi < nm; ++pi) {
        const long long mb = s_mbin[pi];
        const int mbx = (int)(mb % cnt[0]), mby = (int)((mb / cnt[0]) % cnt[1]), mbz = (int)(mb / ((long long)cnt[0] * cnt[1]));
        int vcnt = 0; float sw = 0.f, px = 0.f, py = 0.f, pz = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        float qs[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < n; i += 256) {
            const HgBin b = hg_bin(a, bin, cnt, vx[i], vy[i], vz[i]);
            bool in = b.in;
            const int mbc[3] = {mbx, mby, mbz};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (mbc[d] == b.c[d]) { if (a.use_int && !(b.wc[d] > 0.f)) in = false; }
                else if (a.use_int && mbc[d] == b.c[d] + b.dir[d]) { if (!(1.0f - b.wc[d] > 0.f)) in = false; }
                else in = false;
            }
            if (in && a.use_int) {                                        // the product of the three factors must itself be > 0
                float iw = 1.0f;
#pragma unroll
                for (int d = 0; d < 3; ++d) iw *= mbc[d] == b.c[d] ? b.wc[d] : 1.0f - b.wc[d];
                in = iw > 0.0f;
            }
            member[i] = in ? 1 : 0;
            if (in) {
                const float w = vw[i];
                vcnt++; sw += w; px += vx[i] * w; py += vy[i] * w; pz += vz[i] * w;
                if (a.vbs) { const size_t s = (size_t)vslot[i] * 3; b0 += w * a.vbs[s]; b1 += w * a.vbs[s + 1]; b2 += w * a.vbs[s + 2]; }
                if (a.vbq) quat_scatter_add(qs, w, a.vbq + (size_t)vslot[i] * 4);
            }
        }
        vcnt = block_sum_i(vcnt, s_redi);
        if (vcnt < a.min_votes || vcnt == 0) continue;        // uniform across the block
        sw = block_sum_f(sw, s_redf);
        px = block_sum_f(px, s_redf); py = block_sum_f(py, s_redf); pz = block_sum_f(pz, s_redf);
        b0 = block_sum_f(b0, s_redf); b1 = block_sum_f(b1, s_redf); b2 = block_sum_f(b2, s_redf);
        if (a.vbq) {
#pragma unroll
            for (int e = 0; e < 10; ++e) qs[e] = block_sum_f(qs[e], s_redf);
        }
        __syncthreads();
        // instance tally: as k_find_maxima (LDS hash, 2^-32 fixed point, ties -> smallest id, weights <= 0 never win)
        const int HEMPTY = (int)0x80000000;
        for (int i = tid; i < cap; i += 256) { hkey[i] = HEMPTY; hval[i] = 0ull; }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            if (!member[i]) continue;
            const int id = vinst[i];
            const float wv_ = vw[i];
            const unsigned long long fx = wv_ > 0.f ? (unsigned long long)((double)wv_ * 4294967296.0) : 0ull;
            unsigned slot = ((unsigned)id * 2654435761u) & (unsigned)(cap - 1);
            for (int probe = 0; probe < cap; ++probe) {
                const int old = atomicCAS(&hkey[slot], HEMPTY, id);
                if (old == HEMPTY || old == id) { atomicAdd(&hval[slot], fx); break; }
                slot = (slot + 1) & (unsigned)(cap - 1);
            }
        }
        __syncthreads();
        unsigned long long bS = 0ull; int bI = 0x7fffffff;
        for (int i = tid; i < cap; i += 256) {
            const int id = hkey[i];
            if (id == HEMPTY) continue;
            const unsigned long long S = hval[i];
            if (S > bS || (S == bS && S > 0ull && (unsigned)id < (unsigned)bI)) { bS = S; bI = id; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long oS = __shfl_xor(bS, off, 64); const int oI = __shfl_xor(bI, off, 64);
            if (oS > bS || (oS == bS && oS > 0ull && (unsigned)oI < (unsigned)bI)) { bS = oS; bI = oI; }
        }
        if (lane == 0) { s_bS[wv] = bS; s_bi[wv] = bI; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long fS = s_bS[0]; int bestI = s_bi[0];
            for (int k = 1; k < 4; ++k)
                if (s_bS[k] > fS || (s_bS[k] == fS && fS > 0ull && (unsigned)s_bi[k] < (unsigned)bestI)) { fS = s_bS[k]; bestI = s_bi[k]; }
            const float bestS = (float)((double)fS * 2.3283064365386963e-10);
            const int m = s_nmax;
            if (m < MX_MAXM_C) {
                float* r = rec + (size_t)m * MX_REC;
                r[0] = px / sw; r[1] = py / sw; r[2] = pz / sw; r[3] = sw;
                r[4] = __int_as_float(bestS > 0.f ? bestI : -1); r[5] = bestS > 0.f ? bestS : 0.f;
                r[6] = b0 / sw; r[7] = b1 / sw; r[8] = b2 / sw; r[9] = __int_as_float(vcnt);
                r[10] = 1.f; r[11] = 0.f; r[12] = 0.f; r[13] = 0.f;
                if (a.vbq) quat_from_scatter(qs, r + 10);
                s_nmax = m + 1;
            } else atomicAdd(a.truncated, 1u);
        }
        __syncthreads();
    }
    if (tid == 0) a.rec_count[(size_t)o * C + c] = s_nmax;
}

}  // namespace

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

// workspace of the big-object kernels: counts[n_oc] | offsets[n_oc] | regions (<= 2 * slots + nothing for absent classes)
static int big_object_workspace(ismhip_ctx* ctx, int n_obj, int n_classes, const uint32_t* slot_offsets_h, const uint32_t* slot_off_d, const int32_t* vote_class,
                                unsigned char** work, const uint32_t** work_off, const uint32_t** class_count) {
    const size_t n_oc = (size_t)n_obj * n_classes;
    const size_t total = slot_offsets_h[n_obj];
    const size_t region_slots = 2 * total + 64 * std::min(n_oc, total);          // pow2_cap(n) <= max(64, 2 n - 1), absent classes take nothing
    if (region_slots >= (1ull << 32)) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "find_maxima: more than 2^31 vote slots in one call not built");
    const size_t head = (2 * n_oc * sizeof(uint32_t) + 255) / 256 * 256;
    unsigned char* w = (unsigned char*)ism_scratch(ctx, SCR_MAX_WORK, head + region_slots * MX_WORK_STRIDE);
    if (!w) return ISMHIP_ERR_NOMEM;
    uint32_t* counts = (uint32_t*)w; uint32_t* offs = counts + n_oc;
    hipLaunchKernelGGL(k_class_counts, dim3(n_obj), dim3(256), 0, ctx->stream, n_classes, slot_off_d, vote_class, counts);
    ISM_CHECK_LAUNCH(ctx, "k_class_counts");
    hipLaunchKernelGGL(k_work_offsets, dim3(1), dim3(1024), 0, ctx->stream, (uint32_t)n_oc, counts, offs);
    ISM_CHECK_LAUNCH(ctx, "k_work_offsets");
    *work = w + head; *work_off = offs; *class_count = counts;
    return ISMHIP_OK;
}

extern "C" int ismhip_find_maxima(ismhip_ctx* ctx, int n_obj, const uint32_t* slot_offsets_h,
                                  const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                                  const int32_t* vote_instance, const float* vote_bbox_size,
                                  const ismhip_maxima_params* P,
                                  int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                                  int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                                  float* max_bbox_size_out, int32_t* max_n_votes_out, float* class_score_out) {
    if (!ctx || n_obj <= 0 || !slot_offsets_h || !vote_pos || !vote_weight || !vote_class || !vote_instance || !P ||
        !n_maxima_out || !max_pos_out || !max_weight_out || !max_class_out || !max_instance_out || !max_instance_weight_out ||
        !max_n_votes_out || !class_score_out || P->n_classes <= 0 || P->max_maxima <= 0 || !(P->bandwidth > 0.f || P->class_bandwidth_h))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: bad argument");
    if (P->n_classes > MX_MAXC) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "find_maxima: more than 256 classes not built");
    if (P->kernel != ISMHIP_KERNEL_GAUSSIAN && P->kernel != ISMHIP_KERNEL_UNIFORM) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: kernel");
    uint32_t max_slots = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (slot_offsets_h[o + 1] < slot_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: offsets not monotone");
        max_slots = std::max(max_slots, slot_offsets_h[o + 1] - slot_offsets_h[o]);
    }
    int cap = 64; while ((uint32_t)cap < max_slots) cap <<= 1;
    const bool big = cap > MX_LDS_SLOTS;
    const size_t dyn = big ? 0 : (size_t)cap * (4 * 4 + 2 * 4 + 8 + 16 + 16 + 1);
    uint32_t* so = ism_upload_offsets(ctx, SCR_SLOT_OFF, slot_offsets_h, n_obj + 1);
    if (!so) return ISMHIP_ERR_HIP;
    float* bw = nullptr;
    if (P->class_bandwidth_h) {
        bw = (float*)ism_scratch(ctx, SCR_CLASS_BW, (size_t)P->n_classes * 4);
        if (!bw) return ISMHIP_ERR_NOMEM;
        ISM_HIP(ctx, hipMemcpyAsync(bw, P->class_bandwidth_h, (size_t)P->n_classes * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    MaxArgs a;
    a.slot_off = so; a.vpos = vote_pos; a.vw = vote_weight; a.vcls = vote_class; a.vinst = vote_instance; a.vbs = vote_bbox_size;
    a.n_classes = P->n_classes; a.class_bw = bw; a.bandwidth = P->bandwidth; a.threshold = P->threshold; a.max_iter = P->max_iter;
    a.kernel = P->kernel; a.suppression = P->suppression; a.min_votes = P->min_votes_threshold; a.min_threshold = P->min_threshold;
    a.best_k = P->best_k; a.max_maxima = P->max_maxima; a.cap = cap;
    if (P->max_filter != ISMHIP_MAXFILTER_NONE && P->max_filter != ISMHIP_MAXFILTER_SIMPLE && P->max_filter != ISMHIP_MAXFILTER_MERGE)
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: MaxFilterType");
    a.max_filter = P->max_filter; a.filter_radius = P->bandwidth;      // MaximaHandler::m_radius = the configured bandwidth (voting_mean_shift.cpp:46)
    if ((P->vote_bbox_quat == nullptr) != (P->max_bbox_quat_out == nullptr)) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: vote_bbox_quat and max_bbox_quat_out go together");
    a.vbq = P->vote_bbox_quat; a.mbq = P->max_bbox_quat_out;
    a.som_type = P->single_object_max_type; a.obj_centroid = P->object_centroid; a.obj_radius = P->object_radius;
    if (a.som_type < ISMHIP_SOM_MEANSHIFT || a.som_type > ISMHIP_SOM_COMPLETE_VOTING_SPACE) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: single_object_max_type");
    if (a.som_type != ISMHIP_SOM_MEANSHIFT && (!a.obj_centroid || (a.som_type == ISMHIP_SOM_MODEL_RADIUS && !a.obj_radius)))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "find_maxima: single-object max types need object_centroid (and object_radius for MODEL_RADIUS)");
    a.n_max = n_maxima_out; a.mpos = max_pos_out; a.mw = max_weight_out; a.mcls = max_class_out; a.minst = max_instance_out;
    a.miw = max_instance_weight_out; a.mbs = max_bbox_size_out; a.mnv = max_n_votes_out; a.class_score = class_score_out;
    if (!ctx->attr_done.count((const void*)k_find_maxima<false>)) {      // the attribute is per device: remembered per ctx, not per process
        ISM_HIP(ctx, hipFuncSetAttribute((const void*)k_find_maxima<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        ctx->attr_done.insert((const void*)k_find_maxima<false>);
    }
    a.truncated = ctx->truncated_d;
    a.work = nullptr; a.work_off = nullptr; a.class_count = nullptr;
    if (big) { int rc = big_object_workspace(ctx, n_obj, P->n_classes, slot_offsets_h, so, vote_class, &a.work, &a.work_off, &a.class_count); if (rc != ISMHIP_OK) return rc; }
    const size_t n_oc = (size_t)n_obj * P->n_classes;
    a.rec = (float*)ism_scratch(ctx, SCR_MAX_REC, n_oc * MX_MAXM_C * MX_REC * sizeof(float) + n_oc * sizeof(int32_t));
    if (!a.rec) return ISMHIP_ERR_NOMEM;
    a.rec_count = (int32_t*)(a.rec + n_oc * MX_MAXM_C * MX_REC);
    TimerScope ts(ctx, "maxima");
    if (big) hipLaunchKernelGGL(k_find_maxima<true>, dim3(n_obj, P->n_classes), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_find_maxima<false>, dim3(n_obj, P->n_classes), dim3(256), dyn, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_find_maxima");
    hipLaunchKernelGGL(k_finalize_maxima, dim3(n_obj), dim3(64), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_finalize_maxima");
    return ISMHIP_OK;
}


extern "C" int ismhip_hough3d_maxima(ismhip_ctx* ctx, int n_obj, const uint32_t* slot_offsets_h,
                                     const float* vote_pos, const float* vote_weight, const int32_t* vote_class,
                                     const int32_t* vote_instance, const float* vote_bbox_size,
                                     const ismhip_hough_params* P,
                                     int32_t* n_maxima_out, float* max_pos_out, float* max_weight_out,
                                     int32_t* max_class_out, int32_t* max_instance_out, float* max_instance_weight_out,
                                     float* max_bbox_size_out, int32_t* max_n_votes_out, float* class_score_out) {
    if (!ctx || n_obj <= 0 || !slot_offsets_h || !vote_pos || !vote_weight || !vote_class || !vote_instance || !P ||
        !n_maxima_out || !max_pos_out || !max_weight_out || !max_class_out || !max_instance_out || !max_instance_weight_out ||
        !max_n_votes_out || !class_score_out || P->n_classes <= 0 || P->max_maxima <= 0 || !(P->bin_size > 0.f || P->class_bin_h))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: bad argument");
    if (P->n_classes > MX_MAXC) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "hough3d_maxima: more than 256 classes not built");
    for (int d = 0; d < 3; ++d) if (!(P->max_coord[d] > P->min_coord[d])) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: MaxCoord must exceed MinCoord");
    if (P->class_bin_h) for (int c = 0; c < P->n_classes; ++c) if (!(P->class_bin_h[c] > 0.f)) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: bin size");
    uint32_t max_slots = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (slot_offsets_h[o + 1] < slot_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: offsets not monotone");
        max_slots = std::max(max_slots, slot_offsets_h[o + 1] - slot_offsets_h[o]);
    }
    int cap = 64; while ((uint32_t)cap < max_slots) cap <<= 1;
    const bool big = cap > MX_LDS_SLOTS;
    // LDS: 37 B per vote slot; the rest of ~150 KB holds the accumulator tile (edge 16 .. 24 bins including the halo)
    const size_t vote_bytes = big ? 0 : (size_t)cap * (4 * 4 + 2 * 4 + 8 + 4 + 1);
    int edge = 24;
    while (edge > 8 && ((vote_bytes + 15) / 16 * 16 + (size_t)edge * edge * edge * 8) > 150 * 1024) --edge;
    const size_t dyn = (vote_bytes + 15) / 16 * 16 + (size_t)edge * edge * edge * 8 + 16;
    uint32_t* so = ism_upload_offsets(ctx, SCR_SLOT_OFF, slot_offsets_h, n_obj + 1);
    if (!so) return ISMHIP_ERR_HIP;
    float* cbin = nullptr;
    if (P->class_bin_h) {
        cbin = (float*)ism_scratch(ctx, SCR_CLASS_BW, (size_t)P->n_classes * 4);
        if (!cbin) return ISMHIP_ERR_NOMEM;
        ISM_HIP(ctx, hipMemcpyAsync(cbin, P->class_bin_h, (size_t)P->n_classes * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t n_oc = (size_t)n_obj * P->n_classes;
    float* rec = (float*)ism_scratch(ctx, SCR_MAX_REC, n_oc * MX_MAXM_C * MX_REC * sizeof(float) + (n_oc + 1) * sizeof(int32_t));
    if (!rec) return ISMHIP_ERR_NOMEM;
    HoughArgs h;
    h.slot_off = so; h.vpos = vote_pos; h.vw = vote_weight; h.vcls = vote_class; h.vinst = vote_instance; h.vbs = vote_bbox_size;
    h.n_classes = P->n_classes; h.class_bin = cbin; h.bin = P->bin_size;
    for (int d = 0; d < 3; ++d) { h.minc[d] = P->min_coord[d]; h.maxc[d] = P->max_coord[d]; }
    h.use_int = P->use_interpolation ? 1 : 0; h.rel = P->rel_threshold; h.min_votes = P->min_votes_threshold; h.cap = cap; h.tile_edge = edge;
    h.rec = rec; h.rec_count = (int32_t*)(rec + n_oc * MX_MAXM_C * MX_REC); h.overflow = h.rec_count + n_oc;
    MaxArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_classes = P->n_classes; a.min_threshold = P->min_threshold; a.best_k = P->best_k; a.max_maxima = P->max_maxima;
    if (P->max_filter != ISMHIP_MAXFILTER_NONE && P->max_filter != ISMHIP_MAXFILTER_SIMPLE && P->max_filter != ISMHIP_MAXFILTER_MERGE)
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: MaxFilterType");
    if (P->max_filter == ISMHIP_MAXFILTER_MERGE && P->class_bin_h) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "hough3d_maxima: MaxFilterType Merge with per-class bin sizes not built");
    a.max_filter = P->max_filter; a.filter_radius = P->bin_size / 2;   // MaximaHandler::setRadius(BinSize[0] / 2) (voting_hough_3d.cpp:45)
    a.bandwidth = P->bin_size / 2;                                     // getSearchDistForClass of the Merge filter (class_bw stays NULL)
    if ((P->vote_bbox_quat == nullptr) != (P->max_bbox_quat_out == nullptr)) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "hough3d_maxima: vote_bbox_quat and max_bbox_quat_out go together");
    h.vbq = P->vote_bbox_quat; a.mbq = P->max_bbox_quat_out;
    a.n_max = n_maxima_out; a.mpos = max_pos_out; a.mw = max_weight_out; a.mcls = max_class_out; a.minst = max_instance_out;
    a.miw = max_instance_weight_out; a.mbs = max_bbox_size_out; a.mnv = max_n_votes_out; a.class_score = class_score_out;
    a.rec = h.rec; a.rec_count = h.rec_count;
    const void* hk = big ? (const void*)k_hough3d<true> : (const void*)k_hough3d<false>;
    if (!ctx->attr_done.count(hk)) {
        ISM_HIP(ctx, hipFuncSetAttribute(hk, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        ctx->attr_done.insert(hk);
    }
    h.work = nullptr; h.work_off = nullptr; h.class_count = nullptr; h.truncated = ctx->truncated_d; a.truncated = ctx->truncated_d;
    if (big) { int rc = big_object_workspace(ctx, n_obj, P->n_classes, slot_offsets_h, so, vote_class, &h.work, &h.work_off, &h.class_count); if (rc != ISMHIP_OK) return rc; }
    TimerScope ts(ctx, "hough3d");
    ISM_HIP(ctx, hipMemsetAsync(h.overflow, 0, 4, ctx->stream));
    if (big) hipLaunchKernelGGL(k_hough3d<true>, dim3(n_obj, P->n_classes), dim3(256), dyn, ctx->stream, h);
    else hipLaunchKernelGGL(k_hough3d<false>, dim3(n_obj, P->n_classes), dim3(256), dyn, ctx->stream, h);
    ISM_CHECK_LAUNCH(ctx, "k_hough3d");
    hipLaunchKernelGGL(k_finalize_maxima, dim3(n_obj), dim3(64), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_finalize_maxima");
    return ISMHIP_OK;
}
