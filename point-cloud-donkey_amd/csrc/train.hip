// train.hip — training-time Codebook::activate on the device.
// Reference: codebook/codebook.cpp:64-368 (activate: kNN activation of every training feature, class sigma^2, K = 1 clean-up,
// statistical class weights term1 * term2 * term3), codeword_distribution.cpp:37-71 (addCodeword: vote = rotateInto(centre -
// keypoint, LRF)), :169-243 (computeWeights: median of exp(-d^2 / 0.25) over the activating features), with one codeword per
// training feature (clustering_none.cpp:25-35). The self-kNN is ismhip_knn on a codebook of all features; everything after it
// (sigma, vote CSR, weights) is built here from the device-resident activation list:
//   k_tr_count / k_tr_flags / k_tr_scan / k_tr_members / k_tr_votes : deterministic CSR of the distributions (votes of a word
//                in activation order = ascending (feature, j): the caller passes the features class-major as the reference
//                iterates them), K = 1 clean-up, vote = quaternion rotateInto
//   k_tr_sigma_dist + k_tr_sigma : the functor values of the (feature, codeword) sample by one wave each (bit-exact functor
//                order), then ONE thread per class adds them sequentially in float, as the reference's loops do
//   k_tr_weights : wave per vote, median by bitwise bisection over the float bits
//   k_tr_stats1/2/3 : votes per class, words per class, sum[word], m_term3 (keyed by class only: the LAST word that holds the
//                class wins, :325-339 — a 64-bit atomicMax over (word, value) reproduces that), class weight per vote
#include "common.h"
#include <cmath>
#include <cstring>
#include <numeric>

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {

#include "functor.h"
#include "quat.h"

__global__ void k_tr_count(int na, const int32_t* __restrict__ act, uint32_t* __restrict__ cnt, uint32_t* __restrict__ rank) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const int w = act[a];
    if (w >= 0) rank[a] = atomicAdd(&cnt[w], 1u);
}
// keep[w] = word survives; kcnt[w] = its votes (0 when dropped)
__global__ void k_tr_flags(int n, int clean_up, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ keep, uint32_t* __restrict__ kcnt) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint32_t c = cnt[w];
    const bool k = c > 0 && (!clean_up || c == 1);
    keep[w] = k ? 1u : 0u; kcnt[w] = k ? c : 0u;
}
// exclusive scans of up to three arrays of n entries by ONE workgroup (n <= a few million: training is not the hot path);
// out[n] = total
__global__ __launch_bounds__(1024) void k_tr_scan(int n, const uint32_t* a0, uint32_t* o0, const uint32_t* a1, uint32_t* o1, const uint32_t* a2, uint32_t* o2) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t* in[3] = {a0, a1, a2}; uint32_t* out[3] = {o0, o1, o2};
    for (int arr = 0; arr < 3; ++arr) {
        if (!in[arr]) continue;
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            const int i = base + threadIdx.x;
            const uint32_t v = i < n ? in[arr][i] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            const int w = threadIdx.x >> 6;
            if (lane_id() == 63) s_wave[w] = incl;
            __syncthreads();
            uint32_t wave_off = 0;
            for (int k = 0; k < w; ++k) wave_off += s_wave[k];
            const uint32_t carry = s_carry;
            if (i < n) out[arr][i] = carry + wave_off + incl - v;
            __syncthreads();
            if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) out[arr][n] = s_carry;
        __syncthreads();
    }
}
__global__ void k_tr_members(int na, const int32_t* __restrict__ act, const uint32_t* __restrict__ all_off, const uint32_t* __restrict__ rank,
                             uint32_t* __restrict__ members) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const int w = act[a];
    if (w >= 0) members[all_off[w] + rank[a]] = (uint32_t)a;
}
// one thread per activation: its stable position inside the word's distribution, the vote and the word list
__global__ void k_tr_votes(int na, int k, const int32_t* __restrict__ act, const uint32_t* __restrict__ all_off, const uint32_t* __restrict__ members,
                           const uint32_t* __restrict__ keep, const uint32_t* __restrict__ word_idx, const uint32_t* __restrict__ vote_off_w,
                           const float* __restrict__ lrf, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                           const float* __restrict__ center, uint32_t* __restrict__ word_src, uint32_t* __restrict__ vote_off,
                           uint32_t* __restrict__ vote_feature, float* __restrict__ vote_xyz) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const int w = act[a];
    if (w < 0 || !keep[w]) return;
    const uint32_t m0 = all_off[w], m1 = all_off[w + 1];
    uint32_t pos = 0;
    for (uint32_t t = m0; t < m1; ++t) pos += members[t] < (uint32_t)a;
    const uint32_t e = word_idx[w], v = vote_off_w[w] + pos;
    if (pos == 0) { word_src[e] = (uint32_t)w; vote_off[e] = vote_off_w[w]; }
    const int fi = a / k;
    // vote = rotateInto(centre - keyPos, LRF): q * p * conj(q) (utils.cpp:154-165, 568-574)
    const Quat q = rot_quaternion(lrf + (size_t)fi * 9);
    const Quat p{0.f, center[fi * 3] - kx[fi], center[fi * 3 + 1] - ky[fi], center[fi * 3 + 2] - kz[fi]};
    const Quat r = qmul(qmul(q, p), qconj(q));
    vote_feature[v] = (uint32_t)fi;
    vote_xyz[(size_t)v * 3] = r.x; vote_xyz[(size_t)v * 3 + 1] = r.y; vote_xyz[(size_t)v * 3 + 2] = r.z;
}

__global__ void k_tr_tail(const uint32_t* __restrict__ n_words_p, const uint32_t* __restrict__ n_votes_p, uint32_t* __restrict__ vote_off) {
    vote_off[*n_words_p] = *n_votes_p;
}

// sigma sample: pair p of class c = (feature sf[f0 + p / nw], codeword act[sw[w0 + p % nw]]) in the reference's loop order
struct SigmaClass { uint32_t f0, nf, w0, nw, d0; };      // offsets into the sample lists and the distance buffer
__global__ __launch_bounds__(256) void k_tr_sigma_dist(int n_classes, const SigmaClass* __restrict__ sc, const uint32_t* __restrict__ sf,
                                                       const int32_t* __restrict__ sw, int metric, const float* __restrict__ feats,
                                                       const float* __restrict__ words, int dim, float* __restrict__ dist) {
    __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
    const int c = blockIdx.y;
    const SigmaClass s = sc[c];
    const uint32_t np = s.nf * s.nw;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (uint32_t p = blockIdx.x * 4 + wv; p < np; p += gridDim.x * 4) {
        const uint32_t fi = sf[s.f0 + p / s.nw]; const int w = sw[s.w0 + p % s.nw];
        const float d = wave_functor(metric, feats + (size_t)fi * dim, words + (size_t)w * dim, dim, lane, s_terms[wv]);
        if (lane == 0) dist[s.d0 + p] = d;
    }
}
__global__ void k_tr_sigma(int n_classes, const SigmaClass* __restrict__ sc, const float* __restrict__ dist, float* __restrict__ sigma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_classes) return;
    const SigmaClass s = sc[c];
    const int num = (int)(s.nf * s.nw);
    if (s.nf == 0) { sigma[c] = __builtin_nanf(""); return; }     // class without training features
    float sum = 0.f;
    for (int i = 0; i < num; ++i) sum += dist[s.d0 + i];
    const float mean = sum / num;
    float variance = 0.f;
    for (int i = 0; i < num; ++i) { const float diff = dist[s.d0 + i] - mean; variance += diff * diff; }
    variance /= num - 1;
    sigma[c] = variance;
}

// computeWeights: one wave per vote; weights of the word's m activating features -> median (rank selection by bisection over
// the float bits: the weights are in [0, 1], non-negative floats order like unsigned integers)
#define TR_MAXM 2048
#define TR_MAXM_BIG 32768
__global__ __launch_bounds__(256) void k_tr_weights(uint32_t n_votes_cap, const uint32_t* __restrict__ n_votes_p, const uint32_t* __restrict__ vote_word,
                                                    const uint32_t* __restrict__ vote_off, const uint32_t* __restrict__ vote_feature,
                                                    const float* __restrict__ vote_xyz, const float* __restrict__ lrf,
                                                    const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                                                    const float* __restrict__ center, float* __restrict__ vote_weight, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t s_w[4][TR_MAXM];
    const uint32_t nv = *n_votes_p;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (uint32_t vi = blockIdx.x * 4 + wv; vi < nv; vi += gridDim.x * 4) {
        const uint32_t e = vote_word[vi];
        const uint32_t v0 = vote_off[e], v1 = vote_off[e + 1], m = v1 - v0;
        const uint32_t fi_i = vote_feature[vi];
        const float vx = vote_xyz[(size_t)vi * 3], vy = vote_xyz[(size_t)vi * 3 + 1], vz = vote_xyz[(size_t)vi * 3 + 2];
        const float mcx = center[fi_i * 3], mcy = center[fi_i * 3 + 1], mcz = center[fi_i * 3 + 2];
        if (m > TR_MAXM) { if (m > TR_MAXM_BIG && lane == 0) { atomicAdd(overflow, 1u); vote_weight[vi] = __builtin_nanf(""); } continue; }   // k_tr_weights_big
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t fj = vote_feature[v0 + j];
            const Quat q = rot_quaternion(lrf + (size_t)fj * 9);
            const Quat r = qmul(qmul(qconj(q), Quat{0.f, vx, vy, vz}), q);                  // rotateBack
            const float cx = kx[fj] + r.x, cy = ky[fj] + r.y, cz = kz[fj] + r.z;
            const float dx = cx - mcx, dy = cy - mcy, dz = cz - mcz;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            const float wgt = (float)exp((double)((-1 * (d * d)) / (0.5f * 0.5f)));
            s_w[wv][j] = __float_as_uint(wgt);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        auto select = [&](uint32_t rank) -> uint32_t {              // the value of the given rank (0-based, ascending)
            uint32_t sel = 0u;
            for (int bit = 30; bit >= 0; --bit) {
                const uint32_t cand = sel | (1u << bit);
                int c = 0;
                for (uint32_t j = lane; j < m; j += 64) c += s_w[wv][j] < cand;
                c = wave_sum_i(c);
                if ((uint32_t)c <= rank) sel = cand;
            }
            return sel;
        };
        float med;
        if (m == 1) med = __uint_as_float(s_w[wv][0]);
        else if (m % 2 == 0) med = (__uint_as_float(select(m / 2 - 1)) + __uint_as_float(select(m / 2))) / 2;
        else med = __uint_as_float(select(m / 2));
        if (lane == 0) vote_weight[vi] = med;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
// the same for the votes of big codewords (clustered codebooks: TR_MAXM < m <= TR_MAXM_BIG): a workgroup per vote, weights in 128 KB of LDS
__global__ __launch_bounds__(256) void k_tr_weights_big(const uint32_t* __restrict__ n_votes_p, const uint32_t* __restrict__ vote_word,
                                                        const uint32_t* __restrict__ vote_off, const uint32_t* __restrict__ vote_feature,
                                                        const float* __restrict__ vote_xyz, const float* __restrict__ lrf,
                                                        const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                                                        const float* __restrict__ center, float* __restrict__ vote_weight) {
    extern __shared__ uint32_t s_big[];
    __shared__ int s_cnt[4];
    const uint32_t nv = *n_votes_p;
    for (uint32_t vi = blockIdx.x; vi < nv; vi += gridDim.x) {
        const uint32_t e = vote_word[vi];
        const uint32_t v0 = vote_off[e], v1 = vote_off[e + 1], m = v1 - v0;
        if (m <= TR_MAXM || m > TR_MAXM_BIG) continue;                               // uniform across the workgroup
        const uint32_t fi_i = vote_feature[vi];
        const float vx = vote_xyz[(size_t)vi * 3], vy = vote_xyz[(size_t)vi * 3 + 1], vz = vote_xyz[(size_t)vi * 3 + 2];
        const float mcx = center[fi_i * 3], mcy = center[fi_i * 3 + 1], mcz = center[fi_i * 3 + 2];
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < m; j += 256) {
            const uint32_t fj = vote_feature[v0 + j];
            const Quat q = rot_quaternion(lrf + (size_t)fj * 9);
            const Quat r = qmul(qmul(qconj(q), Quat{0.f, vx, vy, vz}), q);
            const float cx = kx[fj] + r.x, cy = ky[fj] + r.y, cz = kz[fj] + r.z;
            const float dx = cx - mcx, dy = cy - mcy, dz = cz - mcz;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            s_big[j] = __float_as_uint((float)exp((double)((-1 * (d * d)) / (0.5f * 0.5f))));
        }
        __syncthreads();
        auto select = [&](uint32_t rank) -> uint32_t {
            uint32_t sel = 0u;
            for (int bit = 30; bit >= 0; --bit) {
                const uint32_t cand = sel | (1u << bit);
                int c = 0;
                for (uint32_t j = threadIdx.x; j < m; j += 256) c += s_big[j] < cand;
                c = wave_sum_i(c);
                __syncthreads();
                if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = c;
                __syncthreads();
                c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
                if ((uint32_t)c <= rank) sel = cand;
            }
            return sel;
        };
        const float med = m % 2 == 0 ? (__uint_as_float(select(m / 2 - 1)) + __uint_as_float(select(m / 2))) / 2 : __uint_as_float(select(m / 2));
        if (threadIdx.x == 0) vote_weight[vi] = med;
    }
}
// vote -> word (one thread per word fills its range) and per-class vote counts
__global__ void k_tr_stats1(const uint32_t* __restrict__ n_words_p, const uint32_t* __restrict__ vote_off, const uint32_t* __restrict__ vote_feature,
                            const uint32_t* __restrict__ feat_class, uint32_t* __restrict__ vote_word, uint32_t* __restrict__ num_features,
                            uint32_t* __restrict__ words_per_class) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *n_words_p) return;
    uint32_t prev = 0xffffffffu;
    for (uint32_t v = vote_off[e]; v < vote_off[e + 1]; ++v) {
        vote_word[v] = e;
        const uint32_t c = feat_class[vote_feature[v]];
        atomicAdd(&num_features[c], 1u);
        if (c != prev) { atomicAdd(&words_per_class[c], 1u); prev = c; }    // votes of a word are class-major: runs = distinct classes
    }
}
__global__ void k_tr_stats2(const uint32_t* __restrict__ n_words_p, const uint32_t* __restrict__ vote_off, const uint32_t* __restrict__ vote_feature,
                            const uint32_t* __restrict__ feat_class, const uint32_t* __restrict__ num_features, unsigned long long* __restrict__ term3_key) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *n_words_p) return;
    const uint32_t v0 = vote_off[e], v1 = vote_off[e + 1];
    // sum[e] = sum over the word's classes (ascending) of votes(c, e) / features(c)     (codebook.cpp:268-290)
    float sum = 0.f; bool first = true;
    for (uint32_t v = v0; v < v1;) {
        const uint32_t c = feat_class[vote_feature[v]];
        uint32_t r = v; while (r < v1 && feat_class[vote_feature[r]] == c) ++r;
        const float t = (float)(int)(r - v) / (float)(int)num_features[c];
        if (first) { sum = t; first = false; } else sum += t;
        v = r;
    }
    for (uint32_t v = v0; v < v1;) {
        const uint32_t c = feat_class[vote_feature[v]];
        uint32_t r = v; while (r < v1 && feat_class[vote_feature[r]] == c) ++r;
        const float t3 = ((float)(int)(r - v) / (float)(int)num_features[c]) / sum;
        atomicMax(&term3_key[c], ((unsigned long long)(e + 1u) << 32) | __float_as_uint(t3));     // the LAST word holding c wins
        v = r;
    }
}
__global__ void k_tr_stats3(const uint32_t* __restrict__ n_votes_p, const uint32_t* __restrict__ vote_word, const uint32_t* __restrict__ vote_off,
                            const uint32_t* __restrict__ vote_feature, const uint32_t* __restrict__ feat_class,
                            const uint32_t* __restrict__ words_per_class, const unsigned long long* __restrict__ term3_key,
                            float* __restrict__ vote_class_weight) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *n_votes_p) return;
    const uint32_t e = vote_word[v], c = feat_class[vote_feature[v]];
    const float term1 = 1.0f / (float)(int)words_per_class[c];
    const float term2 = 1.0f / (float)(int)(vote_off[e + 1] - vote_off[e]);
    const float term3 = __uint_as_float((uint32_t)(term3_key[c] & 0xffffffffull));
    vote_class_weight[v] = term1 * term2 * term3;
}

}  // namespace

// a codebook that only serves ismhip_knn: rows from a device matrix, one dummy vote per word
// light: a short-lived codebook (one k-means iteration): no rotated stage-1 image (a host eigen-solve of ~1 s) and no chi-square shadow
int ism_knn_only_codebook(ismhip_ctx* ctx, int n_words, int dim, const float* words_d, ismhip_codebook** out, bool light) {
    std::vector<float> words_h((size_t)n_words * dim);
    ISM_HIP(ctx, hipMemcpyAsync(words_h.data(), words_d, words_h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> ones((size_t)n_words + 1); std::iota(ones.begin(), ones.end(), 0u);
    std::vector<float> zxyz((size_t)n_words * 3, 0.f), sig1(1, 1.f);
    std::vector<uint32_t> zc((size_t)n_words, 0u);
    const bool was = ctx->codebook_light;
    ctx->codebook_light = light;
    const int rc = ismhip_codebook_create(ctx, n_words, dim, words_h.data(), nullptr, ones.data(), zxyz.data(), nullptr, nullptr, zc.data(), zc.data(), nullptr, nullptr,
                                          1, sig1.data(), out);
    ctx->codebook_light = was;
    return rc;
}

extern "C" int ismhip_train_activate(ismhip_ctx* ctx, int metric, int n, int dim, const float* desc, const float* lrf9,
                                     const float* kpx, const float* kpy, const float* kpz,
                                     const uint32_t* feat_class_h, const uint32_t* feat_model_h, const float* feat_center_h,
                                     int n_codewords, const float* codewords,
                                     int k, int clean_up, int n_classes,
                                     int32_t* n_words_out, uint32_t* word_src_out, uint32_t* vote_offsets_out, uint32_t* vote_feature_out,
                                     float* vote_xyz_out, float* vote_weight_out, float* vote_class_weight_out, float* class_sigma_out) {
    if (!ctx || n <= 0 || dim <= 0 || !desc || !lrf9 || !kpx || !kpy || !kpz || !feat_class_h || !feat_model_h || !feat_center_h || k <= 0 ||
        n_classes <= 0 || !n_words_out || !word_src_out || !vote_offsets_out || !vote_feature_out || !vote_xyz_out || !vote_weight_out ||
        !vote_class_weight_out || !class_sigma_out || (metric != ISMHIP_METRIC_L2SQ && metric != ISMHIP_METRIC_CHI2))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "train_activate: bad argument");
    if (dim > 1344) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "train_activate: descriptor longer than 1344 not built");
    for (int i = 0; i < n; ++i) {
        if ((int)feat_class_h[i] >= n_classes) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "train_activate: class id out of range");
        if (i && feat_class_h[i] < feat_class_h[i - 1]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "train_activate: features must be class-major (the reference iterates classes in ascending order)");
    }
    // the codewords: cluster centres (implicit_shape_model.cpp:445-475), or the features themselves (clustering_none.cpp:25-35)
    if (!codewords) { codewords = desc; n_codewords = n; }
    if (n_codewords <= 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "train_activate: no codewords");
    if (n_codewords < k) k = n_codewords;            // FLANN returns as many neighbours as there are rows: every feature activates every codeword
    const size_t na = (size_t)n * k;
    const int nw_in = n_codewords;
    // ---- step 1a: every feature activates its k nearest codewords (exact, ties -> lowest row)
    ismhip_codebook* cb = nullptr;
    int rc = ism_knn_only_codebook(ctx, nw_in, dim, codewords, &cb, false);
    if (rc != ISMHIP_OK) return rc;
    // scratch: activation list + CSR work arrays, carved from one slot (two passes: size, then pointers)
    unsigned long long* term3_key = nullptr; int32_t* act = nullptr; float* actd = nullptr;
    uint32_t *rank = nullptr, *cnt = nullptr, *keep = nullptr, *kcnt = nullptr, *all_off = nullptr, *word_idx = nullptr, *vote_off_w = nullptr, *members = nullptr,
             *word_src = nullptr, *vote_off = nullptr, *vote_feature = nullptr, *vote_word = nullptr, *feat_class = nullptr, *num_features = nullptr,
             *words_per_class = nullptr, *overflow = nullptr;
    float *vote_xyz = nullptr, *vote_weight = nullptr, *vote_class_weight = nullptr, *center = nullptr, *sigma = nullptr;
    char* base = nullptr;
    size_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        size_t off = 0;
        auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 15) / 16 * 16; return p; };
        term3_key = (unsigned long long*)take((size_t)n_classes * 8);
        act = (int32_t*)take(na * 4); actd = (float*)take(na * 4); rank = (uint32_t*)take(na * 4);
        cnt = (uint32_t*)take(((size_t)nw_in + 1) * 4); keep = (uint32_t*)take(((size_t)nw_in + 1) * 4); kcnt = (uint32_t*)take(((size_t)nw_in + 1) * 4);
        all_off = (uint32_t*)take(((size_t)nw_in + 1) * 4); word_idx = (uint32_t*)take(((size_t)nw_in + 1) * 4); vote_off_w = (uint32_t*)take(((size_t)nw_in + 1) * 4);
        members = (uint32_t*)take(na * 4); word_src = (uint32_t*)take(((size_t)nw_in + 1) * 4); vote_off = (uint32_t*)take(((size_t)nw_in + 1) * 4);
        vote_feature = (uint32_t*)take(na * 4); vote_word = (uint32_t*)take(na * 4); feat_class = (uint32_t*)take((size_t)n * 4);
        num_features = (uint32_t*)take((size_t)n_classes * 8 + 32);             // num_features | words_per_class | overflow (zeroed together)
        vote_xyz = (float*)take(na * 12); vote_weight = (float*)take(na * 4); vote_class_weight = (float*)take(na * 4);
        center = (float*)take((size_t)n * 12); sigma = (float*)take((size_t)n_classes * 4);
        if (pass == 0) {
            total = off;
            base = (char*)ism_scratch(ctx, SCR_TRAIN, total);
            if (!base) { ismhip_codebook_destroy(ctx, cb); return ISMHIP_ERR_NOMEM; }
        }
    }
    words_per_class = num_features + n_classes; overflow = words_per_class + n_classes;
    auto done = [&](int code) { ismhip_codebook_destroy(ctx, cb); return code; };
    rc = ismhip_knn(ctx, cb, metric, n, desc, k, act, actd);
    if (rc != ISMHIP_OK) return done(rc);
    hipStream_t st = ctx->stream;
#define TR_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { ism_set_err(ctx, ISMHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); return done(ISMHIP_ERR_HIP); } } while (0)
    TR_HIP(hipMemcpyAsync(feat_class, feat_class_h, (size_t)n * 4, hipMemcpyHostToDevice, st));
    TR_HIP(hipMemcpyAsync(center, feat_center_h, (size_t)n * 12, hipMemcpyHostToDevice, st));
    TR_HIP(hipMemsetAsync(cnt, 0, ((size_t)nw_in + 1) * 4, st));
    TR_HIP(hipMemsetAsync(num_features, 0, (size_t)n_classes * 8 + 32, st));      // num_features, words_per_class, overflow
    TR_HIP(hipMemsetAsync(term3_key, 0, (size_t)n_classes * 8, st));
    // ---- distributions: CSR in activation order, clean-up, votes
    const unsigned ga = (unsigned)((na + 255) / 256), gn = (unsigned)((nw_in + 255) / 256);
    hipLaunchKernelGGL(k_tr_count, dim3(ga), dim3(256), 0, st, (int)na, act, cnt, rank);
    hipLaunchKernelGGL(k_tr_flags, dim3(gn), dim3(256), 0, st, nw_in, clean_up, cnt, keep, kcnt);
    hipLaunchKernelGGL(k_tr_scan, dim3(1), dim3(1024), 0, st, nw_in, cnt, all_off, keep, word_idx, kcnt, vote_off_w);
    hipLaunchKernelGGL(k_tr_members, dim3(ga), dim3(256), 0, st, (int)na, act, all_off, rank, members);
    hipLaunchKernelGGL(k_tr_votes, dim3(ga), dim3(256), 0, st, (int)na, k, act, all_off, members, keep, word_idx, vote_off_w, lrf9, kpx, kpy, kpz, center,
                       word_src, vote_off, vote_feature, vote_xyz);
    hipLaunchKernelGGL(k_tr_tail, dim3(1), dim3(1), 0, st, word_idx + nw_in, vote_off_w + nw_in, vote_off);
    // vote_off[n_words] = n_votes: the totals of the scans
    TR_HIP(hipMemcpyAsync(n_words_out, word_idx + nw_in, 4, hipMemcpyDeviceToHost, st));
    uint32_t n_votes_h = 0;
    TR_HIP(hipMemcpyAsync(&n_votes_h, vote_off_w + nw_in, 4, hipMemcpyDeviceToHost, st));
    // ---- class sigma^2: sample lists from the (host) class / model ids, distances and sequential sums on the device
    {
        std::vector<SigmaClass> sc((size_t)n_classes);
        std::vector<uint32_t> sf; std::vector<uint32_t> swf;           // sample features; features whose activations form the word sample
        uint32_t d0 = 0;
        int i0 = 0;
        for (int c = 0; c < n_classes; ++c) {
            int i1 = i0; while (i1 < n && (int)feat_class_h[i1] == c) ++i1;
            SigmaClass s{(uint32_t)sf.size(), 0u, 0u, 0u, d0};
            const int nc = i1 - i0;
            if (nc > 0) {
                const int max_elements = (int)std::sqrt((double)nc);
                int words = 0, t = i0;
                s.w0 = (uint32_t)swf.size() * (uint32_t)k;
                while (t < i1 && words < max_elements) { swf.push_back((uint32_t)t); words += k; ++t; }   // :146-147 (checked before each feature's k words)
                s.nw = (uint32_t)words;
                int m = i0;
                while (m < i1 && (int)(sf.size() - s.f0) < max_elements) {                              // :149-150 whole models
                    int e = m; while (e < i1 && feat_model_h[e] == feat_model_h[m]) ++e;
                    for (int q = m; q < e; ++q) sf.push_back((uint32_t)q);
                    m = e;
                }
                s.nf = (uint32_t)(sf.size() - s.f0);
                d0 += s.nf * s.nw;
            }
            sc[c] = s; i0 = i1;
        }
        // word sample = the activations of the features in swf, flattened (every feature has k activations when n >= k)
        const size_t nsw = swf.size() * (size_t)k;
        char* sb = (char*)ism_scratch(ctx, SCR_TRAIN2, sc.size() * sizeof(SigmaClass) + (sf.size() + 1) * 4 + (nsw + 1) * 4 + ((size_t)d0 + 1) * 4 + (swf.size() + 1) * 4);
        if (!sb) return done(ISMHIP_ERR_NOMEM);
        SigmaClass* d_sc = (SigmaClass*)sb; sb += sc.size() * sizeof(SigmaClass);
        uint32_t* d_sf = (uint32_t*)sb; sb += (sf.size() + 1) * 4;
        int32_t* d_sw = (int32_t*)sb; sb += (nsw + 1) * 4;
        float* d_dist = (float*)sb; sb += ((size_t)d0 + 1) * 4;
        TR_HIP(hipMemcpyAsync(d_sc, sc.data(), sc.size() * sizeof(SigmaClass), hipMemcpyHostToDevice, st));
        if (!sf.empty()) TR_HIP(hipMemcpyAsync(d_sf, sf.data(), sf.size() * 4, hipMemcpyHostToDevice, st));
        // the sampled features of a class are its FIRST ones, so their activations are k-wide row ranges of act: copy device to device
        {
            size_t o = 0;
            for (int c = 0; c < n_classes; ++c) {
                const uint32_t cnt_f = sc[c].nw / (uint32_t)k;
                if (cnt_f) TR_HIP(hipMemcpyAsync(d_sw + o, act + (size_t)swf[sc[c].w0 / (uint32_t)k] * k, (size_t)cnt_f * k * 4, hipMemcpyDeviceToDevice, st));
                o += (size_t)cnt_f * k;
            }
        }
        if (d0) {
            hipLaunchKernelGGL(k_tr_sigma_dist, dim3(1024, n_classes), dim3(256), 0, st, n_classes, d_sc, d_sf, d_sw, metric, desc, codewords, dim, d_dist);
        }
        hipLaunchKernelGGL(k_tr_sigma, dim3((n_classes + 63) / 64), dim3(64), 0, st, n_classes, d_sc, d_dist, sigma);
    }
    // ---- weights
    const uint32_t* n_words_p = word_idx + nw_in; const uint32_t* n_votes_p = vote_off_w + nw_in;
    hipLaunchKernelGGL(k_tr_stats1, dim3(gn), dim3(256), 0, st, n_words_p, vote_off, vote_feature, feat_class, vote_word, num_features, words_per_class);
    hipLaunchKernelGGL(k_tr_weights, dim3(2048), dim3(256), 0, st, (uint32_t)na, n_votes_p, vote_word, vote_off, vote_feature, vote_xyz, lrf9, kpx, kpy, kpz, center,
                       vote_weight, overflow);
    {   // codewords with 2049 .. 32768 votes: clustered codebooks, but also a hub word or many duplicate descriptors with Clustering
        // "None" and k > 1 / KNNRule (every tie goes to the lowest row) -- the kernel skips every smaller word by itself
        if (!ctx->attr_done.count((const void*)k_tr_weights_big)) {
            TR_HIP(hipFuncSetAttribute((const void*)k_tr_weights_big, hipFuncAttributeMaxDynamicSharedMemorySize, TR_MAXM_BIG * 4));
            ctx->attr_done.insert((const void*)k_tr_weights_big);
        }
        hipLaunchKernelGGL(k_tr_weights_big, dim3(1024), dim3(256), TR_MAXM_BIG * 4, st, n_votes_p, vote_word, vote_off, vote_feature, vote_xyz, lrf9, kpx, kpy, kpz, center,
                           vote_weight);
    }
    hipLaunchKernelGGL(k_tr_stats2, dim3(gn), dim3(256), 0, st, n_words_p, vote_off, vote_feature, feat_class, num_features, term3_key);
    hipLaunchKernelGGL(k_tr_stats3, dim3(ga), dim3(256), 0, st, n_votes_p, vote_word, vote_off, vote_feature, feat_class, words_per_class, term3_key, vote_class_weight);
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ism_set_err(ctx, ISMHIP_ERR_HIP, std::string("train_activate launch: ") + hipGetErrorString(e)); return done(ISMHIP_ERR_HIP); }
    }
    TR_HIP(hipStreamSynchronize(st));
    const int nw = *n_words_out;
    // vote_off[n_words] was not written by any activation: it is the total
    uint32_t ovf = 0;
    TR_HIP(hipMemcpy(&ovf, overflow, 4, hipMemcpyDeviceToHost));
    if (ovf) { ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "train_activate: a codeword with more than 32768 votes is not built"); return done(ISMHIP_ERR_UNSUPPORTED); }
    if (nw > 0) {
        TR_HIP(hipMemcpy(word_src_out, word_src, (size_t)nw * 4, hipMemcpyDeviceToHost));
        TR_HIP(hipMemcpy(vote_offsets_out, vote_off, (size_t)nw * 4, hipMemcpyDeviceToHost));
    }
    vote_offsets_out[nw] = n_votes_h;
    if (n_votes_h) {
        TR_HIP(hipMemcpy(vote_feature_out, vote_feature, (size_t)n_votes_h * 4, hipMemcpyDeviceToHost));
        TR_HIP(hipMemcpy(vote_xyz_out, vote_xyz, (size_t)n_votes_h * 12, hipMemcpyDeviceToHost));
        TR_HIP(hipMemcpy(vote_weight_out, vote_weight, (size_t)n_votes_h * 4, hipMemcpyDeviceToHost));
        TR_HIP(hipMemcpy(vote_class_weight_out, vote_class_weight, (size_t)n_votes_h * 4, hipMemcpyDeviceToHost));
    }
    TR_HIP(hipMemcpy(class_sigma_out, sigma, (size_t)n_classes * 4, hipMemcpyDeviceToHost));
#undef TR_HIP
    return done(ISMHIP_OK);
}
