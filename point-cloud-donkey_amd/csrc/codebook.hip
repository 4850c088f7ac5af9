// codebook.hip — device-resident codebook (FlannHelper dataset + CodewordDistribution vote tables) and vote casting.
// Reference: utils/flann_helper.cpp:21-70 (row-major Nc x D dataset in getCodewords() order),
// codebook/codeword_distribution.cpp:73-167 (castVotes / castVote), codebook/codebook.cpp:541-554 (second loop of
// Codebook::castVotes), utils/utils.cpp:136-178, 342-394, 560-574 (LRF -> quaternion, rotateBack), voting/voting.cpp:58-77.
//
// HBM layout: words [n_words_pad x dim_pad] zero-padded so the kNN tiles never branch on bounds (n_words_pad multiple
// of 256, dim_pad multiple of 32); votes as CSR over words with SoA payloads. Vote casting is HBM-bound
// (V*(12+4+4+4) read + V*32 written) and tiny next to kNN.
#include "common.h"
#include <cstring>
#include <cmath>
#include <limits>

int ism_codebook_split_bf16(ismhip_ctx* ctx, ismhip_codebook* cb, uint32_t absmax_bits);
int ism_codebook_build_pca(ismhip_ctx* ctx, ismhip_codebook* cb);       // pca.hip: rotated, truncated stage-1 image
void ism_codebook_free_pca(ismhip_codebook* cb);

namespace {

__global__ void k_word_norms(const float* __restrict__ words, int n_words, int n_words_pad, int dim_pad, float* __restrict__ norm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_words_pad) return;
    const int lane = lane_id();
    float s = 0.f;
    for (int c = lane; c < dim_pad; c += 64) { const float v = words[(size_t)row * dim_pad + c]; s += v * v; }
    s = wave_sum_f(s);
    if (lane == 0) norm[row] = row < n_words ? s : __builtin_inff();   // padding rows can never win
}

// shadow row p = sqrt(word perm[p]) (p < n_words), zero rows behind
__global__ void k_sqrt_permuted(const float* __restrict__ src, const uint32_t* __restrict__ perm, int n_words, int dim_pad, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t row = i / dim_pad; const int col = (int)(i % dim_pad);
    dst[i] = row < (size_t)n_words ? sqrtf(src[(size_t)perm[row] * dim_pad + col]) : 0.f;
}

#include "quat.h"

struct VoteArgs {
    const float* word_weight; const uint32_t* vote_off; const float* vote_xyz; const float* vote_weight;
    const float* vote_class_weight; const uint32_t* vote_class; const uint32_t* vote_instance;
    const float* vote_bbox_quat; const float* vote_bbox_size; const float* class_sigma;
    int n_words, n_classes, maxv; uint32_t flags;
    int nq, k; const float* lrf; const float *kx, *ky, *kz; const int32_t* idx; const float* dist;
    float* pos; float* w; int32_t* cls; int32_t* inst; int32_t* cw; float* bq; float* bs;
};

__global__ __launch_bounds__(256) void k_cast_votes(VoteArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nq * a.k) return;
    const int f = t / a.k;
    const size_t slot0 = (size_t)t * a.maxv;
    for (int v = 0; v < a.maxv; ++v) {
        const size_t s = slot0 + v;
        a.cls[s] = -1; a.inst[s] = -1; a.cw[s] = -1; a.w[s] = 0.f;
        a.pos[s * 3] = 0.f; a.pos[s * 3 + 1] = 0.f; a.pos[s * 3 + 2] = 0.f;
        if (a.bq) { a.bq[s * 4] = 1.f; a.bq[s * 4 + 1] = 0.f; a.bq[s * 4 + 2] = 0.f; a.bq[s * 4 + 3] = 0.f; }
        if (a.bs) { a.bs[s * 3] = 0.f; a.bs[s * 3 + 1] = 0.f; a.bs[s * 3 + 2] = 0.f; }
    }
    const int c = a.idx[t];
    if (c < 0 || c >= a.n_words) return;
    const float d = a.dist[t];                                               // codeword_distribution.cpp:87
    const Quat rq = rot_quaternion(a.lrf + (size_t)f * 9);
    const Quat rqc = qconj(rq);
    const uint32_t v0 = a.vote_off[c], v1 = a.vote_off[c + 1];
    for (uint32_t v = v0; v < v1; ++v) {
        const size_t s = slot0 + (v - v0);
        const uint32_t classId = a.vote_class[v];
        const float classWeight = a.vote_class_weight ? a.vote_class_weight[v] : 1.0f;
        const float classSigma = ((int)classId < a.n_classes) ? a.class_sigma[classId] : 1.0f;   // :108-117
        float weight = 1.0f;
        if (a.flags & ISMHIP_W_CLASS) weight = weight * classWeight;
        if (a.flags & ISMHIP_W_VOTE) weight = weight * (a.vote_weight ? a.vote_weight[v] : 1.0f);
        if (a.flags & ISMHIP_W_MATCHING) {
            // gaussDist (:23-26), evaluated in double like the reference
            const double sg = (double)classSigma, dd = (double)d;
            const float matching = (float)((1.0 / sqrt(2.0 * 3.14159265358979323846 * sg)) * exp(-(dd * dd) / (2.0 * sg)));
            weight = weight * matching;
        }
        if (a.flags & ISMHIP_W_CODEWORD) weight = weight * (a.word_weight ? a.word_weight[c] : 1.0f);
        if (fabsf(d) > 2 * classSigma) continue;                              // :131
        if (weight < 1.1920928955078125e-07f) continue;                       // :137 FLT_EPSILON
        // rotateBack: conj(q) * p * q  (utils.cpp:167-178, 560-566)
        const Quat p{0.f, a.vote_xyz[v * 3], a.vote_xyz[v * 3 + 1], a.vote_xyz[v * 3 + 2]};
        const Quat rb = qmul(qmul(rqc, p), rq);
        a.pos[s * 3 + 0] = a.kx[f] + rb.x; a.pos[s * 3 + 1] = a.ky[f] + rb.y; a.pos[s * 3 + 2] = a.kz[f] + rb.z;
        a.w[s] = weight; a.cls[s] = (int32_t)classId; a.inst[s] = (int32_t)a.vote_instance[v]; a.cw[s] = c;
        if (a.bq) {
            const Quat b = a.vote_bbox_quat ? Quat{a.vote_bbox_quat[v * 4], a.vote_bbox_quat[v * 4 + 1], a.vote_bbox_quat[v * 4 + 2], a.vote_bbox_quat[v * 4 + 3]}
                                            : Quat{1.f, 0.f, 0.f, 0.f};
            const Quat r = qmul(b, rq);                                       // :162
            a.bq[s * 4] = r.w; a.bq[s * 4 + 1] = r.x; a.bq[s * 4 + 2] = r.y; a.bq[s * 4 + 3] = r.z;
        }
        if (a.bs && a.vote_bbox_size) { a.bs[s * 3] = a.vote_bbox_size[v * 3]; a.bs[s * 3 + 1] = a.vote_bbox_size[v * 3 + 1]; a.bs[s * 3 + 2] = a.vote_bbox_size[v * 3 + 2]; }
    }
}

template <typename Tt>
bool upload(Tt** dst, const Tt* src_h, size_t n) {
    if (n == 0) n = 1;
    if (hipMalloc((void**)dst, n * sizeof(Tt)) != hipSuccess) return false;
    if (src_h && hipMemcpy(*dst, src_h, n * sizeof(Tt), hipMemcpyHostToDevice) != hipSuccess) return false;
    return true;
}

}  // namespace

extern "C" {

int ismhip_codebook_create(ismhip_ctx* ctx, int n_words, int dim, const float* words_h, const float* word_weight_h,
                           const uint32_t* vote_offsets_h, const float* vote_xyz_h, const float* vote_weight_h,
                           const float* vote_class_weight_h, const uint32_t* vote_class_h, const uint32_t* vote_instance_h,
                           const float* vote_bbox_quat_h, const float* vote_bbox_size_h,
                           int n_classes, const float* class_sigma_h, ismhip_codebook** out) {
    if (!ctx || !out || n_words <= 0 || dim <= 0 || !words_h || !vote_offsets_h || !vote_xyz_h || !vote_class_h ||
        !vote_instance_h || n_classes <= 0 || !class_sigma_h)
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "codebook_create: bad argument");
    *out = nullptr;
    if (vote_offsets_h[0] != 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "codebook_create: vote offsets must start at 0");
    int maxv = 0;
    for (int c = 0; c < n_words; ++c) {
        if (vote_offsets_h[c + 1] < vote_offsets_h[c]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "codebook_create: vote offsets not monotone");
        maxv = std::max(maxv, (int)(vote_offsets_h[c + 1] - vote_offsets_h[c]));
    }
    ISM_HIP(ctx, hipSetDevice(ctx->device));
    ismhip_codebook* cb = new ismhip_codebook();
    cb->n_words = n_words; cb->dim = dim; cb->n_classes = n_classes;
    cb->dim_pad = (dim + 31) / 32 * 32;
    cb->n_words_pad = (n_words + 255) / 256 * 256;      // multiple of every kNN tile height (64, 128, 256)
    cb->n_votes = (int)vote_offsets_h[n_words]; cb->max_votes = maxv;
    auto fail = [&](int code, const char* msg) { ismhip_codebook_destroy(ctx, cb); return ism_set_err(ctx, code, msg); };
    const size_t wbytes = (size_t)cb->n_words_pad * cb->dim_pad * sizeof(float);
    if (hipMalloc((void**)&cb->words, wbytes) != hipSuccess) return fail(ISMHIP_ERR_NOMEM, "codebook_create: words");
    if (hipMemset(cb->words, 0, wbytes) != hipSuccess) return fail(ISMHIP_ERR_HIP, "codebook_create: memset");
    if (hipMemcpy2D(cb->words, (size_t)cb->dim_pad * 4, words_h, (size_t)dim * 4, (size_t)dim * 4, n_words, hipMemcpyHostToDevice) != hipSuccess)
        return fail(ISMHIP_ERR_HIP, "codebook_create: words copy");
    if (hipMalloc((void**)&cb->word_norm, (size_t)cb->n_words_pad * 4) != hipSuccess) return fail(ISMHIP_ERR_NOMEM, "codebook_create: norms");
    const size_t nv = cb->n_votes;
    bool ok = upload(&cb->vote_off, vote_offsets_h, (size_t)n_words + 1) && upload(&cb->vote_xyz, vote_xyz_h, nv * 3) &&
              upload(&cb->vote_class, vote_class_h, nv) && upload(&cb->vote_instance, vote_instance_h, nv) &&
              upload(&cb->class_sigma, class_sigma_h, (size_t)n_classes);
    if (ok && word_weight_h) ok = upload(&cb->word_weight, word_weight_h, (size_t)n_words);
    if (ok && vote_weight_h) ok = upload(&cb->vote_weight, vote_weight_h, nv);
    if (ok && vote_class_weight_h) ok = upload(&cb->vote_class_weight, vote_class_weight_h, nv);
    if (ok && vote_bbox_quat_h) ok = upload(&cb->vote_bbox_quat, vote_bbox_quat_h, nv * 4);
    if (ok && vote_bbox_size_h) ok = upload(&cb->vote_bbox_size, vote_bbox_size_h, nv * 3);
    if (ok) {
        std::vector<uint32_t> wc(n_words, 0u);
        for (int c = 0; c < n_words; ++c) if (vote_offsets_h[c + 1] > vote_offsets_h[c]) wc[c] = vote_class_h[vote_offsets_h[c]];
        ok = upload(&cb->word_class, wc.data(), (size_t)n_words);
    }
    if (!ok) return fail(ISMHIP_ERR_NOMEM, "codebook_create: vote tables");
    hipLaunchKernelGGL(k_word_norms, dim3((cb->n_words_pad + 3) / 4), dim3(256), 0, ctx->stream, cb->words, n_words, cb->n_words_pad, cb->dim_pad, cb->word_norm);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ISMHIP_ERR_HIP, "codebook_create: norms kernel");
    {
        std::vector<float> nh(n_words);
        if (hipMemcpy(nh.data(), cb->word_norm, (size_t)n_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(ISMHIP_ERR_HIP, "codebook_create: norms copy");
        float mx = 0.f;
        for (float v : nh) if (v > mx || v != v) mx = v;        // a NaN word poisons the bound -> every query takes the exact fallback
        cb->max_norm2 = mx;
    }
    {
        uint32_t amax = 0u;                                       // largest |element| as float bits (NaN/inf sort last)
        bool nonneg = true;                                       // histograms: every element >= 0 (NaN counts as not)
        for (size_t i = 0; i < (size_t)n_words * dim; ++i) {
            uint32_t b; memcpy(&b, &words_h[i], 4);
            if (!(words_h[i] >= 0.f)) nonneg = false;
            b &= 0x7fffffffu; amax = b > amax ? b : amax;
        }
        cb->words_nonneg = nonneg;
        int rc = ism_codebook_split_bf16(ctx, cb, amax);
        if (rc != ISMHIP_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(rc != ISMHIP_OK ? rc : ISMHIP_ERR_HIP, "codebook_create: bf16 split");
        if (!ctx->codebook_light) rc = ism_codebook_build_pca(ctx, cb);
        if (rc != ISMHIP_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(rc != ISMHIP_OK ? rc : ISMHIP_ERR_HIP, "codebook_create: rotated image");
        // chi-square candidates on the matrix cores: 16-bit images, norms and scales of sqrt(words) in a shadow codebook (histogram
        // codebooks of descriptors longer than 64 only: short ones take the exact-f32 contraction for L2 and the VALU kernel for chi-square)
        if (nonneg && dim > 64 && ctx->knn_hellinger && n_words >= 1024 && !ctx->codebook_light) {
            ismhip_codebook* sh = new ismhip_codebook();
            cb->chi_shadow = sh;
            sh->n_words = n_words; sh->dim = dim; sh->dim_pad = cb->dim_pad; sh->n_words_pad = cb->n_words_pad; sh->n_classes = n_classes;
            if (hipMalloc((void**)&sh->words, wbytes) != hipSuccess || hipMalloc((void**)&sh->word_norm, (size_t)cb->n_words_pad * 4) != hipSuccess)
                return fail(ISMHIP_ERR_NOMEM, "codebook_create: sqrt image");
            // The rows of the shadow are a fixed pseudo-random PERMUTATION of the codebook's. Codebooks are class-major (one class ~ one
            // 256-row tile at 10 k words / 51 classes), so the dozens of rows a proof must hold as candidates would all sit in the eight
            // candidate lists of ONE tile's split; shuffled, they spread over all splits and slots (measured, CSHOT-1344: 36 % -> 3 % of
            // the queries unproven). k_knn_rerank_hell maps candidate rows back through shadow_perm.
            std::vector<uint32_t> perm(n_words);
            for (int i = 0; i < n_words; ++i) perm[i] = (uint32_t)i;
            uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)n_words;
            for (int i = n_words - 1; i > 0; --i) {
                st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                std::swap(perm[i], perm[(size_t)(z % (uint64_t)(i + 1))]);
            }
            if (!upload(&sh->shadow_perm, perm.data(), (size_t)n_words)) return fail(ISMHIP_ERR_NOMEM, "codebook_create: sqrt image permutation");
            const size_t tot = (size_t)cb->n_words_pad * cb->dim_pad;
            hipLaunchKernelGGL(k_sqrt_permuted, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, cb->words, sh->shadow_perm, n_words, cb->dim_pad, sh->words, tot);
            hipLaunchKernelGGL(k_word_norms, dim3((cb->n_words_pad + 3) / 4), dim3(256), 0, ctx->stream, sh->words, n_words, cb->n_words_pad, cb->dim_pad, sh->word_norm);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ISMHIP_ERR_HIP, "codebook_create: sqrt image kernels");
            std::vector<float> nh(n_words);
            if (hipMemcpy(nh.data(), sh->word_norm, (size_t)n_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(ISMHIP_ERR_HIP, "codebook_create: sqrt norms copy");
            float mx = 0.f;
            for (float v : nh) if (v > mx || v != v) mx = v;
            sh->max_norm2 = mx;
            float amaxf; memcpy(&amaxf, &amax, 4);
            const float samax = sqrtf(amaxf);                           // largest sqrt element = sqrt of the largest element
            uint32_t samax_bits; memcpy(&samax_bits, &samax, 4);
            rc = ism_codebook_split_bf16(ctx, sh, samax_bits);
            if (rc != ISMHIP_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(rc != ISMHIP_OK ? rc : ISMHIP_ERR_HIP, "codebook_create: sqrt 16-bit images");
            (void)hipFree(sh->words); sh->words = nullptr;             // only the images are needed
        }
    }
    *out = cb;
    return ISMHIP_OK;
}

int ismhip_codebook_destroy(ismhip_ctx* ctx, ismhip_codebook* cb) {
    if (!cb) return ISMHIP_ERR_INVALID;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    void* ptrs[] = {cb->words, cb->word_norm, cb->word_weight, cb->vote_off, cb->vote_xyz, cb->vote_weight, cb->vote_class_weight,
                    cb->vote_class, cb->vote_instance, cb->vote_bbox_quat, cb->vote_bbox_size, cb->class_sigma, cb->word_class, cb->words_bf16_hi, cb->words_f16t};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    ism_codebook_free_pca(cb);
    if (cb->chi_shadow) {
        ismhip_codebook* sh = cb->chi_shadow;
        void* sp[] = {sh->words, sh->word_norm, sh->words_bf16_hi, sh->words_f16t, sh->shadow_perm};
        for (void* p : sp) if (p) (void)hipFree(p);
        delete sh;
    }
    delete cb;
    return ISMHIP_OK;
}

int ismhip_codebook_set_word_class(ismhip_ctx* ctx, ismhip_codebook* cb, const uint32_t* word_class_h) {
    if (!ctx || !cb || !word_class_h) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "codebook_set_word_class: bad argument");
    ISM_HIP(ctx, hipMemcpy(cb->word_class, word_class_h, (size_t)cb->n_words * 4, hipMemcpyHostToDevice));
    return ISMHIP_OK;
}

int ismhip_codebook_max_votes_per_word(const ismhip_codebook* cb) { return cb ? cb->max_votes : ISMHIP_ERR_INVALID; }
int ismhip_codebook_stage1_dims(const ismhip_codebook* cb, float* energy_out) {
    if (!cb) return ISMHIP_ERR_INVALID;
    if (energy_out) *energy_out = cb->pca.m > 0 ? cb->pca.energy : 1.0f;
    return cb->pca.m;
}

int ismhip_codebook_stage2_dims(const ismhip_codebook* cb, float* energy_out) {
    if (!cb) return ISMHIP_ERR_INVALID;
    if (energy_out) *energy_out = cb->pca2.m > 0 ? cb->pca2.energy : 1.0f;
    return cb->pca2.m;
}

int ismhip_cast_votes(ismhip_ctx* ctx, const ismhip_codebook* cb, uint32_t weight_flags,
                      int nq, const float* lrf9, const float* kpx, const float* kpy, const float* kpz,
                      int k, const int32_t* idx, const float* dist,
                      float* vote_pos_out, float* vote_weight_out, int32_t* vote_class_out,
                      int32_t* vote_instance_out, int32_t* vote_codeword_out,
                      float* vote_bbox_quat_out, float* vote_bbox_size_out) {
    if (!ctx || !cb || nq < 0 || k <= 0 || !lrf9 || !kpx || !kpy || !kpz || !idx || !dist || !vote_pos_out || !vote_weight_out ||
        !vote_class_out || !vote_instance_out || !vote_codeword_out)
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cast_votes: bad argument");
    if (nq == 0 || cb->max_votes == 0) return ISMHIP_OK;
    VoteArgs a;
    a.word_weight = cb->word_weight; a.vote_off = cb->vote_off; a.vote_xyz = cb->vote_xyz; a.vote_weight = cb->vote_weight;
    a.vote_class_weight = cb->vote_class_weight; a.vote_class = cb->vote_class; a.vote_instance = cb->vote_instance;
    a.vote_bbox_quat = cb->vote_bbox_quat; a.vote_bbox_size = cb->vote_bbox_size; a.class_sigma = cb->class_sigma;
    a.n_words = cb->n_words; a.n_classes = cb->n_classes; a.maxv = cb->max_votes; a.flags = weight_flags;
    a.nq = nq; a.k = k; a.lrf = lrf9; a.kx = kpx; a.ky = kpy; a.kz = kpz; a.idx = idx; a.dist = dist;
    a.pos = vote_pos_out; a.w = vote_weight_out; a.cls = vote_class_out; a.inst = vote_instance_out; a.cw = vote_codeword_out;
    a.bq = vote_bbox_quat_out; a.bs = vote_bbox_size_out;
    TimerScope ts(ctx, "cast_votes");
    const int n = nq * k;
    hipLaunchKernelGGL(k_cast_votes, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_cast_votes");
    return ISMHIP_OK;
}

}  // extern "C"
