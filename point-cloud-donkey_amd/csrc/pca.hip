// pca.hip — rotated, truncated stage-1 image of the exact squared-L2 codeword search (DESIGN.md §4.1 "stage 1 on fewer dimensions").
// Reference contract served: ActivationStrategyKNN::activateKNN with FLANNExactMatch (activation_strategy/activation_strategy_knn.h:57-72):
// the k smallest values of flann::L2 (utils/distance.h:45). Nothing here changes a result; it only makes the candidate stage cheaper.
//
// Descriptor codebooks have a steep spectrum (SHOT-352 of the bench generator: the 192 leading principal directions of the
// codebook's second-moment matrix hold 99.1 % of it). For an orthonormal R (m x D), |R (q - c)|^2 <= |q - c|^2: a distance over the
// m leading rotated coordinates is a LOWER bound of the functor value, which is exactly what the drop-bound proof of k_knn_rerank
// consumes. So stage 1 of the two-stage search runs k_knn_l2_ring16 on m / 32 slices of the rotated f16 images instead of D / 32
// slices of the original ones; the re-rank, stage 2 (all D dimensions) and the exact scan stay in the original coordinates.
//
// What is rigorous about it (the proof must hold for the matrices and images actually used, not for their ideal versions):
//   R        the fp32 matrix that is uploaded; sigma_max(R)^2 <= 1 + |R R^T - I|_F, computed in fp64 from those fp32 values
//   x^       := f16 image of fl(R x) divided by its scale (an exactly representable fp32 vector). The candidate kernel's score is
//            |c^|^2 - 2 c^.q^ up to ACCUMULATION error only (products of f16 values are exact in fp32), so
//            |q^ - c^|^2 >= |q^|^2 + score - eps_acc                                        (eps_acc: VerifyParams of knn.hip)
//   |x^ - R x|_2 <= d_rel |x|_2 + d_abs: fp32 rotation (any summation order, 2^-23 per add) + round-to-nearest f16 + flush-to-zero floor
//   => |q - c| >= (|q^ - c^| - delta(q) - delta(c)) / sigma_max(R)
// The query image uses a scale FIXED per codebook (queries are assumed to be at most twice as long as the longest codeword; three
// more bits of f16 headroom above that; a query beyond it overflows to inf, fails its proof and is searched by stage 2 in the
// original coordinates): no pass over the query batch to find its largest element.
#include "common.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <numeric>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

// ---- second-moment matrix S = W^T W, in row chunks (partial sums are added on the host in a fixed order) ----------------------
#define SM_ROWS 64
__global__ __launch_bounds__(256) void k_second_moment(const float* __restrict__ w, int n_rows, int ld, int rows_per_chunk, int nb, float* __restrict__ part) {
    __shared__ float sA[SM_ROWS][33], sB[SM_ROWS][33];
    // upper-triangle block pairs only: blockIdx.x enumerates (bi <= bj)
    int bi = 0, rem = blockIdx.x;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r_beg = blockIdx.y * rows_per_chunk, r_end = min(n_rows, r_beg + rows_per_chunk);
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    for (int r0 = r_beg; r0 < r_end; r0 += SM_ROWS) {
        __syncthreads();
        for (int e = threadIdx.x; e < SM_ROWS * 32; e += 256) {
            const int rr = e >> 5, c = e & 31, r = r0 + rr;
            sA[rr][c] = r < r_end ? w[(size_t)r * ld + bi * 32 + c] : 0.f;
            sB[rr][c] = r < r_end ? w[(size_t)r * ld + bj * 32 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < SM_ROWS; ++rr) {
            const float x0 = sA[rr][2 * ty], x1 = sA[rr][2 * ty + 1], y0 = sB[rr][2 * tx], y1 = sB[rr][2 * tx + 1];
            a00 = fmaf(x0, y0, a00); a01 = fmaf(x0, y1, a01); a10 = fmaf(x1, y0, a10); a11 = fmaf(x1, y1, a11);
        }
    }
    const int dimp = nb * 32;
    float* o = part + (size_t)blockIdx.y * dimp * dimp;
    o[(size_t)(bi * 32 + 2 * ty) * dimp + bj * 32 + 2 * tx] = a00;     o[(size_t)(bi * 32 + 2 * ty) * dimp + bj * 32 + 2 * tx + 1] = a01;
    o[(size_t)(bi * 32 + 2 * ty + 1) * dimp + bj * 32 + 2 * tx] = a10; o[(size_t)(bi * 32 + 2 * ty + 1) * dimp + bj * 32 + 2 * tx + 1] = a11;
}

// ---- Y = X R^T on the FP32 matrix cores, written as the scaled f16 image in the ring kernel's streaming layout -------------------
// v_mfma_f32_32x32x2_f32 with the BASIS VECTORS as rows and the descriptors as columns: a lane then holds, for ONE descriptor
// (column lane & 31), the rotated coordinates 8 j + 4 h + 0..3 of every 32-wide output tile -- four consecutive halves of the
// image row, one 8-byte store. 128 descriptors per workgroup (4 waves x 32), all m outputs per wave (m / 32 accumulator tiles),
// K walked in 32-wide chunks staged through LDS (rows padded to 36 floats: conflict-free 16-byte fragment reads).
// Image layout (k_to_f16_tiled of knn.hip): [256-row tile][32-k slice][row][4 x 16 B], segment p of row r holds logical segment
// p ^ F[(r >> 2) & 3], F = {0,2,3,1}.
#define ROT_LD 36
template <int NT>
__global__ __launch_bounds__(256, 2) void k_rotate_f16t(const float* __restrict__ x, int n, int ldx, int kdim,
                                                         const float* __restrict__ rmat, int m, float scale, u16* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) float rot_smem[];
    float* sX = rot_smem;                       // [128][ROT_LD]
    float* sR = rot_smem + 128 * ROT_LD;        // [NT * 32][ROT_LD]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int nct = m >> 5;                     // output tiles in use (<= NT)
    const size_t row0 = (size_t)blockIdx.x * 128;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int srow = tid >> 3, scol = (tid & 7) * 4;          // staging: thread -> (row srow + 32 i, float4 column)
    for (int k0 = 0; k0 < kdim; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            size_t r = row0 + srow + 32 * i; r = r < (size_t)n ? r : (size_t)n - 1;       // clamped duplicates are zeroed on store
            *(f32x4*)(sX + (srow + 32 * i) * ROT_LD + scol) = *(const f32x4*)(x + r * ldx + k0 + scol);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (i < nct) *(f32x4*)(sR + (srow + 32 * i) * ROT_LD + scol) = *(const f32x4*)(rmat + (size_t)(srow + 32 * i) * kdim + k0 + scol);
        __syncthreads();
        // lane half h owns k = 16 h .. 16 h + 15 of the chunk; MFMA step s consumes element s of both halves (A and B agree)
        f32x4 fb[4];
        const float* pb = sX + (wv * 32 + r32) * ROT_LD + h * 16;
#pragma unroll
        for (int v = 0; v < 4; ++v) fb[v] = *(const f32x4*)(pb + 4 * v);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= nct) continue;                                                            // uniform
            f32x4 fa[4];
            const float* pa = sR + (t * 32 + r32) * ROT_LD + h * 16;
#pragma unroll
            for (int v = 0; v < 4; ++v) fa[v] = *(const f32x4*)(pa + 4 * v);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[v][e], fb[v][e], acc[t], 0, 0, 0);
        }
    }
    // C layout of the 32x32 tile: column = lane & 31 (descriptor), row = (e & 3) + 8 (e >> 2) + 4 h (rotated coordinate)
    const size_t row = row0 + wv * 32 + r32;
    const size_t n_img = ((size_t)n + 255) / 256 * 256;
    if (row >= n_img) return;
    const bool live = row < (size_t)n;
    const size_t tile = row >> 8; const int r = (int)(row & 255);
    const int fsw = (0x78 >> (2 * ((r >> 2) & 3))) & 3;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= nct) continue;
        u16* base = dst + ((tile * nct + t) * 256 + r) * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            union { _Float16 hf[4]; uint2 u; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk.hf[e] = live ? (_Float16)(acc[t][4 * j + e] * scale) : (_Float16)0.f;
            *(uint2*)(base + ((j ^ fsw) << 3)) = pk.u;
        }
    }
}

// sum of squares of every image row (one wave per row): sumsq[row] = sum h^2 (h = stored f16 value), and the ring kernel's
// pre-scaled C operand cn[row] = -sumsq * cn_factor (= |c^|^2 / out_scale), -inf for the padding rows (they can never win)
__global__ __launch_bounds__(256) void k_f16t_norms(const u16* __restrict__ img, int n_rows_pad, int n_rows, int nk, float cn_factor,
                                                    float* __restrict__ sumsq, float* __restrict__ cn) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows_pad) return;
    const int lane = lane_id();
    const size_t tile = (size_t)(row >> 8); const int r = row & 255;
    float s = 0.f;
    for (int i = lane; i < nk * 32; i += 64) {
        const int kc = i >> 5, j = i & 31;
        const float v = (float)__builtin_bit_cast(_Float16, img[((tile * nk + kc) * 256 + r) * 32 + j]);
        s += v * v;
    }
    s = wave_sum_f(s);
    if (lane == 0) {
        if (sumsq) sumsq[row] = s;
        cn[row] = row < n_rows ? -s * cn_factor : -__builtin_inff();
    }
}

// cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed); on return row i of V is the eigenvector of w[i].
// Accuracy beyond ~1e-9 is irrelevant here: only the orthonormality of the rows matters for the bound, and that is MEASURED.
void jacobi_eig(int n, std::vector<double>& A, std::vector<double>& V, std::vector<double>& w) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (!(off > 1e-18 * diag)) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = A[(size_t)p * n + q];
            if (std::fabs(apq) < 1e-300) continue;
            const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            double* Ap = &A[(size_t)p * n]; double* Aq = &A[(size_t)q * n];
            for (int k = 0; k < n; ++k) { const double x = Ap[k], y = Aq[k]; Ap[k] = c * x - s * y; Aq[k] = s * x + c * y; }
            for (int k = 0; k < n; ++k) { double* r = &A[(size_t)k * n]; const double x = r[p], y = r[q]; r[p] = c * x - s * y; r[q] = s * x + c * y; }
            double* Vp = &V[(size_t)p * n]; double* Vq = &V[(size_t)q * n];
            for (int k = 0; k < n; ++k) { const double x = Vp[k], y = Vq[k]; Vp[k] = c * x - s * y; Vq[k] = s * x + c * y; }
        }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

float f16_scale_of_bound(float bound) {          // power of two s with bound * s in [2^13, 2^14) (clamped to 2^+-40), as knn.hip's f16_scale_for
    uint32_t b; memcpy(&b, &bound, 4);
    const int e = (int)((b >> 23) & 255u);
    if (e == 0 || e == 255) return 1.0f;
    int k = 13 - (e - 127);
    k = k > 40 ? 40 : (k < -40 ? -40 : k);
    const uint32_t u = (uint32_t)(127 + k) << 23;
    float f; memcpy(&f, &u, 4);
    return f;
}

int launch_rotate(ismhip_ctx* ctx, const float* x, int n, int ldx, int kdim, const float* rmat, int m, float scale, u16* dst) {
    const int nct = m / 32;
    const void* kern = nct <= 4 ? (const void*)k_rotate_f16t<4> : nct <= 6 ? (const void*)k_rotate_f16t<6> : (const void*)k_rotate_f16t<8>;
    const int nt = nct <= 4 ? 4 : nct <= 6 ? 6 : 8;
    const size_t lds = (size_t)(128 + nt * 32) * ROT_LD * sizeof(float);
    if (!ctx->attr_done.count(kern)) { ISM_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ctx->attr_done.insert(kern); }
    const unsigned blocks = (unsigned)((((size_t)n + 255) / 256 * 256) / 128);
    void* args[] = {&x, &n, &ldx, &kdim, &rmat, &m, &scale, &dst};
    ISM_HIP(ctx, hipLaunchKernel(kern, dim3(blocks), dim3(256), args, lds, ctx->stream));
    ISM_CHECK_LAUNCH(ctx, "k_rotate_f16t");
    return ISMHIP_OK;
}

}  // namespace

// the query batch -> rotated f16 image (stage 1 of the two-stage search); q rows of ldq >= dim_pad floats, zero beyond dim
int ism_pca_rotate_queries(ismhip_ctx* ctx, const ismhip_codebook* cb, const PcaImage* P, const float* q, int nq, int ldq, unsigned short* dst) {
    return launch_rotate(ctx, q, nq, ldq, cb->dim_pad, P->R, P->m, P->sq, dst);
}

// One image from the leading m rows of the eigenbasis (R: [m x dim_pad] fp32 as uploaded): scales, error constants, the f16 image in the
// ring kernel's layout and its |c^|^2 row. Leaves P.m == 0 (never an error) when the image cannot be trusted.
static int build_image(ismhip_ctx* ctx, ismhip_codebook* cb, PcaImage& P, const std::vector<float>& R, int m, double energy, double trace) {
    const int dp = cb->dim_pad;
    P.m = 0;
    // sigma_max(R)^2 <= 1 + |R R^T - I|_F and |R|_F, from the fp32 values that are uploaded
    double e2 = 0, fro2 = 0;
    for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) {
        double g = 0; for (int c = 0; c < dp; ++c) g += (double)R[(size_t)i * dp + c] * (double)R[(size_t)j * dp + c];
        if (i == j) { fro2 += g; if (g != 0.0) { g -= 1.0; e2 += g * g; } } else e2 += 2.0 * g * g;     // all-zero rows (padding) only shrink the image
    }
    const double sig2 = 1.0 + std::sqrt(e2), sig = std::sqrt(sig2), fro = std::sqrt(fro2);
    if (!(sig2 < 1.01)) return ISMHIP_OK;                                              // a basis this far from orthonormal is a bug, not a bound
    const double cmax = std::sqrt((double)cb->max_norm2);
    P.sc = f16_scale_of_bound((float)(1.001 * sig * cmax));
    P.sq = f16_scale_of_bound((float)(2.002 * sig * cmax));
    const double gamma = 1.01 * dp * 1.1920929e-07;                                    // fp32 rotation: K adds of relative error <= 2^-23, any order
    P.d_rel = (float)(1.001 * (gamma * fro + 4.8828125e-04 * (sig + gamma * fro)));
    P.dq_abs = (float)(1.001 * std::sqrt((double)m) * 6.103515625e-05 / P.sq);
    P.dc_abs = (float)(1.001 * std::sqrt((double)m) * 6.103515625e-05 / P.sc);
    P.inv_sig2 = (float)((1.0 / sig2) * (1.0 - 1e-6));
    P.energy = (float)energy;
    P.resid2 = (float)((1.0 - energy) * trace / (double)cb->n_words);
    const int nk = m / 32, n_tiles = cb->n_words_pad / 256;
    if (hipMalloc((void**)&P.R, R.size() * sizeof(float)) != hipSuccess) return ism_set_err(ctx, ISMHIP_ERR_NOMEM, "codebook rotation matrix");
    ISM_HIP(ctx, hipMemcpy(P.R, R.data(), R.size() * sizeof(float), hipMemcpyHostToDevice));
    if (hipMalloc((void**)&P.f16t, (size_t)n_tiles * nk * 8192 * sizeof(u16)) != hipSuccess) return ism_set_err(ctx, ISMHIP_ERR_NOMEM, "codebook rotated f16 image");
    if (hipMalloc((void**)&P.cn_scaled, ((size_t)cb->n_words_pad + 256) * sizeof(float) + 16) != hipSuccess) return ism_set_err(ctx, ISMHIP_ERR_NOMEM, "codebook rotated norms");
    P.osc = P.cn_scaled + cb->n_words_pad + 256;
    P.m = m;
    int rc = launch_rotate(ctx, cb->words, cb->n_words_pad, dp, dp, P.R, m, P.sc, P.f16t);
    if (rc != ISMHIP_OK) { P.m = 0; return rc; }
    float* sumsq_d = (float*)ism_scratch(ctx, SCR_PCA, (size_t)cb->n_words_pad * sizeof(float));
    if (!sumsq_d) { P.m = 0; return ISMHIP_ERR_NOMEM; }
    ISM_HIP(ctx, hipMemsetAsync(P.cn_scaled, 0, ((size_t)cb->n_words_pad + 256) * sizeof(float), ctx->stream));
    const float cn_factor = P.sq / (2.0f * P.sc);                          // |c^|^2 / out_scale = -(sum h^2 / sc^2) sq sc / 2: powers of two, exact
    hipLaunchKernelGGL(k_f16t_norms, dim3((cb->n_words_pad + 3) / 4), dim3(256), 0, ctx->stream, P.f16t, cb->n_words_pad, cb->n_words, nk, cn_factor, sumsq_d, P.cn_scaled);
    ISM_CHECK_LAUNCH(ctx, "k_f16t_norms");
    std::vector<float> ss(cb->n_words);
    ISM_HIP(ctx, hipMemcpyAsync(ss.data(), sumsq_d, (size_t)cb->n_words * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    const float osc = -2.0f / (P.sq * P.sc);
    ISM_HIP(ctx, hipMemcpyAsync(P.osc, &osc, sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float mx = 0.f;
    for (float v : ss) mx = v > mx || v != v ? v : mx;
    P.cmax2 = mx / (P.sc * P.sc) * 1.00001f;
    if (!(P.cmax2 < 1e30f)) P.m = 0;                                        // overflowed image: keep the original path
    return ISMHIP_OK;
}

// Builds the rotated image of a codebook (called once from ismhip_codebook_create, after the fp32 words and their norms exist).
// Leaves cb->pca.m == 0 when the codebook does not qualify or its spectrum is too flat to pay; never an error for that.
int ism_codebook_build_pca(ismhip_ctx* ctx, ismhip_codebook* cb) {
    cb->pca.m = 0; cb->pca2.m = 0;
    const int dp = cb->dim_pad;
    // (descriptors longer than 512: the host eigen-solve is cubic in the length -- 352 dimensions ~1 s, 1344 tens of seconds -- and the only
    // long descriptor of the path, CSHOT-1344, is searched with chi-square in every shipped configuration; ISMHIP_KNN_PCA_M > 0 overrides)
    if (ctx->knn_pca_m == 0 || !cb->words_f16t || dp < 64 || cb->n_words_pad < 4096 || (dp > 512 && ctx->knn_pca_m < 0)) return ISMHIP_OK;
    if (!(cb->max_norm2 > 0.f) || !(cb->max_norm2 < 1e30f)) return ISMHIP_OK;          // NaN / inf / all-zero codebooks: nothing to gain
    // Short descriptors (FPFH-33: 64 padded dimensions, elements up to 100): nothing to truncate, but the SAME machinery with R = I
    // replaces the exact-f32 MFMA contraction these codebooks otherwise need. The older error model charges the f16 rounding as
    // 2 * 2^-11 |q||c| on the SQUARED distance (|q||c| ~ 2e4 for FPFH: larger than the neighbour distances themselves); here it enters
    // as delta = 2^-11 |x| on the DISTANCE (triangle inequality), ~0.1 against distances of 5-20, and the f16 ring kernel proves
    // nearly every query (configs[4]: the FPFH model's candidate stage 260 -> see DESIGN.md §5 ms per step).
    const bool identity = dp <= 64;
    std::vector<float> R;
    std::vector<double> w_sorted, V_sorted;                                            // eigenvalues (descending) and their vectors, row by row
    int m = 0; double energy = 1.0, trace = 0.0;
    if (identity) {
        m = dp;
        R.assign((size_t)m * dp, 0.f);
        for (int j = 0; j < cb->dim; ++j) R[(size_t)j * dp + j] = 1.f;                 // rows beyond dim stay zero: padded coordinates
        std::vector<float> nh(cb->n_words);
        ISM_HIP(ctx, hipMemcpy(nh.data(), cb->word_norm, (size_t)cb->n_words * 4, hipMemcpyDeviceToHost));
        for (float v : nh) trace += v;
    } else {
    const int nb = dp / 32, n_pairs = nb * (nb + 1) / 2;
    const int n_chunks = std::max(1, std::min(64, cb->n_words / 1024));
    const int rows_per_chunk = (cb->n_words + n_chunks - 1) / n_chunks;
    const size_t pbytes = (size_t)n_chunks * dp * dp * sizeof(float);
    float* part_d = (float*)ism_scratch(ctx, SCR_PCA, pbytes);
    if (!part_d) return ISMHIP_ERR_NOMEM;
    ISM_HIP(ctx, hipMemsetAsync(part_d, 0, pbytes, ctx->stream));
    hipLaunchKernelGGL(k_second_moment, dim3(n_pairs, n_chunks), dim3(256), 0, ctx->stream, cb->words, cb->n_words, dp, rows_per_chunk, nb, part_d);
    ISM_CHECK_LAUNCH(ctx, "k_second_moment");
    std::vector<float> part((size_t)n_chunks * dp * dp);
    ISM_HIP(ctx, hipMemcpyAsync(part.data(), part_d, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> S((size_t)dp * dp, 0.0), V, w;
    for (int c = 0; c < n_chunks; ++c)
        for (int i = 0; i < dp; ++i)
            for (int j = i / 32 * 32; j < dp; ++j) S[(size_t)i * dp + j] += (double)part[((size_t)c * dp + i) * dp + j];
    for (int i = 0; i < dp; ++i) for (int j = i + 1; j < dp; ++j) { const double v = i / 32 == j / 32 ? 0.5 * (S[(size_t)i * dp + j] + S[(size_t)j * dp + i]) : S[(size_t)i * dp + j]; S[(size_t)i * dp + j] = v; S[(size_t)j * dp + i] = v; }
    for (int i = 0; i < dp; ++i) trace += S[(size_t)i * dp + i];
    if (!(trace > 0) || !std::isfinite(trace)) return ISMHIP_OK;
    jacobi_eig(dp, S, V, w);
    std::vector<int> order(dp); std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    w_sorted.resize(dp); V_sorted.resize((size_t)dp * dp);
    for (int j = 0; j < dp; ++j) { w_sorted[j] = w[order[j]]; std::copy(V.begin() + (size_t)order[j] * dp, V.begin() + (size_t)(order[j] + 1) * dp, V_sorted.begin() + (size_t)j * dp); }
    // leading coordinates kept: forced (ISMHIP_KNN_PCA_M), else the smallest multiple of 32 that holds 97 % of the second moment. Bench
    // data (929 792 queries x 102 400 words, measured): 128 of 352 coordinates -> 7.0 % of the queries fail the stage-1 proof and are
    // searched again in all dimensions, kNN 35.6 ms per launch; 160 -> 3.0 %, 36.4 ms; 192 -> 1.0 %, 42.9 ms; 96 -> 14.5 %, 38.6 ms;
    // all 352 (no rotated image) -> 0.1 %, 58.7 ms. If the rule leaves less than 64 dimensions saved, or needs more than the 256 the
    // rotation kernel is built for, the original image stays the stage-1 image.
    const int m_cap = std::min(256, dp);
    double cum = 0; energy = 0;
    if (ctx->knn_pca_m > 0) {
        m = std::max(64, std::min(m_cap, (ctx->knn_pca_m + 31) / 32 * 32));
        for (int i = 0; i < m; ++i) energy += w[order[i]];
        energy /= trace;
    } else {
        // (at least 64: the ring kernel prefetches four slices ahead and keeps the |c|^2 rows of four tiles, i.e. it needs >= 2 slices per tile)
        for (int i = 0; i < dp && !m; ++i) { cum += w[order[i]]; if ((i + 1) % 32 == 0 && i + 1 >= 64 && cum >= 0.97 * trace) { m = i + 1; energy = cum / trace; } }
        if (m == 0 || m > m_cap || m + 64 > dp) return ISMHIP_OK;
    }
    R.assign((size_t)m * dp, 0.f);
    for (int j = 0; j < m; ++j) for (int c = 0; c < dp; ++c) R[(size_t)j * dp + c] = c < cb->dim ? (float)V[(size_t)order[j] * dp + c] : 0.f;
    }   // principal axes / identity
    int rc = build_image(ctx, cb, cb->pca, R, m, energy, trace);
    if (rc != ISMHIP_OK || cb->pca.m == 0 || identity || ctx->knn_pca_m2 == 0) return rc;
    // Stage 2 (the queries whose stage-1 proof failed) on a LONGER prefix of the same basis instead of all dimensions: the smallest
    // multiple of 32 beyond m + 32 that holds 99.7 % of the second moment (bench data: 256 of 352 coordinates, 8 slices per tile
    // instead of 11), if that still saves 64 dimensions; what it cannot prove goes to the exact scan as before.
    int m2 = 0; double cum2 = 0, energy2 = 0;
    if (ctx->knn_pca_m2 > 0) m2 = std::max(m + 32, std::min(std::min(256, dp), (ctx->knn_pca_m2 + 31) / 32 * 32));
    for (int i = 0; i < dp; ++i) {
        cum2 += w_sorted[i];
        if (m2 == 0 && (i + 1) % 32 == 0 && i + 1 >= m + 64 && cum2 >= 0.997 * trace) m2 = i + 1;
        if (m2 && i + 1 == m2) energy2 = cum2 / trace;
    }
    if (m2 == 0 || m2 > std::min(256, dp) || (m2 + 64 > dp && ctx->knn_pca_m2 < 0)) return ISMHIP_OK;
    std::vector<float> R2((size_t)m2 * dp, 0.f);
    for (int j = 0; j < m2; ++j) for (int c2 = 0; c2 < dp; ++c2) R2[(size_t)j * dp + c2] = c2 < cb->dim ? (float)V_sorted[(size_t)j * dp + c2] : 0.f;
    return build_image(ctx, cb, cb->pca2, R2, m2, energy2, trace);
}

void ism_codebook_free_pca(ismhip_codebook* cb) {
    for (PcaImage* P : {&cb->pca, &cb->pca2}) {
        if (P->R) (void)hipFree(P->R);
        if (P->f16t) (void)hipFree(P->f16t);
        if (P->cn_scaled) (void)hipFree(P->cn_scaled);
        *P = PcaImage();
    }
}
