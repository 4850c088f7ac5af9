// knn.hip — exact k-nearest-codeword search (codebook activation).
// Reference seam: ActivationStrategyKNN::activateKNN (activation_strategy/activation_strategy_knn.h:41-126) with
// FLANNExactMatch semantics (flann::SearchParams(-1)); distance functors utils/distance.h:45,65 (FLANN L2 = squared
// Euclidean, ChiSquareDistance), SURVEY Appendix A.6.
//
// Three stages:
//  1. candidate generation (the dominant kernel of the whole path). Score(c,q) = |c|^2 - 2 c.q as a dense [codewords x queries]
//     contraction on the matrix cores; codewords are the MFMA ROWS and queries the COLUMNS, so a lane holds 16 codeword scores of
//     ONE query per accumulator tile: the running top-T per query lives in registers with no cross-lane traffic and the Nq x Nc
//     matrix is never materialised. Every kernel keeps, per lane slot, the T best candidates AND the value of the best candidate
//     it dropped (the slot's bound).
//       k_knn_l2_ring    f16 MFMA, 256x256 tile, LDS-DMA ring          default for launches >= 4096 queries x 4096 codewords
//       k_knn_l2_mfma16  f16 (or bf16x3) MFMA, register-staged tiles   smaller launches; ISMHIP_KNN_MODE=bf16x3 for A/B runs
//       k_knn_l2_mfma    f32 MFMA (exact fma chain)                    ISMHIP_KNN_MODE=f32: the independent route used by the tests
//       k_knn_chi2       VALU 64x64 tile, v_rcp_f32                    chi-square is not a contraction
//  2. k_knn_rerank — one wave per query: candidates that can still matter are recomputed with the FLANN functor's own summation
//     order (bit-identical to the CPU functor), the k smallest (distance, row) pairs are selected (ties: lowest row), and the
//     result is PROVEN slot by slot from the slot bounds and a rigorous bound of the candidate kernel's error; a (query, slot) pair
//     that cannot be proven is queued for
//  3. k_knn_fallback / k_knn_fallback_merge — exact scan of the queued slots' rows. Rare for descriptor data, the rule for
//     adversarial inputs (un-normalised magnitudes, hundreds of near-duplicates): it keeps the answer exact in every case.
#include "common.h"

int ism_pca_rotate_queries(ismhip_ctx* ctx, const ismhip_codebook* cb, const PcaImage* P, const float* q, int nq, int ldq, unsigned short* dst);   // pca.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KNN_BM 128       // codeword rows per tile
#define KNN_BN 128       // queries per tile
#define KNN_BK 32
#define KNN_LDK 36       // padded row stride (floats) of the LDS tiles: 144 B keeps 16-B alignment, spreads banks

template <int T>
struct TopT {
    float v[T]; int i[T];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < T; ++t) { v[t] = __builtin_inff(); i[t] = -1; }
    }
    // the same insertion without a branch (k_knn_l2_ring16's epilogue: a wave enters it when ANY lane has a score to insert, and
    // nested exec-mask branches cost more than thirteen predicated instructions); x = +inf leaves the list as it is
    __device__ __forceinline__ void push_flat(float x, int idx) {
        bool c[T];
#pragma unroll
        for (int t = 0; t < T; ++t) c[t] = x < v[t];
#pragma unroll
        for (int t = T - 1; t > 0; --t) {
            v[t] = c[t - 1] ? v[t - 1] : (c[t] ? x : v[t]);
            i[t] = c[t - 1] ? i[t - 1] : (c[t] ? idx : i[t]);
        }
        v[0] = c[0] ? x : v[0];
        i[0] = c[0] ? idx : i[0];
    }
    // insert keeping ascending order; strict < keeps the earlier (lower row) on ties
    __device__ __forceinline__ void push(float x, int idx) {
        if (!(x < v[T - 1])) return;
        v[T - 1] = x; i[T - 1] = idx;
#pragma unroll
        for (int t = T - 1; t > 0; --t) {
            if (v[t] < v[t - 1]) {
                float tv = v[t]; v[t] = v[t - 1]; v[t - 1] = tv;
                int ti = i[t]; i[t] = i[t - 1]; i[t - 1] = ti;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// L2 candidates on the FP32 matrix cores
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256, 2) void k_knn_l2_mfma(const float* __restrict__ words, const float* __restrict__ word_norm,
                                                        int n_tiles_m, int dim_pad,
                                                        const float* __restrict__ q, int nq, int ldq,
                                                        int tiles_per_split, int n_splits,
                                                        float* __restrict__ cand_val, int* __restrict__ cand_idx, int cand_stride,
                                                        float* __restrict__ cand_bound, int bound_stride, int last_steps) {
    __shared__ __attribute__((aligned(16))) float sA[2][KNN_BM * KNN_LDK];
    __shared__ __attribute__((aligned(16))) float sB[2][KNN_BN * KNN_LDK];
    __shared__ float sCn[KNN_BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware block -> (query tile, codebook split) map. Blocks are dealt round-robin over the 8 XCDs, each with a private
    // 4 MiB L2: block id = 8*j + x runs on XCD group x and takes query tile 8*(j / n_splits) + x, split j % n_splits, so the
    // blocks co-resident on one XCD cover few query tiles (their hi/lo images stay in that L2 while every split's codeword
    // slices stream through it) instead of 32 different ones that thrash it. Placement only affects speed, never results.
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = jx % n_splits, qtile = (jx / n_splits) * 8 + xcd;
    if (qtile * KNN_BN >= nq) return;
    const int mt0 = split * tiles_per_split;
    const int mt1 = min(n_tiles_m, mt0 + tiles_per_split);
    const int nk = dim_pad / KNN_BK;

    // staging map: thread -> (row = tid/8 + 32*i, float4 column = tid%8)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const float* qbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int qr = qtile * KNN_BN + srow + 32 * i;
        qr = qr < nq ? qr : nq - 1;                       // clamp: duplicates are never written back
        qbase[i] = q + (size_t)qr * ldq + scol;
    }

    TopT<T + 1> top[2];            // T candidates + the best value that gets dropped
    top[0].init(); top[1].init();

    for (int mt = mt0; mt < mt1; ++mt) {
        const float* abase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) abase[i] = words + (size_t)(mt * KNN_BM + srow + 32 * i) * dim_pad + scol;

        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

        f32x4 ga[4], gb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ga[i] = *(const f32x4*)(abase[i]); gb[i] = *(const f32x4*)(qbase[i]); }
        __syncthreads();                                   // previous tile's epilogue has finished reading sCn / LDS
        if (tid < KNN_BM) sCn[tid] = word_norm[mt * KNN_BM + tid];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4*)(&sA[0][(srow + 32 * i) * KNN_LDK + scol]) = ga[i];
            *(f32x4*)(&sB[0][(srow + 32 * i) * KNN_LDK + scol]) = gb[i];
        }
        __syncthreads();

        for (int kc = 0; kc < nk; ++kc) {
            const int cur = kc & 1;
            if (kc + 1 < nk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[i] = *(const f32x4*)(abase[i] + (kc + 1) * KNN_BK);
                    gb[i] = *(const f32x4*)(qbase[i] + (kc + 1) * KNN_BK);
                }
            }
            // operand fragments: lane half h owns k = 16h .. 16h+15 of the slice (any pairing of k is valid as long as
            // A and B agree); step s of the 32x32x2 MFMA consumes element s of both halves.
            f32x4 fa[2][4], fb[2][4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float* p = &sA[cur][(wr * 64 + mi * 32 + r) * KNN_LDK + h * 16];
#pragma unroll
                for (int v = 0; v < 4; ++v) fa[mi][v] = *(const f32x4*)(p + v * 4);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float* p = &sB[cur][(wc * 64 + ni * 32 + r) * KNN_LDK + h * 16];
#pragma unroll
                for (int v = 0; v < 4; ++v) fb[ni][v] = *(const f32x4*)(p + v * 4);
            }
            // the last slice holds dim - 32 (nk - 1) real columns, the rest is zero padding: MFMA step s covers columns s and 16 + s,
            // so only the first last_steps steps carry anything (FPFH-33: 1 of 16 -- the padded steps were 47 % of this kernel's MFMAs)
            const int steps = kc + 1 < nk ? 16 : last_steps;
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * v + e >= steps) continue;                 // uniform
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][v][e], fb[ni][v][e], acc[mi][ni], 0, 0, 0);
                }
            if (kc + 1 < nk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *(f32x4*)(&sA[cur ^ 1][(srow + 32 * i) * KNN_LDK + scol]) = ga[i];
                    *(f32x4*)(&sB[cur ^ 1][(srow + 32 * i) * KNN_LDK + scol]) = gb[i];
                }
            }
            __syncthreads();
        }
        // epilogue: C/D layout of the 32x32 tile: col = lane&31 (query), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (codeword)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float cn[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) cn[e] = sCn[wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float tau = top[ni].v[T];
                bool any = false;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[mi][ni][e] = cn[e] - 2.0f * acc[mi][ni][e]; any |= acc[mi][ni][e] < tau; }
                if (__any(any)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) top[ni].push(acc[mi][ni][e], mt * KNN_BM + wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h);
                }
            }
        }
    }
    // candidates: slot = split*(4T) + (wr*2 + h)*T + t
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int qi = qtile * KNN_BN + wc * 64 + ni * 32 + r;
        if (qi < nq) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const size_t o = (size_t)qi * cand_stride + split * (4 * T) + (wr * 2 + h) * T + t;
                cand_val[o] = top[ni].v[t]; cand_idx[o] = top[ni].i[t];
            }
            cand_bound[(size_t)qi * bound_stride + split * 4 + (wr * 2 + h)] = top[ni].v[T];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// L2 candidates on the BF16 matrix cores with a 3-term split (q = qh + ql, c = ch + cl, dot ~ qh.ch + qh.cl + ql.ch)
// ---------------------------------------------------------------------------------------------
// The candidate stage only has to be accurate enough for the proof in k_knn_rerank to go through; its result is never
// returned. bf16 keeps fp32's exponent range (no underflow of the residuals) and v_mfma_f32_32x32x16_bf16 runs at 16x the
// rate of the f32-input MFMA, so three of them per 16-k step are ~5x cheaper than the exact-f32 contraction.
// Error bound used by the proof (VerifyParams::dot_rel, relative to |q||c|):
//   representation : |q - qh - ql| <= 2^-16 |q| element-wise (two RN-to-bf16 steps, 8 significant bits: u = 2^-8), the dropped
//                    ql.cl term and the two residual cross terms give <= 3.1 * 2^-16
//   accumulation   : products of bf16 pairs are exact in fp32; the 3K-term sum is modelled as fp32 additions in ANY order
//                    with a per-add unit roundoff of 2^-23 (i.e. not even assuming round-to-nearest inside the MFMA)
//                    -> 1.01 * 3K * 2^-23
// tests/test_gpu_parity.py::test_knn_bf16x3_error_model checks the measured error against this model on random and
// adversarial (all-positive, large-norm) data.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f32_to_bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// rows beyond n and columns beyond dim are zero; hi = RN_bf16(x), lo = RN_bf16(x - hi)
__global__ void k_split_bf16(const float* __restrict__ src, int n, int dim, int ld, int n_pad, int dim_pad,
                             u16* __restrict__ hi, u16* __restrict__ lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_pad * dim_pad) return;
    const int row = (int)(i / dim_pad), col = (int)(i % dim_pad);
    float x = 0.f;
    if (row < n && col < dim) x = src[(size_t)row * ld + col];
    const u16 h = f32_to_bf16_rn(x);
    const float hf = __uint_as_float((unsigned)h << 16);
    hi[i] = h; lo[i] = f32_to_bf16_rn(x - hf);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- f16 image (NTERM = 1). x -> RN_f16(x * s), s a power of two that puts the largest |element| into [2^13, 2^14) (clamped to
// 2^+-40), so neither overflow nor the fp16 subnormal range matters: element error <= 2^-11 |x| + 2^-14 / s, where the second
// term assumes the worst (subnormal results flushed to zero). absmax is kept as float bits (non-negative floats order like uints;
// NaN/inf sort last and select s = 1, the affected scores become NaN/inf and those queries take the exact path).
__global__ void k_absmax(const float* __restrict__ src, int n, int dim, int ld, uint32_t* __restrict__ out_bits) {
    uint32_t m = 0u;
    const size_t tot = (size_t)n * dim;
    if (ld == dim) {                                   // contiguous rows: a flat, 16-byte-wide sweep
        const size_t n4 = tot >> 2;
        const uint4* s4 = (const uint4*)src;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = s4[i];
            const uint32_t a = max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu));
            m = a > m ? a : m;
        }
        for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
            const uint32_t b = __float_as_uint(src[i]) & 0x7fffffffu; m = b > m ? b : m;
        }
    } else
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / dim), col = (int)(i % dim);
        const uint32_t b = __float_as_uint(src[(size_t)row * ld + col]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    __shared__ uint32_t s_m[4];                        // one atomic per workgroup: thousands of same-address atomics serialise
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(out_bits, m);
    }
}
__host__ __device__ inline float f16_scale_for(uint32_t absmax_bits) {
    const int e = (int)(absmax_bits >> 23);          // biased exponent; 0 = zero/subnormal, 255 = inf/NaN
    if (e == 0 || e == 255) return 1.0f;
    int k = 13 - (e - 127);                          // absmax * 2^k in [2^13, 2^14)
    k = k > 40 ? 40 : (k < -40 ? -40 : k);
    union { uint32_t u; float f; } v; v.u = (uint32_t)(127 + k) << 23;
    return v.f;
}
// sc[0] = absmax bits (in), sc[1] = -2 / (s * other_scale) (out), sc[2] = 2^-14 / s (out: worst-case absolute element error)
__global__ void k_to_f16(const float* __restrict__ src, int n, int dim, int ld, int n_pad, int dim_pad,
                         uint32_t* __restrict__ sc, float other_scale, u16* __restrict__ dst) {
    const float s = f16_scale_for(sc[0]);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ((float*)sc)[1] = -2.0f / (s * other_scale);
        ((float*)sc)[2] = (sc[0] >> 23) == 255u ? __builtin_inff() : 6.103515625e-05f / s;     // inf/NaN in the batch: nothing is provable
    }
    if (i >= (size_t)n_pad * dim_pad) return;
    const int row = (int)(i / dim_pad), col = (int)(i % dim_pad);
    float x = 0.f;
    if (row < n && col < dim) x = src[(size_t)row * ld + col];
    const _Float16 hx = (_Float16)(x * s);            // v_cvt_f16_f32, round to nearest even
    dst[i] = __builtin_bit_cast(u16, hx);
}

// The same conversion into the layout k_knn_l2_ring / k_knn_l2_ring16 stream: [256-row tile][32-k slice][row][4 x 16-byte
// segments], i.e. every (tile, slice) is one contiguous 16 KB block that already is the LDS image (segment p of row r holds
// logical segment p ^ F[(r>>2)&3], F = {0,2,3,1}). A DMA instruction then copies 1 KB of consecutive, fully used 128-byte lines; with a row-major image each
// 32-k slice touches only half of every line and the other half is fetched again one slice later.
__global__ void k_to_f16_tiled(const float* __restrict__ src, int n, int dim, int ld, int n_tiles, int nk,
                               uint32_t* __restrict__ sc, float other_scale, u16* __restrict__ dst) {
    const float s = f16_scale_for(sc[0]);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ((float*)sc)[1] = -2.0f / (s * other_scale);
        ((float*)sc)[2] = (sc[0] >> 23) == 255u ? __builtin_inff() : 6.103515625e-05f / s;
    }
    if (i >= (size_t)n_tiles * nk * 8192) return;
    const int e = (int)(i & 7), p = (int)((i >> 3) & 3), r = (int)((i >> 5) & 255);
    const size_t blk = i >> 13;
    const int kc = (int)(blk % nk); const size_t tile = blk / nk;
    const int col = kc * 32 + ((p ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) << 3) + e;      // F = {0,2,3,1}[(r>>2)&3]: conflict-free for both MFMA shapes
    const size_t row = tile * 256 + r;
    float x = 0.f;
    if (row < (size_t)n && col < dim) x = src[row * ld + col];
    const _Float16 hx = (_Float16)(x * s);
    dst[i] = __builtin_bit_cast(u16, hx);
}

// Tile geometry is a template: WR x WC waves, each MI x NI MFMA tiles of 32x32 -> BM = WR*MI*32 codeword rows by
// BN = WC*NI*32 queries per workgroup. The CU's load path delivers ~30 B/clk from L2 (MI355X_MICROARCH 'Indexed rows'), a
// 128x128 tile needs 32 KB per 32-k slice for 768 MFMA cycles per wave and is load-bound; the 256x256 tile (8 waves, 64 KB per
// slice for 1536 MFMA cycles per wave, 128 KB of LDS, one workgroup per CU) is MFMA-bound.
//
// NTERM = 3: bf16x3 (hi/lo images, three MFMAs per product, |error| ~ 2^-16 |q||c|).
// NTERM = 1: f16 (one fp16 image scaled by a power of two so that the largest element sits in [2^13, 2^14), ONE MFMA per
//            product, |error| ~ 2^-11 |q||c|). The scores only have to RANK the codewords well enough for the top-T slots to
//            hold the true neighbours; k_knn_rerank recomputes every surviving candidate with the exact functor and proves the
//            result with the rigorous bound of this kernel's error, so the answer stays exact at a third of the MFMA work.
//            out_scale = -2 / (codebook scale * query scale) is read from device memory (the query scale is found on device).
// KB = halves per LDS row = k-depth of one staged slice (32 or 64). A 64-deep slice moves whole 128-byte lines per codeword /
// query row: with 32-deep slices every line is fetched twice (the halves are used one slice apart and a slice's lines exceed L1).
// 16-byte segments of a row are XOR-swizzled with row bits so that both the staging stores and the fragment reads (32 rows x one
// segment per half-wave) are bank-conflict free: 64-B rows by (row>>2)&3, 128-B rows by (row>>1)&7.
// ld = row stride (halves) of the 16-bit images, a multiple of 64 (zero padded); k_steps = ceil(dim / 16) MFMA k-steps carry data.
// EMIT = 1: no candidate lists; every row whose score is <= emit_tau[query] is appended to emit_list[query * emit_cap ...] (count in
// emit_cnt[query], which may exceed emit_cap: the caller checks). Used by the chi-square search for the queries whose Hellinger
// proof failed: with tau derived from the best chi-square value already found, the emitted rows are ALL rows that can still beat it.
template <int T, int WR, int WC, int MI, int NI, int NTERM, int KB, int EMIT = 0>
__global__ __launch_bounds__(WR * WC * 64, 2) void k_knn_l2_mfma16(const u16* __restrict__ wh, const u16* __restrict__ wl,
                                                          const float* __restrict__ word_norm, int n_tiles_m, int ld, int k_steps,
                                                          const u16* __restrict__ qh, const u16* __restrict__ ql, int nq,
                                                          const float* __restrict__ out_scale,
                                                          int tiles_per_split, int n_splits,
                                                          float* __restrict__ cand_val, int* __restrict__ cand_idx, int cand_stride,
                                                          float* __restrict__ cand_bound, int bound_stride,
                                                          const float* __restrict__ emit_tau, uint32_t* __restrict__ emit_cnt, uint32_t* __restrict__ emit_list, int emit_cap) {
    constexpr int BM = WR * MI * 32, BN = WC * NI * 32, NT = WR * WC * 64;
    constexpr int SEGS = KB / 8;                      // 16-byte segments per row
    constexpr int KS = KB / 16;                       // MFMA k-steps per slice
    constexpr int RPP = NT / SEGS;                    // rows staged per pass (SEGS threads x 16 B per row)
    constexpr int PA = BM / RPP, PB = BN / RPP;       // passes per array
    constexpr int SW_SH = KB == 64 ? 1 : 2, SW_MASK = SEGS - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_smem[];
    constexpr bool X3 = NTERM == 3;
    u16* sAh = (u16*)knn_smem;                        // [2][BM*KB]
    u16* sAl = sAh + (X3 ? 2 * BM * KB : 0);
    u16* sBh = sAl + 2 * BM * KB;                     // [2][BN*KB]
    u16* sBl = sBh + (X3 ? 2 * BN * KB : 0);
    float* sCn = (float*)(sBl + 2 * BN * KB);         // [BM]
    const float oscale = NTERM == 1 ? out_scale[0] : -2.0f;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wr = wv / WC, wc = wv % WC;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware block -> (query tile, codebook split) map, see k_knn_l2_mfma
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = jx % n_splits, qtile = (jx / n_splits) * 8 + xcd;
    if (qtile * BN >= nq) return;
    const int mt0 = split * tiles_per_split;
    const int mt1 = min(n_tiles_m, mt0 + tiles_per_split);
    const int nk = (k_steps + KS - 1) / KS;

    // staging: thread -> (row srow + p*RPP, segment sseg); RPP is a multiple of 16, so the swizzle term is the same for every pass
    const int srow = tid / SEGS, sseg = tid % SEGS;
    const int sdst0 = srow * KB + ((sseg ^ ((srow >> SW_SH) & SW_MASK)) << 3);
    const size_t qoff = (size_t)(qtile * BN + srow) * ld + sseg * 8;
    // fragment reads: lane -> row r of a 32-row MFMA tile, k-segment (ks*2 + h) of the slice; tile bases are compile-time offsets
    const int fragA = (wr * (MI * 32) + r) * KB, fragB = (wc * (NI * 32) + r) * KB;
    const int fsw = (r >> SW_SH) & SW_MASK;

    TopT<T + 1> top[NI];
    float tau[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        top[n].init();
        tau[n] = -__builtin_inff();
        if (EMIT) { const int qi_ = qtile * BN + wc * (NI * 32) + n * 32 + r; if (qi_ < nq) tau[n] = emit_tau[qi_]; }
    }

    for (int mt = mt0; mt < mt1; ++mt) {
        const size_t aoff = (size_t)(mt * BM + srow) * ld + sseg * 8;
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

        f32x4 gah[PA], gal[X3 ? PA : 1], gbh[PB], gbl[X3 ? PB : 1];
#pragma unroll
        for (int p = 0; p < PA; ++p) { gah[p] = *(const f32x4*)(wh + aoff + (size_t)p * RPP * ld); if constexpr (X3) gal[p] = *(const f32x4*)(wl + aoff + (size_t)p * RPP * ld); }
#pragma unroll
        for (int p = 0; p < PB; ++p) { gbh[p] = *(const f32x4*)(qh + qoff + (size_t)p * RPP * ld); if constexpr (X3) gbl[p] = *(const f32x4*)(ql + qoff + (size_t)p * RPP * ld); }
        __syncthreads();                                   // previous tile's epilogue has finished reading sCn / LDS
        for (int i = tid; i < BM; i += NT) sCn[i] = word_norm[mt * BM + i];
#pragma unroll
        for (int p = 0; p < PA; ++p) { *(f32x4*)(&sAh[sdst0 + p * RPP * KB]) = gah[p]; if constexpr (X3) *(f32x4*)(&sAl[sdst0 + p * RPP * KB]) = gal[p]; }
#pragma unroll
        for (int p = 0; p < PB; ++p) { *(f32x4*)(&sBh[sdst0 + p * RPP * KB]) = gbh[p]; if constexpr (X3) *(f32x4*)(&sBl[sdst0 + p * RPP * KB]) = gbl[p]; }
        __syncthreads();

        for (int kc = 0; kc < nk; ++kc) {
            const int cur = kc & 1;
            if (kc + 1 < nk) {
                const int ko = (kc + 1) * KB;
#pragma unroll
                for (int p = 0; p < PA; ++p) { gah[p] = *(const f32x4*)(wh + aoff + (size_t)p * RPP * ld + ko); if constexpr (X3) gal[p] = *(const f32x4*)(wl + aoff + (size_t)p * RPP * ld + ko); }
#pragma unroll
                for (int p = 0; p < PB; ++p) { gbh[p] = *(const f32x4*)(qh + qoff + (size_t)p * RPP * ld + ko); if constexpr (X3) gbl[p] = *(const f32x4*)(ql + qoff + (size_t)p * RPP * ld + ko); }
            }
            const u16* cAh = sAh + cur * BM * KB + fragA; const u16* cAl = sAl + cur * BM * KB + fragA;
            const u16* cBh = sBh + cur * BN * KB + fragB; const u16* cBl = sBl + cur * BN * KB + fragB;
            const int ks_n = min(KS, k_steps - kc * KS);   // the last slice may be partly padding: skip its all-zero k-steps
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < ks_n) {
                    const int so = ((ks * 2 + h) ^ fsw) << 3;
                    if constexpr (X3) {
                        bf16x8 bh[NI], bl[NI];
#pragma unroll
                        for (int n = 0; n < NI; ++n) { bh[n] = *(const bf16x8*)(cBh + n * 32 * KB + so); bl[n] = *(const bf16x8*)(cBl + n * 32 * KB + so); }
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const bf16x8 ah = *(const bf16x8*)(cAh + mi * 32 * KB + so);
                            const bf16x8 al = *(const bf16x8*)(cAl + mi * 32 * KB + so);
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ni], acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
                            }
                        }
                    } else {
                        f16x8 bh[NI];
#pragma unroll
                        for (int n = 0; n < NI; ++n) bh[n] = *(const f16x8*)(cBh + n * 32 * KB + so);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const f16x8 ah = *(const f16x8*)(cAh + mi * 32 * KB + so);
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
                        }
                    }
                }
            }
            if (kc + 1 < nk) {
                const int nx = cur ^ 1;
#pragma unroll
                for (int p = 0; p < PA; ++p) { *(f32x4*)(&sAh[nx * BM * KB + sdst0 + p * RPP * KB]) = gah[p]; if constexpr (X3) *(f32x4*)(&sAl[nx * BM * KB + sdst0 + p * RPP * KB]) = gal[p]; }
#pragma unroll
                for (int p = 0; p < PB; ++p) { *(f32x4*)(&sBh[nx * BN * KB + sdst0 + p * RPP * KB]) = gbh[p]; if constexpr (X3) *(f32x4*)(&sBl[nx * BN * KB + sdst0 + p * RPP * KB]) = gbl[p]; }
            }
            __syncthreads();
        }
        // epilogue: score = |c|^2 - 2 c.q. After the first tiles almost no score beats a lane's current T-th best, so the scores
        // are first only compared (2 VALU per value); the insertion code runs for an accumulator tile only if some lane needs it.
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float cn[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) cn[e] = sCn[wr * (MI * 32) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if (EMIT) {
                    bool any = false;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc[mi][ni][e] = cn[e] + oscale * acc[mi][ni][e]; any |= acc[mi][ni][e] <= tau[ni]; }
                    if (__any(any)) {
                        const int qi_ = qtile * BN + wc * (NI * 32) + ni * 32 + r;
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (acc[mi][ni][e] <= tau[ni]) {
                                const int row_ = mt * BM + wr * (MI * 32) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (row_ < cand_stride) {                // EMIT: cand_stride = number of real rows (padding rows score +inf, tau may be +inf too)
                                    const uint32_t slot = atomicAdd(&emit_cnt[qi_], 1u);
                                    if (slot < (uint32_t)emit_cap) emit_list[(size_t)qi_ * emit_cap + slot] = (uint32_t)row_;
                                }
                            }
                    }
                    continue;
                }
                const float tau_ = top[ni].v[T];
                bool any = false;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[mi][ni][e] = cn[e] + oscale * acc[mi][ni][e]; any |= acc[mi][ni][e] < tau_; }
                if (__any(any)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        top[ni].push(acc[mi][ni][e], mt * BM + wr * (MI * 32) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h);
                }
            }
        }
    }
    if (EMIT) return;
    // candidates: slot = split*(2*WR*T) + (wr*2 + h)*T + t; bound slot = split*(2*WR) + wr*2 + h
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int qi = qtile * BN + wc * (NI * 32) + ni * 32 + r;
        if (qi < nq) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const size_t o = (size_t)qi * cand_stride + split * (2 * WR * T) + (wr * 2 + h) * T + t;
                cand_val[o] = top[ni].v[t]; cand_idx[o] = top[ni].i[t];
            }
            cand_bound[(size_t)qi * bound_stride + split * (2 * WR) + (wr * 2 + h)] = top[ni].v[T];
        }
    }
}

// |c|^2 / out_scale for every codebook row (out_scale is known on the device only: it holds the batch's query scale)
__global__ void k_scale_norms(const float* __restrict__ norm, int n, const float* __restrict__ out_scale, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = norm[i] * (1.0f / out_scale[0]);
}

// ---------------------------------------------------------------------------------------------
// f16 candidates, LDS-DMA ring (the default squared-L2 kernel for big launches)
// ---------------------------------------------------------------------------------------------
// k_knn_l2_mfma16 moves every slice global -> VGPR -> LDS between two barriers, so load latency, the staging stores, the
// fragment reads and the MFMAs of a workgroup run one after the other (measured: 29 % MFMA-busy at 20 GB/s per CU). Here the
// slices are written straight into LDS by global_load_lds_dwordx4 (no staging registers, no ds_write), three slices ahead of the
// one being multiplied, in a ring of four 32 KB stages; the prefetch stream runs across codeword tiles, so it also covers the
// top-T epilogue. One barrier per slice: "my DMAs for slice g have landed" (s_waitcnt vmcnt) + s_barrier makes slice g visible
// to all waves and proves that everybody is done reading the stage that the next DMA overwrites.
//   tile 256 codewords x 256 queries, 8 waves (2 x 4), wave = 128 x 64 = 4 x 2 MFMA tiles of 32x32x16 f16
//   stage: rows 0..255 = codeword slice, 256..511 = query slice, 64 B per row (32 k), 16-B segments XOR-swizzled by (row>>2)&3;
//          a DMA instruction fills 1 KB = 16 rows in LDS order, so the swizzle is applied to the SOURCE address of each lane
//   |c|^2 of a tile arrives the same way (one 1 KB DMA by wave 0) in a ring of four tiles
#define RG_BM 256
#define RG_BN 256
#define RG_KB 32
#define RG_STAGES 4
#define RG_STAGE_HALVES ((RG_BM + RG_BN) * RG_KB)
__device__ __forceinline__ void lds_dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma16_nt(const void* g, void* l) {      // non-temporal: see MI355X_MICROARCH 'nt-weights'
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 2);
}
template <int T, int DBG = 0>
__global__ __launch_bounds__(512, 2) void k_knn_l2_ring(const u16* __restrict__ wh, const float* __restrict__ word_norm, int n_tiles_m, int ld, int k_steps,
                                                        const u16* __restrict__ qh, int nq, const float* __restrict__ out_scale,
                                                        int tiles_per_split, int n_splits,
                                                        float* __restrict__ cand_val, int* __restrict__ cand_idx, int cand_stride,
                                                        float* __restrict__ cand_bound, int bound_stride) {
    constexpr int WR = 2, WC = 4, MI = 4, NI = 2, KB = RG_KB, BM = RG_BM, BN = RG_BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_smem[];
    u16* ring = (u16*)knn_smem;                                        // [RG_STAGES][512 rows][32 halves]
    float* sCn = (float*)(ring + RG_STAGES * RG_STAGE_HALVES);        // [4][BM]
    float* sThr = sCn + 4 * BM;                                        // [8 waves][NI][64]: thresholds published to the partner wave
    const float oscale = out_scale[0];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform values must live in SGPRs (DMA bases, ring pointers)
    const int wr = wv / WC, wc = wv % WC;
    const int r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = jx % n_splits, qtile = (jx / n_splits) * 8 + xcd;
    if (qtile * BN >= nq) return;
    const int mt0 = split * tiles_per_split;
    const int n_t = min(n_tiles_m, mt0 + tiles_per_split) - mt0;
    if (n_t <= 0) return;
    const int nk = (k_steps + 1) / 2;                                  // 32-deep slices
    const int G = n_t * nk;                                            // slices in this workgroup's stream

    // DMA role of this wave: waves 0-3 bring codeword rows, 4-7 query rows; 4 instructions x 16 rows per slice. The address of a
    // lane is a wave-uniform 64-bit base (tile, slice, instruction: scalar arithmetic) plus a per-lane byte offset that never changes.
    // (the 16-bit images are stored tile by tile, slice by slice, already swizzled: see k_to_f16_tiled -- a slice of a tile is a
    // linear 16 KB copy; ld is unused here)
    const bool dma_a = wv < 4;
    const unsigned lane_off = (unsigned)(lane * 16);                   // bytes
    const char* dbase = (dma_a ? (const char*)(wh + (size_t)mt0 * nk * (BM * KB)) : (const char*)(qh + (size_t)qtile * nk * (BN * KB)))
                        + (wv & 3) * (64 * KB * 2);
    const int ddst = (dma_a ? 0 : BM * KB) + (wv & 3) * 64 * KB;       // halves, wave-uniform
    const size_t tile_stride = dma_a ? (size_t)nk * (BM * KB * 2) : 0; // bytes
    int pt = 0, pkc = 0, pg = 0;                                       // prefetch cursor (tile, slice, stream index), clamped at the end
    auto issue = [&]() {
        u16* st = ring + (pg & (RG_STAGES - 1)) * RG_STAGE_HALVES + ddst;
        if (wv == 0 && pkc == 0) lds_dma16(word_norm + (size_t)(mt0 + pt) * BM + lane * 4, sCn + (pt & 3) * BM);
        const char* sp = ((DBG & 512) ? (dma_a ? (const char*)wh : (const char*)qh) + (wv & 3) * (64 * KB * 2) : dbase + pt * tile_stride) + (size_t)pkc * (BM * KB * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((DBG & 64) && dma_a) lds_dma16_nt(sp + j * (16 * KB * 2) + lane_off, st + j * 16 * KB);
            else if ((DBG & 128) && !dma_a) lds_dma16_nt(sp + j * (16 * KB * 2) + lane_off, st + j * 16 * KB);
            else lds_dma16(sp + j * (16 * KB * 2) + lane_off, st + j * 16 * KB);
        }
        ++pg;
        if (pt * nk + pkc + 1 < G) { if (++pkc == nk) { pkc = 0; ++pt; } }   // past the end: re-load the last slice into a free stage
    };
    issue(); issue(); issue(); issue();

    const int fragA = (wr * (MI * 32) + r) * KB, fragB = BM * KB + (wc * (NI * 32) + r) * KB;
    const int fsw = (0x78 >> (2 * ((r >> 2) & 3))) & 3;                 // F[(row >> 2) & 3], see k_to_f16_tiled
    const int so0 = ((0 + h) ^ fsw) << 3, so1 = ((2 + h) ^ fsw) << 3;    // k-step 0 / 1 segment of this lane
    TopT<T + 1> top[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) top[n].init();
    // The accumulators start at |c|^2 / out_scale instead of 0 (out_scale = -2/(s_q s_c) < 0, a power of two up to the factor -2,
    // so the division is exact; word_norm here is that pre-scaled row, see k_scale_norms): after the last slice
    // acc = (|c|^2 - 2 c.q) / out_scale, and ranking the scores ascending is ranking acc DESCENDING. The epilogue is then one
    // compare per value against the lane's current threshold; TopT keeps -acc.
    float thr[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) thr[n] = -__builtin_inff();
    f32x16 acc[MI][NI];
    // A query column is scanned by four lane slots of this workgroup (two accumulator halves h x two wave rows wr). A value that
    // is not better than the (T+1)-th best of ANY of them can be dropped by all of them: thresholds only rise, every dropped value
    // is <= the threshold its lane used at the time <= that lane's final threshold, which is what the slot reports as its bound.
    // Sharing cuts the insertions ~4x. Partner half: one cross-lane read per tile; partner wave: a 4-byte slot in LDS (a stale
    // value is only a lower, i.e. more conservative, threshold).
    const int pw = (1 - wr) * WC + wc;
#pragma unroll
    for (int n = 0; n < NI; ++n) sThr[(wv * NI + n) * 64 + lane] = -__builtin_inff();

    // Software pipeline. Step g multiplies slice g: its k-step-1 fragments (set Y) are read at the top of the step, behind the
    // k-step-0 MFMAs (set X, read during step g-1); slice g+1's k-step-0 fragments are read behind the k-step-1 MFMAs, so no MFMA
    // waits on an LDS round trip. ONE barrier per step, in the middle: before it every wave has waited for its own DMAs of slice
    // g+1 (slices g+2, g+3 = 8 instructions stay in flight) and for its own LDS reads (lgkmcnt(0): slice g's stage is no longer
    // read by anybody), after it slice g+1 is visible to all and slice g+4 is sent into slice g's stage: three slices (96 KB)
    // of look-ahead in a ring of four stages.
    f16x8 xa[MI], xb[NI], ya[MI], yb[NI];
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");     // slice 0 (and the first |c|^2 row) landed
    {
#pragma unroll
        for (int n = 0; n < NI; ++n) xb[n] = *(const f16x8*)(ring + fragB + n * 32 * KB + so0);
#pragma unroll
        for (int m = 0; m < MI; ++m) xa[m] = *(const f16x8*)(ring + fragA + m * 32 * KB + so0);
    }
    int t = 0, kc = 0;
    for (int g = 0; g < G; ++g) {
        const u16* st = ring + (g & (RG_STAGES - 1)) * RG_STAGE_HALVES;
        const u16* sn = ring + ((g + 1) & (RG_STAGES - 1)) * RG_STAGE_HALVES;
        if (!(DBG & 16) || g == 0) {
#pragma unroll
        for (int n = 0; n < NI; ++n) yb[n] = *(const f16x8*)(st + fragB + n * 32 * KB + so1);
#pragma unroll
        for (int m = 0; m < MI; ++m) ya[m] = *(const f16x8*)(st + fragA + m * 32 * KB + so1);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (kc == 0) {
            // first slice of tile t: the accumulators START from the tile's row of |c|^2 / out_scale (pre-scaled per launch by
            // k_scale_norms, landed by DMA with this slice), passed as the C operand -- no re-arming moves in the epilogue
            const float* cnp = sCn + (t & 3) * BM + wr * (MI * 32) + 4 * h;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                f32x16 c0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *(const f32x4*)(cnp + mi * 32 + 8 * j);     // rows 8j + 4h + 0..3 = elements 4j..4j+3
                    c0[4 * j] = v[0]; c0[4 * j + 1] = v[1]; c0[4 * j + 2] = v[2]; c0[4 * j + 3] = v[3];
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[mi], xb[ni], c0, 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                if (!(DBG & 2)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[mi], xb[ni], acc[mi][ni], 0, 0, 0);
                else acc[mi][ni][0] += (float)xa[mi][0] * (float)xb[ni][0];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 32)) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(DBG & 4) || g < 4) issue();                               // slice g+4 -> the stage of slice g
        if (!(DBG & 16)) {
#pragma unroll
        for (int n = 0; n < NI; ++n) xb[n] = *(const f16x8*)(sn + fragB + n * 32 * KB + so0);
#pragma unroll
        for (int m = 0; m < MI; ++m) xa[m] = *(const f16x8*)(sn + fragA + m * 32 * KB + so0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                if (!(DBG & 2)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ya[mi], yb[ni], acc[mi][ni], 0, 0, 0);
                else acc[mi][ni][0] += (float)ya[mi][0] * (float)yb[ni][0];
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (++kc == nk) {
            // epilogue of tile t. A lane inserts ~ (T+1)/n of the n values it has seen, so after the first tiles a 64-lane vector of
            // values rarely holds an insertion: each vector is tested at wave level and the insertion code runs only for those.
                        const int row0 = (mt0 + t) * BM + wr * (MI * 32) + 4 * h;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float a = acc[mi][ni][e];
                        if (DBG & 1) { if (e == 0) top[ni].v[0] += a; }
                        else if (__builtin_expect(__any(a > thr[ni]), 0)) {
                            if (DBG & 256) { top[ni].i[0] += 1; }
                            else {
                            if (a > thr[ni]) top[ni].push(-a, row0 + mi * 32 + (e & 3) + 8 * (e >> 2));
                            thr[ni] = fmaxf(thr[ni], -top[ni].v[T]);
                            }
                        }
                    }
                }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float sh = fmaxf(thr[ni], __shfl_xor(thr[ni], 32, 64));
                sThr[(wv * NI + ni) * 64 + lane] = sh;
                thr[ni] = fmaxf(sh, sThr[(pw * NI + ni) * 64 + lane]);
            }
            kc = 0; ++t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the clamped re-loads past the end: LDS must be quiet before exit
    // candidates: slot = split*(2*WR*T) + (wr*2 + h)*T + t; bound slot = split*(2*WR) + wr*2 + h
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int qi = qtile * BN + wc * (NI * 32) + ni * 32 + r;
        if (qi < nq) {
#pragma unroll
            for (int tt = 0; tt < T; ++tt) {                            // v = -acc, score = out_scale * acc (+inf stays +inf: out_scale < 0)
                const size_t o = (size_t)qi * cand_stride + split * (2 * WR * T) + (wr * 2 + h) * T + tt;
                cand_val[o] = -oscale * top[ni].v[tt]; cand_idx[o] = top[ni].i[tt];
            }
            cand_bound[(size_t)qi * bound_stride + split * (2 * WR) + (wr * 2 + h)] = oscale * thr[ni];   // the lane's final threshold
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same ring on v_mfma_f32_16x16x32_f16
// ---------------------------------------------------------------------------------------------
// A bare MFMA loop (tools/mfma_shape_bench.hip: operands in registers, two waves per SIMD, random f16) sustains 1.96 PFLOP/s
// with the 16x16x32 shape against 1.63 with 32x32x16 on this chip: same cycles per FLOP, but the chip holds a higher clock on the
// small shape (MI355X_MICROARCH 'DVFS give-back' item 7). Geometry, ring, DMA and synchronisation are those of k_knn_l2_ring;
// what changes is the fragment and accumulator layout:
//   A / B fragment of a 16-row tile: lane l reads row (l & 15), 16-byte segment (l >> 4) of the 64-byte slice row: ONE ds_read_b128
//     per 16 x 32 tile (8 for the wave's 128 codeword rows + 4 for its 64 queries per slice); conflict-free with segments XOR-swizzled
//     by F[(row >> 2) & 3], F = {0,2,3,1} (worked out against ds_read_b128's lane groups {0-3,12-15,20-27}, ...)
//   C tile: lane l holds rows 4 (l >> 4) + j, j = 0..3, of column (l & 15): a lane now serves FOUR query columns (one per n-tile)
//     with four codeword rows per tile each, so a query column is scanned by 8 lane slots per workgroup (4 row groups x 2 wave
//     rows) and the kernel leaves 8 slots per codebook split (the host limits it to two splits: 64 candidates per query)
#ifdef ISM_KNN_DBG_VARIANTS
// DBG & 256: [0] wave-tiles, [1] with a hit, [2] flagged columns, [3] flagged groups, [4] flagged scores, [5] inserting lanes;
// [8 + t] flagged scores of tile index t (t < 248) summed over waves
__device__ unsigned long long g_knn_dbg[256];
#endif
// WR = 2: the 256 x 256 tile, 8 waves (2 x 4), one workgroup per CU, four ring stages (three slices in flight).
// WR = 1 (ISMHIP_KNN_HALF=1): a 128 x 256 tile, 4 waves, 76 KB of LDS: TWO independent workgroups per CU, three stages (two in
// flight). The eight waves of the big workgroup meet at a barrier every slice, so their DMA issue and their epilogues coincide
// and the matrix pipes idle meanwhile; two small workgroups drift apart and fill each other's gaps, at 1.5x the DMA per flop.
// QP = 1 (ISMHIP_KNN_QPANEL=1, WR = 2, descriptors of at most 352 elements): a 256 x 128 tile whose QUERY panel (128 queries x all
// slices, 88 KB) is loaded into LDS once per workgroup; only the codeword tiles stream through the ring. The 256 x 256 kernel
// re-reads its 180 KB query tile for every codeword tile, and that is what falls out of the XCD L2s.
// PRE = 1 (WR = 2, QP = 0): the SAMPLING PRE-PASS. The workgroup sweeps every tile_step-th codeword tile (one split) and keeps, per
// query column, only the best score it meets; thr_out[query] = that score (lowered by a few ulps). The main launch (PRE = 0) then
// STARTS every lane slot of the query from thr_init[query] instead of -inf. Why this is sound for ANY start value: thresholds only
// rise, a dropped score is <= the threshold at the time <= the final threshold, which is what the slot reports as its bound -- a
// start value that is too high only makes proofs fail (stage 2 then answers). Why it pays: the insertion code runs whenever ANY of
// a wave's 256 (query, slot) lists takes a score, and from a cold start each of the 24 lists of a query (8 slots x 3 splits) fills
// and refines itself independently (measured: 28 inserting lanes per wave and tile, 27 % of the kernel at 4 slices per tile); the
// best of a 1/16 sample is about the 16th best score of the query overall, so with it as the start only a few dozen scores per
// QUERY (not per list) ever reach the insertion code.
template <int T, int WR = 2, int DBG = 0, int QP = 0, int PRE = 0>
__global__ __launch_bounds__(WR * 256, 2) void k_knn_l2_ring16(const u16* __restrict__ wh, const float* __restrict__ word_norm, int n_tiles_m, int ld, int k_steps,
                                                          const u16* __restrict__ qh, int nq, const float* __restrict__ out_scale,
                                                          int tiles_per_split, int n_splits,
                                                          float* __restrict__ cand_val, int* __restrict__ cand_idx, int cand_stride,
                                                          float* __restrict__ cand_bound, int bound_stride, unsigned int* __restrict__ stream_clock,
                                                          const float* __restrict__ thr_init, float* __restrict__ thr_out, int tile_step, float thr_relax) {
    constexpr int WC = 4, MT = 8, NT = QP == 1 ? 2 : 4, KB = RG_KB, BM = WR * 128, BN = QP == 1 ? 128 : RG_BN;
    constexpr int PROWS = QP == 2 ? 256 : 128;                           // query rows of the resident panel
    // QP = 2 (round 3; stage 1 on <= 160 rotated coordinates): the 256 x 256 tile WITH its whole query panel (256 queries x <= 5 slices,
    // <= 80 KB) resident in LDS: the ring then streams codeword slices only (16 KB per step instead of 32 KB, half the DMA
    // instructions); the panel did not fit next to a four-stage ring at 11 slices, and QP = 1 pays for it with half the queries per tile
    constexpr int STAGES = WR == 2 ? 4 : 3, STAGE_HALVES = (QP ? BM : BM + BN) * KB, CNS = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_smem[];
    u16* ring = (u16*)knn_smem;                                        // [STAGES][BM + BN rows][32 halves] (QP: codeword rows only)
    float* sCn = (float*)(ring + STAGES * STAGE_HALVES);              // [4][CNS]: |c|^2 of four tiles (a DMA always delivers 256 floats)
    float* sThr = sCn + 4 * CNS;                                       // [8 waves][NT][64] (WR = 2 only)
    u16* panel = (u16*)(sThr + 8 * NT * 64);                           // QP: [slices][PROWS queries][32 halves]
    const float oscale = out_scale[0];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv / WC, wc = wv % WC;
    const int fr = lane & 15, fq = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = jx % n_splits, qtile = (jx / n_splits) * 8 + xcd;
    if (qtile * BN >= nq) return;
    const int mt0 = split * tiles_per_split;
    const int n_t = PRE ? (n_tiles_m + tile_step - 1) / tile_step : min(n_tiles_m, mt0 + tiles_per_split) - mt0;
    if (n_t <= 0) return;
    const int nk = (k_steps + 1) / 2;
    const int G = n_t * nk;
    // Joined codeword streams. The workgroups of an XCD that work on the same codebook split read the same codeword tiles, but a
    // workgroup that starts later (second and later rounds of the grid) would begin at the split's first tile while the others are
    // somewhere in the middle: no two of them would ever touch a tile at the same time and every tile would come from beyond
    // the L2 once per workgroup. The order of the tiles does not matter for the result, so a workgroup begins where the stream
    // of its (XCD, split) currently is -- a clock in global memory that every workgroup advances as it finishes tiles -- and wraps
    // around. Nobody waits for anybody.
    int toff = 0; unsigned c0 = 0u;
    if (stream_clock) {
        unsigned int* clk = stream_clock + xcd * n_splits + split;
        if (tid == 0) *(volatile unsigned*)ring = __hip_atomic_load(clk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        c0 = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)ring);
        __syncthreads();                                              // the ring is free for the first DMA
        toff = (int)(c0 % (unsigned)n_t);
        stream_clock = clk;
    }
    auto tile_of = [&](int i) { if (PRE) return i * tile_step; const int x = i + toff; return mt0 + (x >= n_t ? x - n_t : x); };   // i-th tile of this workgroup's sweep

    // DMA shares per slice (pieces of 16 rows x 64 B = 1 KB per wave instruction). WR = 2: waves 0-3 bring 64 codeword rows each,
    // waves 4-7 64 query rows each; WR = 1: every wave brings 32 codeword rows and 64 query rows. Both images are stored in
    // 256-row tiles [tile][slice][row][64 B]; a 128-row codeword tile is one half of such a block.
    constexpr int NA = QP ? 2 : (WR == 2 ? 4 : 2), NB = 4;
    const bool dma_a = QP || WR == 1 || wv < 4, dma_b = !QP && (WR == 1 || wv >= 4);
    const unsigned lane_off = (unsigned)(lane * 16);
    const int row_a = QP ? wv * 32 : (WR == 2 ? (wv & 3) * 64 : wv * 32), row_b = (wv & 3) * 64;
    const char* qbase = QP == 2 ? (const char*)(qh + (size_t)qtile * nk * (256 * KB)) + (wv * 32) * (KB * 2)
                      : QP ? (const char*)(qh + (size_t)(qtile >> 1) * nk * (256 * KB)) + ((qtile & 1) * 128 + wv * 16) * (KB * 2)
                           : (const char*)(qh + (size_t)qtile * nk * (BN * KB)) + row_b * (KB * 2);
    if (QP == 2) {       // the query panel: wave w brings rows 32 w .. 32 w + 31 of every slice (two 16-row pieces)
        for (int s_ = 0; s_ < nk; ++s_) {
            lds_dma16(qbase + (size_t)s_ * (256 * KB * 2) + lane_off, panel + (s_ * 256 + wv * 32) * KB);
            lds_dma16(qbase + (size_t)s_ * (256 * KB * 2) + 16 * KB * 2 + lane_off, panel + (s_ * 256 + wv * 32 + 16) * KB);
        }
    } else if (QP) {     // the query panel: wave w brings rows 16 w .. 16 w + 15 of every slice
        for (int s_ = 0; s_ < nk; ++s_) lds_dma16(qbase + (size_t)s_ * (256 * KB * 2) + lane_off, panel + (s_ * 128 + wv * 16) * KB);
    }
    int pt = 0, pkc = 0, ps = 0;
    auto issue = [&]() {
        u16* st = ring + ps * STAGE_HALVES;
        const int tt = tile_of(pt);
        if (wv == 0 && pkc == 0) lds_dma16(word_norm + (size_t)tt * BM + lane * 4, sCn + (pt & 3) * CNS);
        if (dma_a) {
            const char* sp = (const char*)wh + ((size_t)(WR == 2 ? tt : (tt >> 1)) * nk + pkc) * (256 * KB * 2) + ((WR == 2 ? 0 : (tt & 1) * 128) + row_a) * (KB * 2);
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (DBG & 128) lds_dma16_nt(sp + j * (16 * KB * 2) + lane_off, st + (row_a + j * 16) * KB);   // timing / traffic experiments
                else lds_dma16(sp + j * (16 * KB * 2) + lane_off, st + (row_a + j * 16) * KB);
            }
        }
        if (dma_b) {
            const char* sp = qbase + (size_t)pkc * (BN * KB * 2);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (DBG & 1024) lds_dma16_nt(sp + j * (16 * KB * 2) + lane_off, st + BM * KB + (row_b + j * 16) * KB);
                else lds_dma16(sp + j * (16 * KB * 2) + lane_off, st + BM * KB + (row_b + j * 16) * KB);
            }
        }
        if (++ps == STAGES) ps = 0;
        if (pt * nk + pkc + 1 < G) { if (++pkc == nk) { pkc = 0; ++pt; } }
    };
#pragma unroll
    for (int i = 0; i < STAGES; ++i) issue();

    // fragment address of this lane inside a 16-row tile: row fr, physical segment fq ^ F[(fr >> 2) & 3]
    const int fso = (fr * KB) + ((fq ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 3);
    const int fragA = wr * (MT * 16) * KB + fso, fragB = (QP ? 0 : BM * KB) + wc * (NT * 16) * KB + fso;
    TopT<T + 1> top[NT];
    float thr[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        top[n].init(); thr[n] = -__builtin_inff();
        if (!PRE && thr_init) { const int qi_ = qtile * BN + wc * (NT * 16) + n * 16 + fr; if (qi_ < nq) thr[n] = thr_init[qi_]; }
    }
    f32x4 acc[MT][NT];
#ifdef ISM_KNN_DBG_VARIANTS
    unsigned dbg_c[6] = {0, 0, 0, 0, 0, 0};
#endif
    const int pw = (1 - wr) * WC + wc;
    if (WR == 2) {
#pragma unroll
        for (int n = 0; n < NT; ++n) sThr[(wv * NT + n) * 64 + lane] = -__builtin_inff();
    }

    // pipeline as in k_knn_l2_ring, with the fragments split by codeword rows instead of k-steps: X = tiles 0-3 (read during the
    // previous step), Y = tiles 4-7 and the four query fragments (read at the top of the step)
    f16x8 xa[4], ya[4], bq[NT];
    if (QP) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");        // panel + slice 0 landed (2 pieces per wave and slice)
    else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m) xa[m] = *(const f16x8*)(ring + fragA + m * 16 * KB);
    int t = 0, kc = 0, gs = 0;
    for (int g = 0; g < G; ++g) {
        const int gn = gs + 1 == STAGES ? 0 : gs + 1;
        const u16* st = ring + gs * STAGE_HALVES;
        const u16* sn = ring + gn * STAGE_HALVES;
        gs = gn;
        if (!(DBG & 16) || g == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[n] = *(const f16x8*)((QP ? panel + kc * (PROWS * KB) : st) + fragB + n * 16 * KB);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // first slice of a tile: the accumulators START from the tile's pre-scaled |c|^2 row (rows 16 mt + 4 fq + j), passed as
        // the C operand of the tile's first MFMAs. The fragment reads of the other half-step are issued one per four MFMAs, so
        // the first MFMAs wait only for the query fragments and the reads ride inside the MFMA stream.
        const float* cnp = sCn + (t & 3) * CNS + wr * (MT * 16) + 4 * fq;
        auto mma4 = [&](int mb, const f16x8* af, f16x8* nxt, const u16* nsrc) {
            if (kc == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 c0 = (DBG & 2048) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(cnp + (mb + mt) * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { if (!(DBG & 2)) acc[mb + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bq[nt], c0, 0, 0, 0); else acc[mb + mt][nt] = c0; }
                    if (!(DBG & 16) || g == 0) nxt[mt] = *(const f16x8*)(nsrc + mt * 16 * KB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) if (!(DBG & 2)) acc[mb + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bq[nt], acc[mb + mt][nt], 0, 0, 0);
                    if (!(DBG & 16) || g == 0) nxt[mt] = *(const f16x8*)(nsrc + mt * 16 * KB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        mma4(0, xa, ya, st + fragA + 4 * 16 * KB);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 32)) {
            if (QP) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (WR == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (!(DBG & 4) || g < 4) issue();
        __builtin_amdgcn_sched_barrier(0);
        mma4(4, ya, xa, sn + fragA);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (++kc == nk) {
            const int row0 = tile_of(t) * BM + wr * (MT * 16) + 4 * fq;
            if (stream_clock && tid == 0) atomicMax(stream_clock, c0 + (unsigned)t + 1u);
            if (PRE) {
                // pre-pass: the best score of the column so far, nothing else (16 v_max3 per column and tile)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float m = thr[nt];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(acc[mt][nt][0]), "v"(acc[mt][nt][1]));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(acc[mt][nt][2]), "v"(acc[mt][nt][3]));
                    }
                    thr[nt] = m;
                }
            } else if (DBG & 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) top[nt].v[0] += acc[mt][nt][0];
            } else {
                // Epilogue. A scalar branch right behind the vector compare it depends on stalls ~19 cycles, and 128 of those pairs
                // per tile were a good part of the kernel. So: the largest of a column's 32 scores by 16 v_max3_f32, ONE compare
                // per column into its own SGPR pair, one branch per tile (measured: the test itself is free, 13.0 ms with and
                // without it); only a column that does hold a score above its threshold is walked.
                float mx[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float m;
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[0][nt][0]), "v"(acc[0][nt][1]), "v"(acc[0][nt][2]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(acc[0][nt][3]));
#pragma unroll
                    for (int mt = 1; mt < MT; ++mt) {
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(acc[mt][nt][0]), "v"(acc[mt][nt][1]));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(acc[mt][nt][2]), "v"(acc[mt][nt][3]));
                    }
                    mx[nt] = m;
                }
                unsigned long long hit[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) hit[nt] = __ballot(mx[nt] > thr[nt]);
                unsigned long long any_hit = 0ull;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) any_hit |= hit[nt];
#ifdef ISM_KNN_DBG_VARIANTS
                if (DBG & 256) { ++dbg_c[0]; if (any_hit != 0ull) ++dbg_c[1]; }
#endif
                if (DBG & 64) { if (any_hit != 0ull) top[0].i[0] += 1; }
                else if (__builtin_expect(any_hit != 0ull, 0)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (hit[nt] == 0ull) continue;
#ifdef ISM_KNN_DBG_VARIANTS
                        if (DBG & 256) ++dbg_c[2];
#endif
                        // Every step ends in a workgroup barrier, so a tile's epilogue costs what it costs the SLOWEST of the eight
                        // waves: keep the walk of a flagged column short. The largest of each 4-row group (two instructions per
                        // group), eight compares into eight SGPR pairs, eight scalar tests; only a group that holds a score above
                        // the threshold has its four scores compared and inserted (the empty asm keeps the compiler from sinking
                        // every compare next to its branch again). A score is re-tested against the threshold as it stands when
                        // its turn comes; the insertion itself is branch-free.
                        float gm[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(gm[mt]) : "v"(acc[mt][nt][0]), "v"(acc[mt][nt][1]), "v"(acc[mt][nt][2]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(gm[mt]) : "v"(gm[mt]), "v"(acc[mt][nt][3]));
                        }
                        unsigned long long gk[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) gk[mt] = __ballot(gm[mt] > thr[nt]);
                        asm volatile("" :: "s"(gk[0]), "s"(gk[1]), "s"(gk[2]), "s"(gk[3]), "s"(gk[4]), "s"(gk[5]), "s"(gk[6]), "s"(gk[7]));
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (gk[mt] == 0ull) continue;
#ifdef ISM_KNN_DBG_VARIANTS
                            if (DBG & 256) ++dbg_c[3];
#endif
                            unsigned long long mk[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) mk[j] = __ballot(acc[mt][nt][j] > thr[nt]);
                            asm volatile("" :: "s"(mk[0]), "s"(mk[1]), "s"(mk[2]), "s"(mk[3]));
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (mk[j] == 0ull) continue;
#ifdef ISM_KNN_DBG_VARIANTS
                                if (DBG & 256) { ++dbg_c[4]; dbg_c[5] += __popcll(mk[j]); if (lane == 0 && t < 248) atomicAdd(&g_knn_dbg[8 + t], 1ull); }
#endif
                                const float a = acc[mt][nt][j];
                                top[nt].push_flat(a > thr[nt] ? -a : __builtin_inff(), row0 + mt * 16 + j);
                                asm("v_max_f32_e64 %0, %1, -%2" : "=v"(thr[nt]) : "v"(thr[nt]), "v"(top[nt].v[T]));
                            }
                        }
                    }
                }
            }
            // thresholds shared by the 8 lane slots of a query column: the four row groups of this wave (lanes fr, fr+16, fr+32,
            // fr+48) by two register swaps (v_permlane32_swap / v_permlane16_swap: no LDS round trip), then the partner wave row
            // through LDS (see k_knn_l2_ring)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (DBG & 64) asm volatile("" : "+v"(thr[nt]));
                const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(thr[nt]), __float_as_uint(thr[nt]), false, false);
                float sh;
                asm("v_max_f32 %0, %1, %2" : "=v"(sh) : "v"(__uint_as_float(h[0])), "v"(__uint_as_float(h[1])));
                const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(sh), __float_as_uint(sh), false, false);
                asm("v_max_f32 %0, %1, %2" : "=v"(sh) : "v"(__uint_as_float(q[0])), "v"(__uint_as_float(q[1])));
                if (WR == 2) {
                    sThr[(wv * NT + nt) * 64 + lane] = sh;
                    const float other = sThr[(pw * NT + nt) * 64 + lane];
                    asm("v_max_f32 %0, %1, %2" : "=v"(thr[nt]) : "v"(sh), "v"(other));
                } else thr[nt] = sh;
            }
            kc = 0; ++t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ISM_KNN_DBG_VARIANTS
    if ((DBG & 256) && lane == 0) for (int c = 0; c < 6; ++c) atomicAdd(&g_knn_dbg[c], (unsigned long long)dbg_c[c]);
#endif
    if (PRE) {
        // thr[] is the best over this wave's four row groups and (through sThr, one tile late) the partner wave row; one more exchange
        // behind a barrier makes it the best of the whole sample: the nearest SAMPLED row in the stage-1 coordinates. The start value
        // handed to the main launch is that score RELAXED by thr_relax (< 0 in accumulator units): the proof of a query needs every
        // row it drops to lie beyond the nearest neighbour's FULL distance, which exceeds its stage-1 distance by the energy the
        // truncation left out -- a start value right at the sample's best makes the proof fail whenever that best is (close to) the
        // nearest neighbour itself (measured: 15.7 % instead of 6.7 % of the queries).
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sThr[(wv * NT + nt) * 64 + lane] = thr[nt];
        __syncthreads();
        if (wr == 0 && fq == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int qi = qtile * BN + wc * (NT * 16) + nt * 16 + fr;
                const float b = fmaxf(thr[nt], sThr[(pw * NT + nt) * 64 + lane]);
                if (qi < nq) thr_out[qi] = b - fabsf(b) * 3.814697265625e-06f + thr_relax;
            }
        }
        return;
    }
    // candidates: slot = split*(WR*4*T) + (wr*4 + fq)*T + t; bound slot = split*WR*4 + wr*4 + fq
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int qi = qtile * BN + wc * (NT * 16) + nt * 16 + fr;
        if (qi < nq) {
#pragma unroll
            for (int tt = 0; tt < T; ++tt) {
                const size_t o = (size_t)qi * cand_stride + split * (WR * 4 * T) + (wr * 4 + fq) * T + tt;
                cand_val[o] = -oscale * top[nt].v[tt]; cand_idx[o] = top[nt].i[tt];
            }
            cand_bound[(size_t)qi * bound_stride + split * (WR * 4) + (wr * 4 + fq)] = oscale * thr[nt];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// chi-square candidates on the vector ALUs
// ---------------------------------------------------------------------------------------------
#define CHI_B 64
#define CHI_LDK 33
// flag[0] != 0: some element of the query batch is negative or NaN
__global__ void k_any_negative(const float* __restrict__ src, int n, int dim, int ld, uint32_t* __restrict__ flag) {
    bool bad = false;
    const size_t tot = (size_t)n * dim;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const float v = src[(i / dim) * (size_t)ld + i % dim];
        bad |= !(v >= 0.f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
typedef float chi_f32x2 __attribute__((ext_vector_type(2)));
template <int T>
__global__ __launch_bounds__(256) void k_knn_chi2(const float* __restrict__ words, int n_words_pad, int dim_pad,
                                                  const float* __restrict__ q, int nq, int ldq, int words_nonneg, const uint32_t* __restrict__ q_negative,
                                                  int tiles_per_split,
                                                  float* __restrict__ cand_val, int* __restrict__ cand_idx, int cand_stride,
                                                  float* __restrict__ cand_bound, int bound_stride) {
    __shared__ float sC[CHI_B * CHI_LDK];
    __shared__ float sQ[CHI_B * CHI_LDK];
    __shared__ float sMv[CHI_B][16][T + 1];
    __shared__ int sMi[CHI_B][16][T + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;       // tx -> 4 query columns, ty -> 4 codeword rows
    const int qtile = blockIdx.x, split = blockIdx.y;
    const int n_tiles = n_words_pad / CHI_B;
    const int mt0 = split * tiles_per_split, mt1 = min(n_tiles, mt0 + tiles_per_split);
    const int nk = dim_pad / 32;
    const bool fast = words_nonneg && q_negative[0] == 0u;        // uniform
    TopT<T + 1> top[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) top[j].init();
    // staging: 64 rows x 32 floats = 2048 floats, 8 per thread: row = tid/4, cols (tid%4)*8 .. +7
    const int srow = tid >> 2, scol = (tid & 3) * 8;
    int qr = qtile * CHI_B + srow; qr = qr < nq ? qr : nq - 1;
    const float* qp = q + (size_t)qr * ldq + scol;
    for (int mt = mt0; mt < mt1; ++mt) {
        const float* cp = words + (size_t)(mt * CHI_B + srow) * dim_pad + scol;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sC[srow * CHI_LDK + scol + e] = cp[kc * 32 + e];
                sQ[srow * CHI_LDK + scol + e] = qp[kc * 32 + e];
            }
            __syncthreads();
            if (fast) {
                // Histogram data (no negative element on either side): sum > 0 unless both elements are 0, and then diff = 0 too.
                // Adding 1e-30 to the codeword element INSIDE the sum only (it vanishes next to any float above 1e-23, and makes a
                // 0 + 0 sum positive: 0 * rcp(1e-30) = 0) replaces the functor's test, and the element pairs go through the packed
                // FP32 instructions: v_pk_add_f32 x2, v_pk_mul_f32, v_pk_fma_f32 and two v_rcp_f32 per TWO elements -- the kernel
                // is VALU-issue bound (it ran at the full issue rate before: 6.2 lane-operations per element; this is 4).
                // A sum is never made larger by more than 1e-30, so the score stays a lower bound of the functor value as before.
#pragma unroll 4
                for (int kk = 0; kk < 32; ++kk) {
                    float cv[4], qv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[i] = sC[(ty * 4 + i) * CHI_LDK + kk];
#pragma unroll
                    for (int j = 0; j < 4; ++j) qv[j] = sQ[(tx * 4 + j) * CHI_LDK + kk];
                    const chi_f32x2 q01 = {qv[0], qv[1]}, q23 = {qv[2], qv[3]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ct = cv[i] + 1e-30f;
                        const chi_f32x2 c2 = {cv[i], cv[i]}, c2t = {ct, ct};
                        const chi_f32x2 s0 = c2t + q01, s1 = c2t + q23, d0 = c2 - q01, d1 = c2 - q23;
                        const chi_f32x2 r0 = {__builtin_amdgcn_rcpf(s0.x), __builtin_amdgcn_rcpf(s0.y)}, r1 = {__builtin_amdgcn_rcpf(s1.x), __builtin_amdgcn_rcpf(s1.y)};
                        chi_f32x2 a0 = {acc[i][0], acc[i][1]}, a1 = {acc[i][2], acc[i][3]};
                        a0 = __builtin_elementwise_fma(d0 * d0, r0, a0);
                        a1 = __builtin_elementwise_fma(d1 * d1, r1, a1);
                        acc[i][0] = a0.x; acc[i][1] = a0.y; acc[i][2] = a1.x; acc[i][3] = a1.y;
                    }
                }
            } else {
#pragma unroll 4
            for (int kk = 0; kk < 32; ++kk) {
                float cv[4], qv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[i] = sC[(ty * 4 + i) * CHI_LDK + kk];
#pragma unroll
                for (int j = 0; j < 4; ++j) qv[j] = sQ[(tx * 4 + j) * CHI_LDK + kk];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float s = cv[i] + qv[j], d = cv[i] - qv[j];
                        const float t = d * d * __builtin_amdgcn_rcpf(s);
                        acc[i][j] += s > 0.f ? t : 0.f;
                    }
            }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) top[j].push(acc[i][j], mt * CHI_B + ty * 4 + i);
    }
    // merge the 16 row-threads of every query column: best T are the candidates, the (T+1)-th smallest value bounds
    // everything that was dropped (each thread's own (T+1)-th value bounds what that thread dropped)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < T + 1; ++t) { sMv[tx * 4 + j][ty][t] = top[j].v[t]; sMi[tx * 4 + j][ty][t] = top[j].i[t]; }
    __syncthreads();
    if (tid < CHI_B) {
        const int qi = qtile * CHI_B + tid;
        if (qi < nq) {
            TopT<T + 1> best; best.init();
            for (int y = 0; y < 16; ++y)
#pragma unroll
                for (int t = 0; t < T + 1; ++t) {
                    // order by (value, row): rows from different threads interleave, so compare rows on equal values
                    const float v = sMv[tid][y][t]; const int id = sMi[tid][y][t];
                    if (id < 0) continue;
                    if (v < best.v[T] || (v == best.v[T] && id < best.i[T]) || best.i[T] < 0) {
                        best.v[T] = v; best.i[T] = id;
#pragma unroll
                        for (int u = T; u > 0; --u)
                            if (best.i[u - 1] < 0 || best.v[u] < best.v[u - 1] || (best.v[u] == best.v[u - 1] && best.i[u] < best.i[u - 1])) {
                                float tv = best.v[u]; best.v[u] = best.v[u - 1]; best.v[u - 1] = tv;
                                int ti = best.i[u]; best.i[u] = best.i[u - 1]; best.i[u - 1] = ti;
                            }
                    }
                }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const size_t o = (size_t)qi * cand_stride + split * T + t;
                cand_val[o] = best.v[t]; cand_idx[o] = best.i[t];
            }
            cand_bound[(size_t)qi * bound_stride + split] = best.i[T] >= 0 ? best.v[T] : __builtin_inff();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// exact re-rank with the FLANN functors' summation order
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float flann_l2(const float* a, const float* b, int size) {
    float result = 0.f;
    int i = 0;
    for (; i + 3 < size; i += 4) {
        const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < size; ++i) { const float d0 = a[i] - b[i]; result += d0 * d0; }
    return result;
}
__device__ __forceinline__ float flann_chi2(const float* a, const float* b, int size) {
    float result = 0.f;
    for (int i = 0; i < size; ++i) {
        const float sum = a[i] + b[i];
        if (sum > 0) { const float diff = a[i] - b[i]; result += diff * diff / sum; }
    }
    return result;
}

// Folds the candidates of MANY codebook splits into one slot. A wave per query keeps the KNN_MERGE_KEEP smallest approximate
// scores of all splits' candidates; the merged bound is the smallest score anything dropped: every split slot's own bound and the
// best candidate this merge leaves out. (A NaN bound stays NaN, so that the proof fails and the query takes the exact scan.)
#define KNN_MERGE_KEEP 16
__device__ __forceinline__ unsigned knn_sortable(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__global__ __launch_bounds__(256) void k_knn_merge_splits(int nq, const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int n_cand,
                                                          const float* __restrict__ cand_bound, int n_bound,
                                                          float* __restrict__ out_val, int* __restrict__ out_idx, float* __restrict__ out_bound) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int lane = lane_id();
    const float* v = cand_val + (size_t)qi * n_cand; const int* id = cand_idx + (size_t)qi * n_cand;
    unsigned long long prev = 0ull;                                    // keys are > 0: bit 63 or the complement of a negative float
    bool first = true;
    float dropped = __builtin_inff();
    for (int r = 0; r <= KNN_MERGE_KEEP; ++r) {
        unsigned long long best = ~0ull;
        for (int i = lane; i < n_cand; i += 64) {
            if (id[i] < 0) continue;                                   // empty slot entry
            const unsigned long long key = ((unsigned long long)knn_sortable(v[i]) << 32) | (unsigned)i;
            if ((first || key > prev) && key < best) best = key;
        }
        best = wave_min_u64(best);
        if (r < KNN_MERGE_KEEP) {
            if (lane == 0) {
                const bool have = best != ~0ull;
                const int i = have ? (int)(best & 0xffffffffull) : 0;
                out_val[(size_t)qi * KNN_MERGE_KEEP + r] = have ? v[i] : __builtin_inff();
                out_idx[(size_t)qi * KNN_MERGE_KEEP + r] = have ? id[i] : -1;
            }
        } else if (best != ~0ull) dropped = v[(int)(best & 0xffffffffull)];
        if (best == ~0ull) { for (int rr = r + 1; rr < KNN_MERGE_KEEP; ++rr) if (lane == 0) { out_val[(size_t)qi * KNN_MERGE_KEEP + rr] = __builtin_inff(); out_idx[(size_t)qi * KNN_MERGE_KEEP + rr] = -1; } break; }
        prev = best; first = false;
    }
    float b = dropped;
    for (int i = lane; i < n_bound; i += 64) { const float x = cand_bound[(size_t)qi * n_bound + i]; if (x < b || x != x) b = x; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float x = __shfl_xor(b, o, 64); if (x < b || x != x) b = x; }
    if (lane == 0) out_bound[qi] = b;
}

struct VerifyParams {
    float ku;         // 1.01 * K * u : relative error bound of a K-term fp32 functor sum (u = 2^-24)
    float dot_rel;    // bound on |approx(q.c) - q.c| / (|q||c|) of the candidate kernel (f32 fma chain: ku; bf16x3: see k_knn_l2_mfma16)
    float cmax2;      // max |c|^2 over the codebook (L2 only)
    float dabs_c;     // f16 candidates: worst-case absolute error of one codebook element (2^-14 / scale), else 0
    const float* dabs_q;   // f16 candidates: the same for the query batch (device scalar), else nullptr
    float sqrt_dim;   // sqrt(dim_pad)
    float cn_acc;     // k_knn_l2_ring adds |c|^2 through the accumulator: extra 1.01 (K+1) 2^-23 |c|max^2 on the score, else 0
};
// absolute part of the candidate kernel's dot-product error: sum |dq_i c_i| + |q_i dc_i| + |dq_i dc_i| with |dq_i| <= dq, |dc_i| <= dc
__device__ __forceinline__ float knn_abs_err(const VerifyParams& vp, float qn2) {
    if (!vp.dabs_q) return 0.f;
    const float dq = vp.dabs_q[0], dc = vp.dabs_c;
    return 1.01f * (vp.sqrt_dim * (dq * sqrtf(vp.cmax2) + dc * sqrtf(qn2)) + vp.sqrt_dim * vp.sqrt_dim * dq * dc);
}
#define KNN_U 5.9604645e-08f
#define KNN_HELL_CPL 4            // candidates per lane of k_knn_rerank_hell (256 per query)

#include "functor.h"

__global__ __launch_bounds__(256) void k_knn_rerank(const float* __restrict__ words, int dim, int dim_pad, int n_words,
                                                    const float* __restrict__ q, int nq, int ldq, int metric,
                                                    const int* __restrict__ cand_idx, const float* __restrict__ cand_val, int cand_stride, int n_cand,
                                                    const float* __restrict__ cand_bound, int n_bound, VerifyParams vp,
                                                    int k, int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                    uint32_t* __restrict__ flag_count, uint32_t* __restrict__ qrec, uint32_t* __restrict__ items) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int lane = lane_id();
    const float* qp = q + (size_t)qi * ldq;
    // |q|^2 by a wave sum (needed by the error bounds below)
    float qn2 = 0.f;
    for (int i = lane; i < dim; i += 64) { const float v = qp[i]; qn2 += v * v; }
    qn2 = wave_sum_f(qn2);
    // n_cand <= 64 by construction (host): one candidate per lane. Only candidates whose approximate score is within twice the
    // error bound of the k-th best approximate score can be among the exact k best; the others skip the functor.
    int id = -1; float av = __builtin_inff();
    if (lane < n_cand) { id = cand_idx[(size_t)qi * cand_stride + lane]; av = cand_val[(size_t)qi * cand_stride + lane]; }
    if (!(id >= 0 && id < n_words)) { id = -1; av = __builtin_inff(); }
    float kth = av;
    {
        float cur = av;                       // k-th smallest approximate score (k <= 4) by k rounds of wave-min
        for (int j = 0; j < k; ++j) {
            const float mn = wave_min_f(cur);
            kth = mn;
            const unsigned long long eq = __ballot(cur == mn);
            if (eq && lane == __ffsll((long long)eq) - 1) cur = __builtin_inff();      // retire one instance
        }
    }
    float slack;
    if (metric == ISMHIP_METRIC_CHI2) slack = 4.f * (((float)dim_pad + 8.f) * KNN_U + vp.ku) * fabsf(kth);
    else slack = 2.f * (17.f * KNN_U * vp.cmax2 + (2.f * vp.dot_rel + 2.f * KNN_U) * sqrtf(qn2 * vp.cmax2) + 2.f * knn_abs_err(vp, qn2) + vp.cn_acc * vp.cmax2) + 4.f * vp.ku * (qn2 + fabsf(kth) + vp.cmax2);
    unsigned long long key = ~0ull;
    {
        __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
        float* sT = s_terms[threadIdx.x >> 6];
        unsigned long long need = __ballot(id >= 0 && !(av > kth + slack));     // NaN scores are never skipped
        while (need) {
            const int src = __ffsll((long long)need) - 1; need &= need - 1;
            const int cid = __shfl(id, src, 64);
            const float d = wave_functor(metric, qp, words + (size_t)cid * dim_pad, dim, lane, sT);
            // distances are >= 0 (or NaN); positive float bit patterns order like unsigned integers
            if (lane == src) key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;
        }
    }
    // every bound slot (L2: split x 4 lane slots, chi2: split) holds the smallest approximate score that slot dropped
    float bnd = __builtin_inff();
    if (lane < n_bound) bnd = cand_bound[(size_t)qi * n_bound + lane];
    float dk = 0.f; bool have_k = true;
    for (int j = 0; j < k; ++j) {
        const unsigned long long mn = wave_min_u64(key);
        if (lane == 0) {
            if (mn == ~0ull) { idx_out[(size_t)qi * k + j] = -1; dist_out[(size_t)qi * k + j] = __builtin_nanf(""); }
            else { idx_out[(size_t)qi * k + j] = (int)(mn & 0xffffffffull); dist_out[(size_t)qi * k + j] = __uint_as_float((unsigned)(mn >> 32)); }
        }
        if (mn == ~0ull) have_k = false; else dk = __uint_as_float((unsigned)(mn >> 32));
        if (key == mn) key = ~0ull;     // rows are unique among candidates, so exactly one lane retires
    }
    // Proof of exactness, slot by slot. A codeword dropped by slot b has approximate score >= bnd_b, so
    //   L2  : true distance D >= |q|^2 (1 - 16u) + bnd_b - eps_s,  eps_s = 17u |c|max^2 + (2 dot_rel + 2u) |q||c|max
    //         (|c|^2 by a short tree sum: 16u; K-term fma chain: 1.01 K u on sum|q_i c_i| <= |q||c|; final subtraction: u)
    //   chi2: all terms are non-negative, v_rcp_f32 is 1 ulp: D >= bnd_b (1 - (K + 8) u)
    // and its functor value is >= D (1 - 1.01 K u). If that is above the k-th exact functor value, the slot cannot hold a
    // better row. Slots that fail (or NaNs) are handed to the exact scan, restricted to the rows of those slots.
    // A slot whose bound is still +inf dropped nothing -- unless scores overflowed (+inf / NaN scores are never kept): that needs
    // an inf or NaN in |q|^2, |c|max^2 or their product (or in the f16 scales), all of which make the error bound below non-finite.
    bool viol = false;
    const float eps_chk = metric == ISMHIP_METRIC_CHI2 ? qn2 + vp.cmax2
                        : 17.f * KNN_U * vp.cmax2 + (2.f * vp.dot_rel + 2.f * KNN_U) * sqrtf(qn2 * vp.cmax2) + 2.f * knn_abs_err(vp, qn2) + vp.cn_acc * vp.cmax2 + qn2;
    if (lane < n_bound && (bnd != __builtin_inff() || !(eps_chk < __builtin_inff()))) {
        if (!have_k) viol = true;
        else if (metric == ISMHIP_METRIC_CHI2) {
            const float lo = bnd * (1.f - ((float)dim_pad + 8.f) * KNN_U) * (1.f - vp.ku);
            viol = !(dk < lo);
        } else {
            const float eps_s = 17.f * KNN_U * vp.cmax2 + (2.f * vp.dot_rel + 2.f * KNN_U) * sqrtf(qn2 * vp.cmax2) + 2.f * knn_abs_err(vp, qn2) + vp.cn_acc * vp.cmax2;
            const float rhs = qn2 * (1.f - 16.f * KNN_U) + bnd - eps_s;
            viol = !(dk < rhs - vp.ku * fabsf(rhs) - 1e-37f);
        }
    }
    // queue: one record per unproven query {query, first item, #items} and one work item {query, slot} per failing slot
    const unsigned long long vmask = __ballot(viol);
    if (vmask != 0ull) {
        const int nv = __popcll(vmask);
        uint32_t ibase = 0;
        if (lane == 0) {
            const uint32_t qs = atomicAdd(&flag_count[0], 1u);
            ibase = atomicAdd(&flag_count[1], (uint32_t)nv);
            qrec[3 * (size_t)qs] = (uint32_t)qi; qrec[3 * (size_t)qs + 1] = ibase; qrec[3 * (size_t)qs + 2] = (uint32_t)nv;
        }
        ibase = __shfl(ibase, 0, 64);
        if (viol) {
            const uint32_t it = ibase + __popcll(vmask & ((1ull << lane) - 1ull));
            items[2 * (size_t)it] = (uint32_t)qi; items[2 * (size_t)it + 1] = (uint32_t)lane;
        }
    }
}

// ---- re-rank + proof for a stage 1 that ran on the ROTATED, TRUNCATED image (pca.hip) --------------------------------------------
// The candidate scores are now |c^|^2 - 2 c^.q^ over the m leading rotated coordinates (x^ = the f16 image of fl(R x) / scale): up
// to accumulation error a LOWER bound piece of the functor value, not an approximation of it. With
//   L(s)  = |q^|^2 (1 - 16u) + s - eps_acc                  <= |q^ - c^|^2     (eps_acc as in k_knn_rerank, accumulation terms only)
//   LB(s) = ((sqrt(L) - delta_q - delta_c)+ )^2 / sigma_max(R)^2                <= |q - c|^2   (pca.hip header)
// a row whose score is s has functor value >= LB(s) (1 - ku). Consequences:
//   * candidates are evaluated with the exact functor in ascending order of their score until the next one's LB exceeds the k-th
//     exact value found so far (the others cannot be among the k best, nor tie with them);
//   * a slot whose dropped-score bound b has LB(b) (1 - ku) above the k-th exact value cannot hide a better row: proven.
struct PcaVerify {
    const u16* qimg; int nk;          // rotated f16 query image (tiled), slices per row
    float inv_sq2;                    // 1 / scale^2 of that image
    float inv_sig2, d_rel, dq_abs, dc;   // 1 / sigma_max^2 (rounded down); |x^ - R x| <= d_rel |x| + abs; dc = the codeword side for |c| = |c|max
    float eps_c2, dot2, cmax2;        // eps_acc = eps_c2 + dot2 |q^||c^|max;  cmax2 = max |c^|^2
    float ku;
};
__device__ __forceinline__ void knn_queue_unproven(bool viol, int qi, int lane, uint32_t* __restrict__ flag_count, uint32_t* __restrict__ qrec, uint32_t* __restrict__ items) {
    // queue: one record per unproven query {query, first item, #items} and one work item {query, slot} per failing slot
    const unsigned long long vmask = __ballot(viol);
    if (vmask != 0ull) {
        const int nv = __popcll(vmask);
        uint32_t ibase = 0;
        if (lane == 0) {
            const uint32_t qs = atomicAdd(&flag_count[0], 1u);
            ibase = atomicAdd(&flag_count[1], (uint32_t)nv);
            qrec[3 * (size_t)qs] = (uint32_t)qi; qrec[3 * (size_t)qs + 1] = ibase; qrec[3 * (size_t)qs + 2] = (uint32_t)nv;
        }
        ibase = __shfl(ibase, 0, 64);
        if (viol) {
            const uint32_t it = ibase + __popcll(vmask & ((1ull << lane) - 1ull));
            items[2 * (size_t)it] = (uint32_t)qi; items[2 * (size_t)it + 1] = (uint32_t)lane;
        }
    }
}
__global__ __launch_bounds__(256) void k_knn_rerank_pca(const float* __restrict__ words, int dim, int dim_pad, int n_words,
                                                        const float* __restrict__ q, int nq, int ldq,
                                                        const int* __restrict__ cand_idx, const float* __restrict__ cand_val, int cand_stride, int n_cand,
                                                        const float* __restrict__ cand_bound, int n_bound, PcaVerify pv,
                                                        int k, int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                        uint32_t* __restrict__ flag_count, uint32_t* __restrict__ qrec, uint32_t* __restrict__ items) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int lane = lane_id();
    const float* qp = q + (size_t)qi * ldq;
    float qn2 = 0.f;
    for (int i = lane; i < dim; i += 64) { const float v = qp[i]; qn2 += v * v; }
    qn2 = wave_sum_f(qn2);
    float qn2h = 0.f;                                                  // |q^|^2 from the image itself (any element order)
    {
        const size_t tile = (size_t)(qi >> 8); const int r = qi & 255;
        for (int i = lane; i < pv.nk * 32; i += 64) {
            const float v = (float)__builtin_bit_cast(_Float16, pv.qimg[((tile * pv.nk + (i >> 5)) * 256 + r) * 32 + (i & 31)]);
            qn2h += v * v;
        }
        qn2h = wave_sum_f(qn2h) * pv.inv_sq2;
    }
    const float dlt = pv.d_rel * (sqrtf(qn2) * 1.00001f) + pv.dq_abs + pv.dc;
    const float eps_s = (pv.eps_c2 + pv.dot2 * sqrtf(qn2h * pv.cmax2)) * 1.00001f;
    auto lb_of = [&](float s) -> float {
        float L = qn2h * (1.f - 16.f * KNN_U) + s - eps_s;
        L -= 4.f * KNN_U * (qn2h + fabsf(s));                          // rounding of the two additions above
        if (!(L > 0.f)) return 0.f;                                    // also NaN
        const float t = sqrtf(L) * (1.f - 4.f * KNN_U) - dlt;
        if (!(t > 0.f)) return 0.f;
        return t * t * pv.inv_sig2 * (1.f - 8.f * KNN_U);
    };
    int id = -1; float av = __builtin_inff();
    if (lane < n_cand) { id = cand_idx[(size_t)qi * cand_stride + lane]; av = cand_val[(size_t)qi * cand_stride + lane]; }
    if (!(id >= 0 && id < n_words)) { id = -1; av = __builtin_inff(); }
    const float lb = id >= 0 ? lb_of(av) : __builtin_inff();
    bool done = id < 0;
    unsigned long long key = ~0ull;
    {
        __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
        float* sT = s_terms[threadIdx.x >> 6];
        float kth = __builtin_inff(); int n_eval = 0;
        for (;;) {
            const float cur = done ? __builtin_inff() : lb;
            const float mn = wave_min_f(cur);
            if (!(mn < __builtin_inff())) break;
            if (n_eval >= k && mn * (1.f - pv.ku) > kth) break;        // every remaining candidate is strictly worse than the k-th exact value
            const unsigned long long eq = __ballot(!done && cur == mn);
            const int src = __ffsll((long long)eq) - 1;
            const int cid = __shfl(id, src, 64);
            const float d = wave_functor(ISMHIP_METRIC_L2SQ, qp, words + (size_t)cid * dim_pad, dim, lane, sT);
            if (lane == src) { key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id; done = true; }
            if (++n_eval >= k) {
                unsigned long long kk = key, m_ = ~0ull;
                for (int j = 0; j < k; ++j) { m_ = wave_min_u64(kk); if (kk == m_) kk = ~0ull; }
                kth = __uint_as_float((unsigned)(m_ >> 32));
            }
        }
    }
    float bnd = __builtin_inff();
    if (lane < n_bound) bnd = cand_bound[(size_t)qi * n_bound + lane];
    float dk = 0.f; bool have_k = true;
    for (int j = 0; j < k; ++j) {
        const unsigned long long mn = wave_min_u64(key);
        if (lane == 0) {
            if (mn == ~0ull) { idx_out[(size_t)qi * k + j] = -1; dist_out[(size_t)qi * k + j] = __builtin_nanf(""); }
            else { idx_out[(size_t)qi * k + j] = (int)(mn & 0xffffffffull); dist_out[(size_t)qi * k + j] = __uint_as_float((unsigned)(mn >> 32)); }
        }
        if (mn == ~0ull) have_k = false; else dk = __uint_as_float((unsigned)(mn >> 32));
        if (key == mn) key = ~0ull;
    }
    // a slot whose bound is still +inf dropped nothing -- unless scores overflowed, which needs a non-finite |q^|^2 or bound term
    bool viol = false;
    if (lane < n_bound && (bnd != __builtin_inff() || !(eps_s + qn2h + dlt < __builtin_inff()))) {
        if (!have_k) viol = true;
        else viol = !(dk < lb_of(bnd) * (1.f - pv.ku) - 1e-37f);
    }
    knn_queue_unproven(viol, qi, lane, flag_count, qrec, items);
}

// fast chi-square of one row by a whole wave: every lane owns the 16-byte chunks lane, lane + 64, ... of the (zero padded) rows, all
// of a row's loads are issued before the first is used (a dependent load per element made this 10 us per row), tree sum at the end.
// Differs from the functor's sequential sum by at most ~2 ku relative. dim_pad <= 1344 (checked by the caller): <= 6 chunks per lane.
#define CHI_FAST_CH 6
__device__ __forceinline__ void chi2_fast_load_q(const float* __restrict__ qp, int n4, int lane, f32x4* qv) {
#pragma unroll
    for (int j = 0; j < CHI_FAST_CH; ++j) { const int g = lane + 64 * j; qv[j] = g < n4 ? *(const f32x4*)(qp + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f}; }
}
__device__ __forceinline__ float chi2_fast(const f32x4* qv, const float* __restrict__ wp, int n4, int lane) {
    f32x4 wv[CHI_FAST_CH];
#pragma unroll
    for (int j = 0; j < CHI_FAST_CH; ++j) { const int g = lane + 64 * j; wv[j] = g < n4 ? *(const f32x4*)(wp + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < CHI_FAST_CH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float sm = qv[j][e] + wv[j][e], df = qv[j][e] - wv[j][e]; part += sm > 0.f ? df * df / sm : 0.f; }
    return wave_sum_f(part);
}

// ---- chi-square with candidates from the squared-L2 kernels on the SQUARE-ROOT images (Hellinger lower bound) --------------------
// For non-negative a, b: (a - b)^2 / (a + b) = (sqrt a - sqrt b)^2 (sqrt a + sqrt b)^2 / (a + b) >= (sqrt a - sqrt b)^2, because
// (sqrt a + sqrt b)^2 >= a + b; hence  chi2(q, c) >= H(q, c) = |sqrt q - sqrt c|^2 = |sqrt q|^2 + |sqrt c|^2 - 2 sqrt q . sqrt c:
// a dense contraction, i.e. matrix-core work, where the functor itself (utils/distance.cpp:33-52, a division per element) is not.
// The candidate kernels score |sqrt c|^2 - 2 sqrt c . sqrt q with the error eps_s of the f16 path (VerifyParams, on the sqrt
// vectors), so a row with score s has functor value >= LB(s) (1 - ku), LB(s) = |sqrt q|^2 (1 - 16u) + s - eps_s. As in
// k_knn_rerank_pca the score is a lower bound, not an approximation: candidates are evaluated with the exact functor in ascending
// order of LB until the next one cannot beat the k-th exact value; a slot is proven when LB(its bound) clears that value.
// Only for query batches and codebooks without negative / NaN elements (the caller checks); everything else keeps k_knn_chi2.
__global__ __launch_bounds__(256) void k_knn_rerank_hell(const float* __restrict__ words, int dim, int dim_pad, int n_words,
                                                         const float* __restrict__ q, int nq, int ldq, const float* __restrict__ sq /* sqrt(q), ld dim_pad */,
                                                         const uint32_t* __restrict__ perm /* candidate (shadow) row -> codebook row */,
                                                         const int* __restrict__ cand_idx, const float* __restrict__ cand_val, int cand_stride, int n_cand,
                                                         const float* __restrict__ cand_bound, int n_bound, VerifyParams vp,
                                                         int k, int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                         uint32_t* __restrict__ flag_count, uint32_t* __restrict__ qrec, uint32_t* __restrict__ items) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int lane = lane_id();
    const float* qp = q + (size_t)qi * ldq;
    float qn2 = 0.f;                                                   // |sqrt q|^2
    for (int i = lane; i < dim; i += 64) { const float v = sq[(size_t)qi * dim_pad + i]; qn2 += v * v; }
    qn2 = wave_sum_f(qn2);
    const float eps_s = (17.f * KNN_U * vp.cmax2 + (2.f * vp.dot_rel + 2.f * KNN_U) * sqrtf(qn2 * vp.cmax2) + 2.f * knn_abs_err(vp, qn2) + vp.cn_acc * vp.cmax2) * 1.00001f;
    auto lb_of = [&](float s) -> float {
        float L = qn2 * (1.f - 16.f * KNN_U) + s - eps_s;
        L -= 4.f * KNN_U * (qn2 + fabsf(s));
        return L > 0.f ? L : 0.f;                                      // also NaN -> 0
    };
    // up to 256 candidates (the Hellinger bound is loose by up to a factor of two: dozens to hundreds of rows can lie below the
    // best chi-square value, so the candidate stage runs eight codebook splits): KNN_HELL_CPL per lane, +inf = empty / evaluated
    int id[KNN_HELL_CPL]; float lb[KNN_HELL_CPL], fd[KNN_HELL_CPL]; unsigned long long key[KNN_HELL_CPL];
#pragma unroll
    for (int c = 0; c < KNN_HELL_CPL; ++c) {
        const int j = lane + 64 * c;
        id[c] = -1; lb[c] = __builtin_inff(); fd[c] = __builtin_inff(); key[c] = ~0ull;
        if (j < n_cand) {
            const int x = cand_idx[(size_t)qi * cand_stride + j];
            if (x >= 0 && x < n_words) { id[c] = (int)perm[x]; lb[c] = lb_of(cand_val[(size_t)qi * cand_stride + j]); }
        }
    }
    // Phase 1: a FAST chi-square (lanes sum their elements, wave tree sum) for the candidates in ascending LB order. The functor's own
    // value f (one sequential chain of dim additions, ~5 us per row at 1344 elements) differs from it by at most 3 ku f, so with
    // mg = 4 ku: stop when the next LB exceeds the k-th fast value by (1 + mg); phase 2 then walks the sequential chain only for the
    // rows whose fast value is within (1 + 3 mg) of that k-th value -- nothing else can be among, or tie with, the k best.
    const float mg = 4.f * vp.ku;
    float kth_fast = __builtin_inff();
    {
        // (rows are padded with zeros to dim_pad on the codebook side; the query row is when ldq > dim, else dim is a multiple of 4
        // whenever dim_pad == dim, so whole 16-byte chunks up to dim rounded up to 4 are safe on both sides)
        const int n4 = (ldq >= dim_pad ? dim_pad : dim) >> 2;
        f32x4 qv[CHI_FAST_CH];
        chi2_fast_load_q(qp, n4, lane, qv);
        int n_eval = 0;
        for (;;) {
            float cur = lb[0];
#pragma unroll
            for (int c = 1; c < KNN_HELL_CPL; ++c) cur = fminf(cur, lb[c]);
            const float mn = wave_min_f(cur);
            if (!(mn < __builtin_inff())) break;
            if (n_eval >= k && mn * (1.f - vp.ku) > kth_fast * (1.f + mg)) break;
            const unsigned long long eq = __ballot(cur == mn);
            const int src = __ffsll((long long)eq) - 1;
            int mine = -1, cs = 0;
#pragma unroll
            for (int c = KNN_HELL_CPL - 1; c >= 0; --c) if (lb[c] == mn) { mine = id[c]; cs = c; }
            const int cid = __shfl(mine, src, 64);
            float d = chi2_fast(qv, words + (size_t)cid * dim_pad, n4, lane);
            if (d != d) d = 0.f;                                        // NaN data: keep the row for the exact chain
            if (lane == src) {
#pragma unroll
                for (int c = 0; c < KNN_HELL_CPL; ++c) if (c == cs) { fd[c] = d; lb[c] = __builtin_inff(); }
            }
            if (++n_eval >= k) {
                float ff[KNN_HELL_CPL], m_ = __builtin_inff();
#pragma unroll
                for (int c = 0; c < KNN_HELL_CPL; ++c) ff[c] = fd[c];
                for (int j = 0; j < k; ++j) {
                    float lm = ff[0];
#pragma unroll
                    for (int c = 1; c < KNN_HELL_CPL; ++c) lm = fminf(lm, ff[c]);
                    m_ = wave_min_f(lm);
                    const unsigned long long e2 = __ballot(lm == m_);
                    if (lane == __ffsll((long long)e2) - 1) {            // retire ONE instance
                        bool gone = false;
#pragma unroll
                        for (int c = 0; c < KNN_HELL_CPL; ++c) if (!gone && ff[c] == m_) { ff[c] = __builtin_inff(); gone = true; }
                    }
                }
                kth_fast = m_;
            }
        }
    }
    // Phase 2: the functor's sequential chain for the contenders
    {
        __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
        float* sT = s_terms[threadIdx.x >> 6];
        const float win = kth_fast * (1.f + 3.f * mg);
        for (;;) {
            int mine = -1, cs = 0;
#pragma unroll
            for (int c = KNN_HELL_CPL - 1; c >= 0; --c) if (fd[c] <= win && key[c] == ~0ull && id[c] >= 0) { mine = id[c]; cs = c; }
            const unsigned long long pend = __ballot(mine >= 0);
            if (pend == 0ull) break;
            const int src = __ffsll((long long)pend) - 1;
            const int cid = __shfl(mine, src, 64);
            const float d = wave_functor(ISMHIP_METRIC_CHI2, qp, words + (size_t)cid * dim_pad, dim, lane, sT);
            if (lane == src) {
#pragma unroll
                for (int c = 0; c < KNN_HELL_CPL; ++c) if (c == cs) { key[c] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id[c]; fd[c] = __builtin_inff(); }
            }
        }
    }
    float bnd = __builtin_inff();
    if (lane < n_bound) bnd = cand_bound[(size_t)qi * n_bound + lane];
    float dk = 0.f; bool have_k = true;
    for (int j = 0; j < k; ++j) {
        unsigned long long lm = key[0];
#pragma unroll
        for (int c = 1; c < KNN_HELL_CPL; ++c) lm = key[c] < lm ? key[c] : lm;
        const unsigned long long mn = wave_min_u64(lm);
        if (lane == 0) {
            if (mn == ~0ull) { idx_out[(size_t)qi * k + j] = -1; dist_out[(size_t)qi * k + j] = __builtin_nanf(""); }
            else { idx_out[(size_t)qi * k + j] = (int)(mn & 0xffffffffull); dist_out[(size_t)qi * k + j] = __uint_as_float((unsigned)(mn >> 32)); }
        }
        if (mn == ~0ull) have_k = false; else dk = __uint_as_float((unsigned)(mn >> 32));
#pragma unroll
        for (int c = 0; c < KNN_HELL_CPL; ++c) if (key[c] == mn) key[c] = ~0ull;       // rows are unique among candidates
    }
    bool viol = false;
    if (lane < n_bound && (bnd != __builtin_inff() || !(eps_s + qn2 < __builtin_inff()))) {
        if (!have_k) viol = true;
        else viol = !(dk < lb_of(bnd) * (1.f - vp.ku) - 1e-37f);
    }
    knn_queue_unproven(viol, qi, lane, flag_count, qrec, items);
}

// Exact scan for the slots that could not be proven. One WAVE per (query, slot) work item; the wave handles 4 codeword rows
// per step (16 lanes each, 64-byte coalesced segments) with the query held in registers; direct (a-b)^2 [/(a+b)] sums pick
// the rows that can still matter, the FLANN functor order ranks them. Each item leaves its k best (distance,row) keys in
// item_out; k_knn_fallback_merge folds them into the query's result.
// Slot -> rows: L2 bit b = split*4 + wr*2 + h owns, in every tile of its split, the rows wr*wr_rows + x (x < wr_rows) with bit 2
// of x equal to h (the C/D layout of the 32x32 MFMA tile, see the candidate kernels); chi2 bit b = split owns all rows of its split.
#define KNN_FB_UNITS 8192u      // work units (item x row range) when items are few
__host__ __device__ inline uint32_t knn_fb_parts(uint32_t n_items) {
    if (n_items == 0 || n_items > KNN_FB_UNITS / 2) return 1u;
    const uint32_t p = KNN_FB_UNITS / n_items;
    return p > 256u ? 256u : p;
}
#define KNN_MAX_K 16
#define KNN_FB_MAXJ 84          // dim_pad <= 1344 -> at most 84 elements per lane of a 16-lane row group
template <int KM>   // KM = capacity of the per-item result lists: 4 (k <= 4, every shipped configuration) or KNN_MAX_K
__global__ __launch_bounds__(256) void k_knn_fallback(const float* __restrict__ words, int dim, int dim_pad, int n_words,
                                                      const float* __restrict__ q, int ldq, int metric, int k, int tiles_per_split, int n_tiles,
                                                      int tile_rows, int wr_rows /* rows per wave-row block = MI*32 (L2) */,
                                                      const uint32_t* __restrict__ flag_count, const uint32_t* __restrict__ items,
                                                      const int32_t* __restrict__ idx_in, const float* __restrict__ dist_in,
                                                      unsigned long long* __restrict__ item_out, size_t part_base) {
    __shared__ __attribute__((aligned(16))) float s_terms[4][1344];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, l16 = lane & 15;
    const uint32_t n_items = flag_count[1];
    const bool l2 = metric != ISMHIP_METRIC_CHI2;
    const bool lay_all = l2 && wr_rows == -2;                     // merged splits (k_knn_merge_splits): the one slot owns every row
    const bool lay16 = l2 && wr_rows == -1;                       // k_knn_l2_ring16: slot b = split*8 + wr*4 + fq owns rows wr*128 + 16 m + 4 fq + j
    const bool lay16h = l2 && wr_rows == -3;                      // k_knn_l2_ring16<T, 1>: 128-row tiles, slot b = split*4 + fq owns rows 16 m + 4 fq + j
    const int rows_per_tile = lay_all ? tile_rows : ((lay16 || lay16h) ? 32 : (l2 ? wr_rows / 2 : tile_rows));   // else a lane slot sees half of its wave-row block (bit 2 of the row == h)
    const int nj = dim_pad / 16;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    // With few items a wave per item would leave the chip idle behind a handful of long scans: every item is cut into P row
    // ranges (P * n_items <= KNN_FB_UNITS), each range leaves its own k best in part_out and the merge kernel folds them.
    const uint32_t P = knn_fb_parts(n_items);
    unsigned long long* outp = P > 1 ? item_out + KM * part_base : item_out;
    for (uint32_t u = gw; u < n_items * P; u += nw) {
        const uint32_t it = u / P, part_i = u % P;
        const int qi = (int)items[2 * (size_t)it], b = (int)items[2 * (size_t)it + 1];
        const float* qp = q + (size_t)qi * ldq;
        const int split = lay_all ? 0 : (lay16 ? (b >> 3) : (l2 ? (b >> 2) : b));
        const int wr = lay16h ? 0 : (lay16 ? (b >> 2) & 1 : (b >> 1) & 1), h = b & 1, fq = b & 3;
        const int mt0 = split * tiles_per_split, mt1 = min(n_tiles, mt0 + tiles_per_split);
        const int total = (mt1 - mt0) * rows_per_tile;
        unsigned long long best[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) best[j] = ~0ull;
        float thr = __builtin_inff();
        {
            const int id = idx_in[(size_t)qi * k + (k - 1)];
            if (id >= 0) thr = dist_in[(size_t)qi * k + (k - 1)];
        }
        const int steps = (total + 3) / 4;
        const int e_beg = 4 * (int)((long long)steps * part_i / P), e_end = min(total, 4 * (int)((long long)steps * (part_i + 1) / P));
        for (int e0 = e_beg; e0 < e_end; e0 += 4) {
            const int e = e0 + g;
            int r = n_words;                                  // out of range = idle group
            if (e < e_end) {
                const int tile = mt0 + e / rows_per_tile, y = e % rows_per_tile;
                const int x = lay_all ? y : ((lay16 || lay16h) ? (wr * 128 + ((y >> 2) << 4) + (fq << 2) + (y & 3)) : (l2 ? (wr * wr_rows + (((y >> 2) << 3) | (h << 2) | (y & 3))) : y));
                r = tile * tile_rows + x;
            }
            float part = 0.f;
            if (r < n_words) {
                const float* wp = words + (size_t)r * dim_pad;
                if (!l2) { for (int j = 0; j < nj; ++j) { const int i = l16 + 16 * j; const float a = i < dim ? qp[i] : 0.f, c = wp[i], sm = a + c, df = a - c; part += sm > 0.f ? df * df / sm : 0.f; } }
                else { for (int j = 0; j < nj; ++j) { const int i = l16 + 16 * j; const float df = (i < dim ? qp[i] : 0.f) - wp[i]; part += df * df; } }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            const bool hit = r < n_words && !(part > thr * 1.0001f + 1e-30f);      // NaN-safe: unordered compares count as hits
            unsigned long long hm = __ballot(hit && l16 == 0);
            while (hm) {                                                             // rare
                const int src = __ffsll((long long)hm) - 1; hm &= hm - 1;
                const int rr = __shfl(r, src, 64);
                const float d = wave_functor(metric, qp, words + (size_t)rr * dim_pad, dim, lane, s_terms[threadIdx.x >> 6]);
                unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)rr;
                if (d != d) key = (0x7fc00000ull << 32) | (unsigned)rr;             // NaN sorts after every finite distance and keeps ITS row
#pragma unroll
                for (int j = 0; j < KM; ++j) if (j < k && key < best[j]) { const unsigned long long tmp = best[j]; best[j] = key; key = tmp; }
                unsigned long long kth = ~0ull;
#pragma unroll
                for (int j = 0; j < KM; ++j) if (j == k - 1) kth = best[j];
                if (kth != ~0ull) thr = fminf(thr, __uint_as_float((unsigned)(kth >> 32)));
            }
        }
#pragma unroll
        for (int j = 0; j < KM; ++j) if (lane == j) outp[KM * (size_t)u + j] = best[j];
    }
}

// one wave per unproven query: the lanes fold the per-unit results (and the re-ranked candidates) into private sorted lists, k
// rounds of a wave-wide minimum then pick the result; a row reached through two paths is taken once
template <int KM>
__global__ __launch_bounds__(256) void k_knn_fallback_merge(int k, const uint32_t* __restrict__ flag_count, const uint32_t* __restrict__ qrec,
                                     const unsigned long long* __restrict__ item_out, size_t part_base, int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const uint32_t n_q = flag_count[0];
    const uint32_t P = knn_fb_parts(flag_count[1]);
    const unsigned long long* outp = P > 1 ? item_out + KM * part_base : item_out;
    const int lane = threadIdx.x & 63;
    for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n_q; t += gridDim.x * 4) {
        const int qi = (int)qrec[3 * (size_t)t]; const uint32_t ibase = qrec[3 * (size_t)t + 1], ni = qrec[3 * (size_t)t + 2];
        unsigned long long fin[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) fin[j] = ~0ull;
        auto ins = [&](unsigned long long key) {
#pragma unroll
            for (int j = 0; j < KM; ++j) if (fin[j] != ~0ull && (fin[j] & 0xffffffffull) == (key & 0xffffffffull)) return;   // same row twice
#pragma unroll
            for (int j = 0; j < KM; ++j) if (key < fin[j]) { const unsigned long long tmp = fin[j]; fin[j] = key; key = tmp; }
        };
        if (lane < k) {
            const int id = idx_out[(size_t)qi * k + lane];
            if (id >= 0) ins(((unsigned long long)__float_as_uint(dist_out[(size_t)qi * k + lane]) << 32) | (unsigned)id);
        }
        for (uint32_t i = lane; i < ni * P; i += 64)
            for (int j = 0; j < k; ++j) { const unsigned long long key = outp[KM * ((size_t)ibase * P + i) + j]; if (key != ~0ull) ins(key); }
        for (int j = 0; j < k; ++j) {
            unsigned long long mn = fin[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; }
            if (lane == 0) {
                if (mn == ~0ull) { idx_out[(size_t)qi * k + j] = -1; dist_out[(size_t)qi * k + j] = __builtin_nanf(""); }
                else { idx_out[(size_t)qi * k + j] = (int)(mn & 0xffffffffull); dist_out[(size_t)qi * k + j] = __uint_as_float((unsigned)(mn >> 32)); }
            }
            if (mn == ~0ull) continue;
            // drop the chosen row from every private list (it can sit in several lanes, with the same key)
            // (a private list is sorted and holds a row at most once: shift the tail down over the hit)
            bool gone = false;
#pragma unroll
            for (int x = 0; x < KM; ++x) {
                if (!gone && fin[x] != ~0ull && (fin[x] & 0xffffffffull) == (mn & 0xffffffffull)) gone = true;
                if (gone) fin[x] = x + 1 < KM ? fin[x + 1] : ~0ull;
            }
        }
    }
}

__global__ void k_pad_rows(const float* __restrict__ src, int n, int dim, float* __restrict__ dst, int dim_pad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * dim_pad) return;
    const int row = (int)(i / dim_pad), col = (int)(i % dim_pad);
    dst[i] = col < dim ? src[(size_t)row * dim + col] : 0.f;
}

__global__ void k_ratio(int nq, float thr, const int32_t* __restrict__ idx2, const float* __restrict__ d2,
                        int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int id = idx2[i * 2]; const float a = d2[i * 2], b = d2[i * 2 + 1];
    if (idx2[i * 2 + 1] >= 0 && a / b > thr) id = -1;        // activation_strategy_knn.h:77-84
    idx_out[i] = id; dist_out[i] = a;
}

// ActivationStrategyKnnRule::activateKNN, detection branch (activation_strategy_knn_rule.h:79-118)
__global__ void k_rule(int nq, float thr, const int32_t* __restrict__ idx3, const float* __restrict__ d3, const uint32_t* __restrict__ word_class,
                       int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int i0 = idx3[i * 3], i1 = idx3[i * 3 + 1], i2 = idx3[i * 3 + 2];
    const float a = d3[i * 3], b = d3[i * 3 + 1], c = d3[i * 3 + 2];
    int id = -1; float dd = __builtin_nanf("");
    if (i2 < 0) {                       // fewer than 4 codewords: the reference returns all of them; report the nearest
        if (i0 >= 0) { id = i0; dd = a; }
    } else {
        const uint32_t c0 = word_class[i0], c1 = word_class[i1], c2 = word_class[i2];
        if (c0 == c1 && c0 == c2) { id = i0; dd = a; }
        else if (c0 == c1 && c0 != c2) { if (a / c < thr) { id = i0; dd = a; } }
        else if (c0 != c1 && c1 == c2) { if (a / b >= thr) { id = i1; dd = b; } }
        else if (c0 != c1 && c1 != c2) { if (a / b < thr) { id = i0; dd = a; } }
    }
    idx_out[i] = id; dist_out[i] = dd;
}

// stage1 != nullptr: FIRST stage of the two-stage search -- candidates + exact re-rank + proof only; the unproven queries are
// left in the queue (*stage1 = {flag_count, qrec}) for the caller instead of going to the exact scan. tname: timer of the
// candidate kernel.
struct KnnStage1 { uint32_t* flag_count; uint32_t* qrec; };
template <int T>
int run_knn(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q, int k,
            int32_t* idx_out, float* dist_out, KnnStage1* stage1 = nullptr, const char* tname = nullptr, bool many_splits = false, int use_pca = 0 /* 1: stage-1 image, 2: stage-2 image */, const float* hell_q = nullptr) {
    // hell_q != nullptr (chi-square only): candidates come from the squared-L2 kernels run on the SQUARE-ROOT images (Hellinger lower
    // bound, see k_knn_rerank_lb): xb = the shadow codebook that owns those images, hell_q = sqrt(q) rows of dim_pad floats
    const bool hell = hell_q != nullptr && metric == ISMHIP_METRIC_CHI2 && cb->chi_shadow;
    const ismhip_codebook* xb = hell ? cb->chi_shadow : cb;
    const int cmetric = hell ? ISMHIP_METRIC_L2SQ : metric;
    if (cb->dim_pad / 16 > KNN_FB_MAXJ) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "knn: descriptor longer than 1344 not built");
    const float* qq = q; int ldq = cb->dim;
    if (cb->dim_pad != cb->dim) {
        float* qpad = (float*)ism_scratch(ctx, SCR_QPAD, (size_t)nq * cb->dim_pad * sizeof(float));
        if (!qpad) return ISMHIP_ERR_NOMEM;
        const size_t tot = (size_t)nq * cb->dim_pad;
        hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, q, nq, cb->dim, qpad, cb->dim_pad);
        ISM_CHECK_LAUNCH(ctx, "k_pad_rows");
        qq = qpad; ldq = cb->dim_pad;
    }
    int n_splits, cand_per_split, n_cand, tiles_per_split;
    // candidate kernel for squared L2: f16 (default), bf16x3 (ISMHIP_KNN_MODE=bf16x3) or the exact-f32 MFMA contraction
    // (ISMHIP_KNN_MODE=f32); the last two are kept for A/B runs and as the reference points of the error model tests
    // short descriptors (FPFH-33: values up to 100, |q||c| ~ 1e4): the f16 error bound is of the order of the neighbour distances, most
    // proofs fail and the exact scan takes over (measured: 135 ms of scan per 524288 queries). The exact-f32 MFMA contraction costs
    // 2 Nq Nc D flop at ~125 TFLOP/s, which for D <= 64 is cheaper than the 16-bit kernels' fixed overheads -- and it proves everything.
    const PcaImage& PI = use_pca == 2 ? cb->pca2 : cb->pca;
    const bool short_dim = cb->dim <= 64 && ctx->knn_mode == 0 && !(use_pca && PI.m > 0);     // (stage 1 of a short-descriptor codebook with an f16 stage-1 image: pca.hip)
    const int mode = cmetric != ISMHIP_METRIC_L2SQ ? -1 : (hell ? 0 : (short_dim ? 2 : (ctx->knn_mode == 0 && cb->words_f16 ? 0 : (ctx->knn_mode <= 1 && cb->words_bf16_hi ? 1 : 2))));
    const bool use_lp = mode == 0 || mode == 1;
    const bool big_tile = use_lp && nq >= 4096 && cb->n_words_pad >= 4096 && !ctx->knn_small_tile;      // 256x256 tile, 8 waves
    const int BM0 = cmetric == ISMHIP_METRIC_L2SQ ? (big_tile ? 256 : KNN_BM) : CHI_B;
    int BNq = cmetric == ISMHIP_METRIC_L2SQ ? (big_tile ? 256 : KNN_BN) : CHI_B;
    const int wr_rows = big_tile ? 128 : 64;
    // (the ring kernels prefetch four slices ahead and keep the |c|^2 rows of four tiles: a tile must have at least two slices)
    const bool use_ring = big_tile && mode == 0 && !ctx->knn_no_ring && xb->words_f16t && (cb->dim + 15) / 16 > 2;
    const bool ring16 = use_ring && !ctx->knn_ring32;                  // 16x16x32 MFMA shape: 8 lane slots per query and split instead of 4
    const bool half = ring16 && ctx->knn_half;                         // 128 x 256 tile, two workgroups per CU (k_knn_l2_ring16<T, 1>)
    const bool qpanel = ring16 && !half && ctx->knn_qpanel && ((cb->dim + 15) / 16 + 1) / 2 <= 11;   // 256 x 128 tile, query panel resident in LDS
    if (qpanel) BNq = 128;
    // 256 x 256 tile with the whole query panel resident (k_knn_l2_ring16<T, 2, 0, 2>): stage 1 on a rotated image of <= 160 coordinates
    const bool qpanel2 = ring16 && !half && !qpanel && ctx->knn_qpanel2 && use_pca && PI.m > 0 && PI.m <= 160;
    const int BM = half ? 128 : BM0;
    const int slots = ring16 && !half ? 8 : 4;
    // stage 1 of the two-stage search on the rotated, truncated image (pca.hip): same kernel, pca_m / 32 slices instead of dim / 32
    const bool pca = use_pca && use_ring && PI.m > 0;
    const int ring_nk = pca ? PI.m / 32 : ((cb->dim + 15) / 16 + 1) / 2;   // 32-k slices per row in the tiled images
    const bool merged = many_splits && cmetric == ISMHIP_METRIC_L2SQ && use_lp && !big_tile;
    if (cmetric == ISMHIP_METRIC_L2SQ) {
        const int n_qt = (nq + BNq - 1) / BNq, n_mt = cb->n_words_pad / BM;
        const int max_s = (hell ? 64 * KNN_HELL_CPL : 64) / (slots * T);
        // at least three codebook splits (two when the candidate slots allow no more): with one, the 32 workgroups of an XCD hold 32
        // different query tiles (6 MB of f16 queries re-read per codeword tile) and fall out of its 4 MB L2; two splits halve that
        // working set (measured 21.0 -> 19.9 ms at 262144 queries), three cost the same time as two and fetch a fifth less from
        // beyond the L2 (joined streams); four are 1.5 % slower
        // (stage 1 on the rotated image: two -- with 4-6 slices per tile the epilogue is a larger share of the kernel, and every split
        // is another eight candidate lists per query to fill: 29.8 -> 28.1 ms per bench launch)
        n_splits = std::max(1, std::min(std::min(max_s, n_mt), std::max(pca ? 2 : 3, (1024 + n_qt - 1) / n_qt)));
        // ... as long as a workgroup still has a few dozen tiles to amortise its prologue over (10 k-word codebook, 40 tiles: 1 / 2 / 3
        // splits = 3.81 / 4.18 / 4.51 ms) and the launch fills the chip without them
        if (big_tile && n_qt >= 512) n_splits = std::min(n_splits, std::max(1, n_mt / 32));
        // Hellinger candidates for chi-square: as many splits as the candidate slots allow (up to eight, at least four tiles each) --
        // the bound of a proof has to lie beyond EVERY row whose Hellinger distance is below the best chi-square value, and
        // those are dozens to hundreds (CSHOT-1344, 10 k words, measured: median 28, 90th percentile 131, 99th 352)
        if (hell) n_splits = std::max(1, std::min(std::min(max_s, 8), n_mt / 4));
        if (ctx->knn_splits > 0) n_splits = std::max(1, std::min(std::min(max_s, n_mt), ctx->knn_splits));
        // few queries (stage 2 of the two-stage search): cut the codebook into as many splits as it takes to fill the chip; the
        // candidates of all splits are then folded into one slot of KNN_MERGE_KEEP by k_knn_merge_splits
        if (merged) n_splits = std::max(1, std::min(n_mt, (1024 + 8 * ((n_qt + 7) / 8) - 1) / (8 * ((n_qt + 7) / 8))));
        tiles_per_split = (n_mt + n_splits - 1) / n_splits;
        n_splits = (n_mt + tiles_per_split - 1) / tiles_per_split;
        cand_per_split = slots * T;
    } else {
        const int n_qt = (nq + CHI_B - 1) / CHI_B, n_mt = cb->n_words_pad / CHI_B;
        const int max_s = 64 / T;
        n_splits = std::max(1, std::min(std::min(max_s, n_mt), (2048 + n_qt - 1) / n_qt));
        if (k > T) n_splits = std::max(n_splits, std::min(std::min(max_s, n_mt), (2 * k + T - 1) / T));    // at least 2k candidates to re-rank
        tiles_per_split = (n_mt + n_splits - 1) / n_splits;
        n_splits = (n_mt + tiles_per_split - 1) / tiles_per_split;
        cand_per_split = T;
    }
    n_cand = n_splits * cand_per_split;
    int n_bound = cmetric == ISMHIP_METRIC_L2SQ ? n_splits * slots : n_splits;
    float* cand_val = (float*)ism_scratch(ctx, SCR_KNN_CAND_VAL, (size_t)nq * (n_cand + n_bound + (merged ? KNN_MERGE_KEEP + 1 : 0)) * sizeof(float));
    int* cand_idx = (int*)ism_scratch(ctx, SCR_KNN_CAND_IDX, (size_t)nq * (n_cand + (merged ? KNN_MERGE_KEEP : 0)) * sizeof(int));
    // queue of unproven work: 16 counters | query records [nq*3] | items [nq*n_bound*2] | item results [nq*n_bound*4] u64
    const size_t q_items = (size_t)nq * n_bound;
    uint32_t* flags = (uint32_t*)ism_scratch(ctx, SCR_KNN_FLAGS, (16 + 3 * (size_t)nq + 2 * q_items) * sizeof(uint32_t) + 8 + (q_items + KNN_FB_UNITS) * (k > 4 ? KNN_MAX_K : 4) * sizeof(unsigned long long));
    if (!cand_val || !cand_idx || !flags) return ISMHIP_ERR_NOMEM;
    float* cand_bound = cand_val + (size_t)nq * n_cand;
    uint32_t* flag_count = flags; uint32_t* qrec = flags + 16; uint32_t* items = qrec + 3 * (size_t)nq;
    unsigned long long* item_out = (unsigned long long*)(((uintptr_t)(items + 2 * q_items) + 7) & ~(uintptr_t)7);
    ISM_HIP(ctx, hipMemsetAsync(flag_count, 0, 64, ctx->stream));
    uint32_t* qsc = flag_count + 4;      // f16 mode: [0] absmax bits of the query batch, [1] -2/(s_q s_c), [2] 2^-14/s_q
    u16 *q_hi = nullptr, *q_lo = nullptr;
    if (use_lp) {
        const int nq_pad = use_ring ? (nq + 255) / 256 * 256 : (nq + BNq - 1) / BNq * BNq;
        const size_t tot = use_ring ? (size_t)(nq_pad / 256) * ring_nk * 8192 : (size_t)nq_pad * cb->ld16;
        q_hi = (u16*)ism_scratch(ctx, SCR_KNN_QSPLIT, tot * 2 * sizeof(u16));
        if (!q_hi) return ISMHIP_ERR_NOMEM;
        q_lo = q_hi + tot;
        if (pca) {
            TimerScope tr(ctx, "knn_rotate");
            const int rc = ism_pca_rotate_queries(ctx, cb, &PI, qq, nq, ldq, q_hi);
            if (rc != ISMHIP_OK) return rc;
            ++ctx->knn_pca_launches;
        } else if (mode == 0) {
            const float* cq = hell ? hell_q : qq; const int cldq = hell ? cb->dim_pad : ldq;     // Hellinger: the images are made from sqrt(q)
            hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, ctx->stream, cq, nq, cb->dim, cldq, qsc);
            ISM_CHECK_LAUNCH(ctx, "k_absmax");
            if (use_ring) hipLaunchKernelGGL(k_to_f16_tiled, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, cq, nq, cb->dim, cldq, nq_pad / 256, ring_nk, qsc, xb->f16_scale, q_hi);
            else hipLaunchKernelGGL(k_to_f16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, cq, nq, cb->dim, cldq, nq_pad, cb->ld16, qsc, xb->f16_scale, q_hi);
            ISM_CHECK_LAUNCH(ctx, "k_to_f16");
        } else {
            hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, qq, nq, cb->dim, ldq, nq_pad, cb->ld16, q_hi, q_lo);
            ISM_CHECK_LAUNCH(ctx, "k_split_bf16");
        }
    }
    {
        TimerScope ts(ctx, tname ? tname : (metric == ISMHIP_METRIC_L2SQ ? "knn_l2_mfma" : "knn_chi2"));      // chi-square: whichever kernel makes its candidates
        if (use_lp) {
            const int n_qt = (nq + BNq - 1) / BNq;
            const dim3 grid(8 * ((n_qt + 7) / 8) * n_splits);
            const u16* wh = mode == 0 ? xb->words_f16 : xb->words_bf16_hi;
            const u16* wl = mode == 0 ? nullptr : xb->words_bf16_lo;
            const int nterm = mode == 0 ? 1 : 3;
            const int kb = (nterm == 1 && !ctx->knn_kb32) ? 64 : 32;       // the two bf16x3 images only fit LDS with 32-deep slices
            const size_t lds = (size_t)2 * (BM + BNq) * kb * sizeof(u16) * (nterm == 3 ? 2 : 1) + BM * sizeof(float);
            const void* kern;
            int ai;
            if (big_tile) {
                if (nterm == 3) { kern = (const void*)k_knn_l2_mfma16<T, 2, 4, 4, 2, 3, 32>; ai = 0; }
                else if (kb == 64) { kern = (const void*)k_knn_l2_mfma16<T, 2, 4, 4, 2, 1, 64>; ai = 1; }
                else { kern = (const void*)k_knn_l2_mfma16<T, 2, 4, 4, 2, 1, 32>; ai = 2; }
            } else {
                if (nterm == 3) { kern = (const void*)k_knn_l2_mfma16<T, 2, 2, 2, 2, 3, 32>; ai = 3; }
                else if (kb == 64) { kern = (const void*)k_knn_l2_mfma16<T, 2, 2, 2, 2, 1, 64>; ai = 4; }
                else { kern = (const void*)k_knn_l2_mfma16<T, 2, 2, 2, 2, 1, 32>; ai = 5; }
            }
            if (use_ring) {
                wh = xb->words_f16t;
                const void* rk = ring16 ? (qpanel2 ? (const void*)k_knn_l2_ring16<T, 2, 0, 2> : qpanel ? (const void*)k_knn_l2_ring16<T, 2, 0, 1> : half ? (const void*)k_knn_l2_ring16<T, 1, 0> : (const void*)k_knn_l2_ring16<T, 2, 0>) : (const void*)k_knn_l2_ring<T, 0>;
#ifdef ISM_KNN_DBG_VARIANTS
                if (ring16) switch (ctx->knn_dbg) {       // 1 no epilogue, 2 no MFMA, 4 no DMA, 16 no fragment reads, 32 no barrier, 64 pre-test only, 256 counters
                    case 1: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 1> : (const void*)k_knn_l2_ring16<T, 2, 1>; break;  case 2: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 2> : (const void*)k_knn_l2_ring16<T, 2, 2>; break;
                    case 5: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 5> : (const void*)k_knn_l2_ring16<T, 2, 5>; break;  case 21: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 21> : (const void*)k_knn_l2_ring16<T, 2, 21>; break;
                    case 53: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 53> : (const void*)k_knn_l2_ring16<T, 2, 53>; break; case 64: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 64> : (const void*)k_knn_l2_ring16<T, 2, 64>; break;
                    case 256: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 256> : (const void*)k_knn_l2_ring16<T, 2, 256>; break;
                    case 2048: rk = (const void*)k_knn_l2_ring16<T, 2, 2048>; break; case 2049: rk = (const void*)k_knn_l2_ring16<T, 2, 2049>; break; case 2069: rk = (const void*)k_knn_l2_ring16<T, 2, 2069>; break;
                    case 128: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 128> : (const void*)k_knn_l2_ring16<T, 2, 128>; break; case 1024: rk = half ? (const void*)k_knn_l2_ring16<T, 1, 1024> : (const void*)k_knn_l2_ring16<T, 2, 1024>; break;
                    default: break;
                } else
                switch (ctx->knn_dbg) {
                    case 1: rk = (const void*)k_knn_l2_ring<T, 1>; break;  case 3: rk = (const void*)k_knn_l2_ring<T, 3>; break;
                    case 5: rk = (const void*)k_knn_l2_ring<T, 5>; break;  case 7: rk = (const void*)k_knn_l2_ring<T, 7>; break;
                    case 9: rk = (const void*)k_knn_l2_ring<T, 9>; break;  case 13: rk = (const void*)k_knn_l2_ring<T, 13>; break;
                    case 15: rk = (const void*)k_knn_l2_ring<T, 15>; break; case 11: rk = (const void*)k_knn_l2_ring<T, 11>; break;
                    case 21: rk = (const void*)k_knn_l2_ring<T, 21>; break; case 513: rk = (const void*)k_knn_l2_ring<T, 513>; break; case 256: rk = (const void*)k_knn_l2_ring<T, 256>; break; case 64: rk = (const void*)k_knn_l2_ring<T, 64>; break; case 128: rk = (const void*)k_knn_l2_ring<T, 128>; break; case 192: rk = (const void*)k_knn_l2_ring<T, 192>; break; case 53: rk = (const void*)k_knn_l2_ring<T, 53>; break; case 37: rk = (const void*)k_knn_l2_ring<T, 37>; break;
                    default: break;
                }
#endif
                const size_t rlds = qpanel2 ? (size_t)4 * 256 * RG_KB * sizeof(u16) + 4 * 256 * sizeof(float) + 8 * 4 * 64 * sizeof(float) + (size_t)ring_nk * 256 * RG_KB * sizeof(u16)
                                  : qpanel ? (size_t)4 * 256 * RG_KB * sizeof(u16) + 4 * 256 * sizeof(float) + 8 * 2 * 64 * sizeof(float) + (size_t)ring_nk * 128 * RG_KB * sizeof(u16)
                                  : half ? (size_t)3 * (128 + 256) * RG_KB * sizeof(u16) + 4 * 256 * sizeof(float)
                                         : (size_t)RG_STAGES * RG_STAGE_HALVES * sizeof(u16) + 4 * RG_BM * sizeof(float) + 8 * 4 * 64 * sizeof(float);
                if (ctx->knn_dbg || !ctx->attr_done.count(rk)) {          // per device, so remembered per ctx
                    ISM_HIP(ctx, hipFuncSetAttribute(rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
                    ctx->attr_done.insert(rk);
                }
                const float* osc = (const float*)(qsc + 1);
                const float* word_norm;
                if (pca) { wh = PI.f16t; osc = PI.osc; word_norm = PI.cn_scaled; }      // scales fixed per codebook: the C operand is precomputed
                else {
                    float* cn_scaled = (float*)ism_scratch(ctx, SCR_QNORM2, ((size_t)cb->n_words_pad + 256) * sizeof(float));   // the |c|^2 DMA of a 128-row tile reads 256 floats
                    if (!cn_scaled) return ISMHIP_ERR_NOMEM;
                    hipLaunchKernelGGL(k_scale_norms, dim3((cb->n_words_pad + 255) / 256), dim3(256), 0, ctx->stream, xb->word_norm, cb->n_words_pad, osc, cn_scaled);
                    ISM_CHECK_LAUNCH(ctx, "k_scale_norms");
                    word_norm = cn_scaled;
                }
                int n_tiles_m = cb->n_words_pad / BM, ld16 = cb->ld16, k_steps = pca ? PI.m / 16 : (cb->dim + 15) / 16, nq_ = nq, tps = tiles_per_split, nsp = n_splits, ncand = n_cand, nb = n_bound;
                const u16* qh_ = q_hi;
                unsigned int* clock = nullptr;                                   // joined codeword streams (k_knn_l2_ring16), one clock per (XCD, split)
                if (ring16 && ctx->knn_join) {
                    clock = (unsigned int*)ism_scratch(ctx, SCR_KNN_CLOCK, 8 * 64 * sizeof(unsigned int));
                    if (!clock) return ISMHIP_ERR_NOMEM;
                    ISM_HIP(ctx, hipMemsetAsync(clock, 0, 8 * 64 * sizeof(unsigned int), ctx->stream));
                }
                // sampling pre-pass (stage 1 of the two-stage search on the 256 x 256 kernel, codebooks of >= 128 tiles): the best score
                // every query meets in every 16th codeword tile becomes the start threshold of all its lane slots (see the kernel)
                const float* thr_init = nullptr; float* thr_out = nullptr; int tile_step = 1;
                // (not for an untruncated stage-1 image: with nothing left out there is no scale to relax the start value by, and the
                // unrelaxed best of the sample is the nearest neighbour itself too often)
                if (stage1 && ring16 && !half && !qpanel && ctx->knn_prepass && n_tiles_m >= 128 && ctx->knn_dbg == 0 && !(pca && PI.resid2 <= 0.f)) {
                    float* thr0 = (float*)ism_scratch(ctx, SCR_KNN_THR0, (size_t)((nq + 255) / 256 * 256) * sizeof(float));
                    if (!thr0) return ISMHIP_ERR_NOMEM;
                    const void* pk = (const void*)k_knn_l2_ring16<T, 2, 0, 0, 1>;
                    if (!ctx->attr_done.count(pk)) { ISM_HIP(ctx, hipFuncSetAttribute(pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds)); ctx->attr_done.insert(pk); }
                    int one = 1, all = n_tiles_m, step = ctx->knn_pre_step; unsigned int* noclk = nullptr; const float* noinit = nullptr;
                    // relaxation: gamma x the second moment the truncation leaves out (codeword + query side, taken as equal), in
                    // accumulator units (score / out_scale); the original image truncates nothing
                    float relax = 0.f;
                    if (pca) relax = -ctx->knn_pre_gamma * 2.0f * PI.resid2 * (PI.sq * PI.sc * 0.5f);
                    void* pargs[] = {&wh, &word_norm, &n_tiles_m, &ld16, &k_steps, &qh_, &nq_, &osc, &all, &one, &cand_val, &cand_idx, &ncand, &cand_bound, &nb, &noclk, &noinit, &thr0, &step, &relax};
                    ISM_HIP(ctx, hipLaunchKernel(pk, dim3(8 * ((n_qt + 7) / 8)), dim3(512), pargs, rlds, ctx->stream));
                    ISM_CHECK_LAUNCH(ctx, "k_knn_l2_ring16<pre>");
                    thr_init = thr0;
                }
                float no_relax = 0.f;
                void* rargs[] = {&wh, &word_norm, &n_tiles_m, &ld16, &k_steps, &qh_, &nq_, &osc, &tps, &nsp, &cand_val, &cand_idx, &ncand, &cand_bound, &nb, &clock, &thr_init, &thr_out, &tile_step, &no_relax};
                ISM_HIP(ctx, hipLaunchKernel(rk, grid, dim3(half ? 256 : 512), rargs, rlds, ctx->stream));
                ISM_CHECK_LAUNCH(ctx, "k_knn_l2_ring");
            } else {
            (void)ai;
            if (!ctx->attr_done.count(kern)) { ISM_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ctx->attr_done.insert(kern); }
            const float* word_norm = xb->word_norm; const float* osc = (const float*)(qsc + 1);
            int n_tiles_m = cb->n_words_pad / BM, ld16 = cb->ld16, k_steps = (cb->dim + 15) / 16, nq_ = nq, tps = tiles_per_split, nsp = n_splits, ncand = n_cand, nb = n_bound;
            const u16* qh_ = q_hi; const u16* ql_ = q_lo;
            const float* no_tau = nullptr; uint32_t* no_u = nullptr; int no_cap = 0;
            void* args[] = {&wh, &wl, &word_norm, &n_tiles_m, &ld16, &k_steps, &qh_, &ql_, &nq_, &osc, &tps, &nsp, &cand_val, &cand_idx, &ncand, &cand_bound, &nb, &no_tau, &no_u, &no_u, &no_cap};
            ISM_HIP(ctx, hipLaunchKernel(kern, grid, dim3(big_tile ? 512 : 256), args, lds, ctx->stream));
            ISM_CHECK_LAUNCH(ctx, "k_knn_l2_mfma16");
            }
        } else if (cmetric == ISMHIP_METRIC_L2SQ) {
            const int n_qt = (nq + KNN_BN - 1) / KNN_BN;
            hipLaunchKernelGGL(k_knn_l2_mfma<T>, dim3(8 * ((n_qt + 7) / 8) * n_splits), dim3(256), 0, ctx->stream, cb->words, cb->word_norm,
                               cb->n_words_pad / KNN_BM, cb->dim_pad, qq, nq, ldq, tiles_per_split, n_splits, cand_val, cand_idx, n_cand,
                               cand_bound, n_bound, std::min(16, cb->dim - (cb->dim_pad / KNN_BK - 1) * KNN_BK));
            ISM_CHECK_LAUNCH(ctx, "k_knn_l2_mfma");
        } else {
            const int n_qt = (nq + CHI_B - 1) / CHI_B;
            uint32_t* q_negative = flag_count + 12;                 // zeroed with the counters above
            if (cb->words_nonneg) {
                hipLaunchKernelGGL(k_any_negative, dim3(512), dim3(256), 0, ctx->stream, qq, nq, cb->dim, ldq, q_negative);
                ISM_CHECK_LAUNCH(ctx, "k_any_negative");
            }
            hipLaunchKernelGGL(k_knn_chi2<T>, dim3(n_qt, n_splits), dim3(256), 0, ctx->stream, cb->words, cb->n_words_pad, cb->dim_pad,
                               qq, nq, ldq, cb->words_nonneg ? 1 : 0, q_negative, tiles_per_split, cand_val, cand_idx, n_cand, cand_bound, n_bound);
            ISM_CHECK_LAUNCH(ctx, "k_knn_chi2");
        }
    }
    if (merged) {
        float* m_val = cand_bound + (size_t)nq * n_bound; float* m_bound = m_val + (size_t)nq * KNN_MERGE_KEEP;
        int* m_idx = cand_idx + (size_t)nq * n_cand;
        hipLaunchKernelGGL(k_knn_merge_splits, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, nq, cand_val, cand_idx, n_cand, cand_bound, n_bound, m_val, m_idx, m_bound);
        ISM_CHECK_LAUNCH(ctx, "k_knn_merge_splits");
        cand_val = m_val; cand_idx = m_idx; cand_bound = m_bound; n_cand = KNN_MERGE_KEEP; n_bound = 1;
    }
    VerifyParams vp;
    vp.ku = 1.01f * (float)cb->dim_pad * KNN_U;
    // relative part of the candidate kernel's dot error (see the kernels): representation + accumulation (<= 2^-23 per add, any order)
    vp.dot_rel = mode == 0 ? (2.002f * 4.8828125e-04f + 1.01f * (float)cb->dim_pad * 1.1920929e-07f)
               : mode == 1 ? (3.1f * 1.52587890625e-05f + 1.01f * 3.f * (float)cb->dim_pad * 1.1920929e-07f) : vp.ku;
    vp.cmax2 = xb->max_norm2;
    vp.dabs_c = mode == 0 ? 6.103515625e-05f / xb->f16_scale : 0.f;
    vp.dabs_q = mode == 0 ? (const float*)(qsc + 2) : nullptr;
    vp.sqrt_dim = sqrtf((float)cb->dim_pad);
    vp.cn_acc = mode == 0 ? 1.01f * (float)(cb->dim_pad + 1) * 1.1920929e-07f : 0.f;
    {
    TimerScope trr(ctx, "knn_rerank");
    if (pca) {
        PcaVerify pv;
        const float acc_rel = 1.01f * (float)PI.m * 1.1920929e-07f;            // accumulation only: products of f16 values are exact in fp32
        pv.qimg = q_hi; pv.nk = ring_nk; pv.inv_sq2 = 1.0f / (PI.sq * PI.sq);
        pv.inv_sig2 = PI.inv_sig2; pv.d_rel = PI.d_rel; pv.dq_abs = PI.dq_abs;
        pv.dc = (PI.d_rel * sqrtf(cb->max_norm2) + PI.dc_abs) * 1.00001f;
        pv.cmax2 = PI.cmax2;
        // subnormal f16 operands may be flushed to zero by the matrix cores: |dq_i| <= 2^-14 / sq, |dc_i| <= 2^-14 / sc per element
        const float fq = 6.103515625e-05f / PI.sq, fc = 6.103515625e-05f / PI.sc, sm = sqrtf((float)PI.m);
        pv.eps_c2 = (17.f * KNN_U + 1.01f * (float)(PI.m + 1) * 1.1920929e-07f) * PI.cmax2 + 2.02f * (sm * fq * sqrtf(PI.cmax2) + sm * sm * fq * fc);
        pv.dot2 = 2.f * acc_rel + 2.f * KNN_U + 2.02f * sm * fc / sqrtf(PI.cmax2);
        pv.ku = vp.ku;
        hipLaunchKernelGGL(k_knn_rerank_pca, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, cb->words, cb->dim, cb->dim_pad, cb->n_words,
                           qq, nq, ldq, cand_idx, cand_val, n_cand, n_cand, cand_bound, n_bound, pv, k, idx_out, dist_out, flag_count, qrec, items);
    } else if (hell) {
        hipLaunchKernelGGL(k_knn_rerank_hell, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, cb->words, cb->dim, cb->dim_pad, cb->n_words,
                           qq, nq, ldq, hell_q, xb->shadow_perm, cand_idx, cand_val, n_cand, n_cand, cand_bound, n_bound, vp, k, idx_out, dist_out, flag_count, qrec, items);
    } else
    hipLaunchKernelGGL(k_knn_rerank, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, cb->words, cb->dim, cb->dim_pad, cb->n_words,
                       qq, nq, ldq, metric, cand_idx, cand_val, n_cand, n_cand, cand_bound, n_bound, vp, k, idx_out, dist_out, flag_count, qrec, items);
    ISM_CHECK_LAUNCH(ctx, "k_knn_rerank");
    }
    if (stage1) { stage1->flag_count = flag_count; stage1->qrec = qrec; return ISMHIP_OK; }
    {
        TimerScope ts(ctx, "knn_fallback");
        const int fb_tps = merged ? cb->n_words_pad / BM : tiles_per_split, fb_lay = merged ? -2 : (half ? -3 : (ring16 ? -1 : wr_rows));
        if (k <= 4) hipLaunchKernelGGL(k_knn_fallback<4>, dim3(1024), dim3(256), 0, ctx->stream, cb->words, cb->dim, cb->dim_pad, cb->n_words, qq, ldq, metric, k,
                                       fb_tps, cb->n_words_pad / BM, BM, fb_lay, flag_count, items, idx_out, dist_out, item_out, q_items);
        else hipLaunchKernelGGL(k_knn_fallback<KNN_MAX_K>, dim3(1024), dim3(256), 0, ctx->stream, cb->words, cb->dim, cb->dim_pad, cb->n_words, qq, ldq, metric, k,
                                fb_tps, cb->n_words_pad / BM, BM, fb_lay, flag_count, items, idx_out, dist_out, item_out, q_items);
        ISM_CHECK_LAUNCH(ctx, "k_knn_fallback");
        if (k <= 4) hipLaunchKernelGGL(k_knn_fallback_merge<4>, dim3(256), dim3(256), 0, ctx->stream, k, flag_count, qrec, item_out, q_items, idx_out, dist_out);
        else hipLaunchKernelGGL(k_knn_fallback_merge<KNN_MAX_K>, dim3(256), dim3(256), 0, ctx->stream, k, flag_count, qrec, item_out, q_items, idx_out, dist_out);
        ISM_CHECK_LAUNCH(ctx, "k_knn_fallback_merge");
    }
    if (ctx->timers_on) ISM_HIP(ctx, hipMemcpyAsync(ctx->knn_stats, flag_count, 8, hipMemcpyDeviceToHost, ctx->stream));   // read back after a sync
    return ISMHIP_OK;
}

// ---- two-stage search -------------------------------------------------------------------------------------------------------
// The top-T bookkeeping in the candidate kernel's epilogue costs time in proportion to T (measured, 262144 queries x 102400
// words: T = 4 18.7 ms, T = 2 16.8 ms, T = 1 16.2 ms), but a small T leaves more queries unproven (T = 2: ~0.1 % of them, T = 1:
// ~3 %), and the exact scan that finishes an unproven query reads its slots' share of the f32 codebook (24 us per query).
// So: stage 1 runs T = 2 over all queries; the few queries its proof rejects are gathered and searched again with T = 4 (a
// launch ~1000x smaller), and only what THAT proof rejects goes to the exact scan. Every answer is still the exact functor
// minimum, proven or scanned.
__global__ __launch_bounds__(256) void k_knn_gather_flagged(const uint32_t* __restrict__ qrec, int n2, const float* __restrict__ q, int dim,
                                                            float* __restrict__ q2, uint32_t* __restrict__ list2) {
    const int i = blockIdx.x;
    if (i >= n2) return;
    const uint32_t qi = qrec[3 * (size_t)i];
    if (threadIdx.x == 0) list2[i] = qi;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) q2[(size_t)i * dim + c] = q[(size_t)qi * dim + c];
}
__global__ void k_knn_scatter_results(const uint32_t* __restrict__ list2, int n2, int k, const int32_t* __restrict__ idx2, const float* __restrict__ dist2,
                                      int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2 * k) return;
    const size_t o = (size_t)list2[t / k] * k + (t % k);
    idx_out[o] = idx2[t]; dist_out[o] = dist2[t];
}

int run_knn_two_stage(ismhip_ctx* ctx, const ismhip_codebook* cb, int nq, const float* q, int k, int32_t* idx_out, float* dist_out) {
    KnnStage1 s1{nullptr, nullptr};
    int rc = ctx->knn_t1 == 1 && k == 1 ? run_knn<1>(ctx, cb, ISMHIP_METRIC_L2SQ, nq, q, k, idx_out, dist_out, &s1, nullptr, false, true)
                                        : run_knn<2>(ctx, cb, ISMHIP_METRIC_L2SQ, nq, q, k, idx_out, dist_out, &s1, nullptr, false, true);
    if (rc != ISMHIP_OK) return rc;
    uint32_t n2u = 0;
    ISM_HIP(ctx, hipMemcpyAsync(&n2u, s1.flag_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the one host round trip of the call: how many queries need stage 2
    ctx->knn_stage2_queries = n2u;
    const int n2 = (int)n2u;
    if (n2 == 0) { ctx->knn_stats[0] = ctx->knn_stats[1] = 0; return ISMHIP_OK; }
    float* q2 = (float*)ism_scratch(ctx, SCR_KNN_Q2, (size_t)n2 * cb->dim * sizeof(float));
    uint32_t* list2 = (uint32_t*)ism_scratch(ctx, SCR_KNN_LIST2, (size_t)n2 * (sizeof(uint32_t) + (size_t)k * (sizeof(int32_t) + sizeof(float))));
    if (!q2 || !list2) return ISMHIP_ERR_NOMEM;
    int32_t* idx2 = (int32_t*)(list2 + n2);
    float* dist2 = (float*)(idx2 + (size_t)n2 * k);
    hipLaunchKernelGGL(k_knn_gather_flagged, dim3(n2), dim3(256), 0, ctx->stream, s1.qrec, n2, q, cb->dim, q2, list2);
    ISM_CHECK_LAUNCH(ctx, "k_knn_gather_flagged");
    // Stage 2 runs 256-query tiles x 2 codebook splits on 256 CUs: 65 536 queries are two full rounds of workgroups, 66 000 are three
    // (measured 4.5 vs 6.7 ms). So a large stage 2 is cut into a multiple of 32 768 queries and a remainder, which (below 4096
    // queries) takes the merged-splits kernel that fills the chip with splits instead of query tiles.
    for (int o = 0; o < n2;) {
        const int left = n2 - o, n = left >= 32768 ? left / 32768 * 32768 : left;
        // a chunk below one full round (a shard of the split on N GPUs: 8 300 stage-2 queries per rank at N = 8) is 33 query tiles x 2
        // codebook splits = 66 workgroups on 256 CUs (2.17 ms for 8 300 queries, 2.24 for 16 600). Two candidates per lane slot instead
        // of four would allow four splits, but 1 % of these queries then fail their proof in a whole split and the exact scan costs
        // more than was won (measured: 9.18 vs 9.39 ms per step of 114 objects, 17.6 vs 15.0 of 227). The 128-query tile variant
        // (k_knn_l2_ring16<T, 1>: four lane slots per split, so four splits at T = 4) doubles the workgroups twice over instead.
        const bool was_half = ctx->knn_half;
        // (on the stage-2 image, if the codebook has one: 8 slices per tile instead of 11 on the bench data)
        const int lvl2 = cb->pca2.m > 0 && !(n < 4096) ? 2 : 0;
        if (n >= 4096 && n < 32768 && !ctx->knn_stage2_t4) ctx->knn_half = true;
        rc = run_knn<4>(ctx, cb, ISMHIP_METRIC_L2SQ, n, q2 + (size_t)o * cb->dim, k, idx2 + (size_t)o * k, dist2 + (size_t)o * k, nullptr, "knn_stage2", n < 4096, lvl2);
        ctx->knn_half = was_half;
        if (rc != ISMHIP_OK) return rc;
        o += n;
    }
    hipLaunchKernelGGL(k_knn_scatter_results, dim3((n2 * k + 255) / 256), dim3(256), 0, ctx->stream, list2, n2, k, idx2, dist2, idx_out, dist_out);
    ISM_CHECK_LAUNCH(ctx, "k_knn_scatter_results");
    return ISMHIP_OK;
}


// sqrt of every element (rows padded to dim_pad with zeros); flag[0] |= 1 when an element is negative or NaN
__global__ void k_sqrt_rows(const float* __restrict__ src, int n, int dim, int ld, int dim_pad, float* __restrict__ dst, uint32_t* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < (size_t)n * dim_pad) {
        const int row = (int)(i / dim_pad), col = (int)(i % dim_pad);
        float v = 0.f;
        if (col < dim) { v = src[(size_t)row * ld + col]; bad = !(v >= 0.f); }
        dst[i] = sqrtf(v);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// ---- chi-square, k = 1: the queries the Hellinger proof left open ------------------------------------------------------------------
// Such a query has dozens to hundreds of rows whose Hellinger distance lies below its best chi-square value (more than fixed-size
// candidate lists hold), but stage 1 has already found a very good -- usually the -- nearest row, with exact value dk. Every row
// that can still beat or tie it has LB(score) (1 - ku) <= dk, i.e. score <= tau(dk): k_hell_tau computes tau per query, the EMIT
// variant of k_knn_l2_mfma16 sweeps the sqrt images once more and appends exactly those rows to a per-query list, k_hell_fast
// evaluates a fast chi-square for every listed row (a wave each), k_hell_final walks the functor's sequential chain for the rows
// within rounding of the smallest fast value and writes the winner. Proof by construction: a row that is not listed cannot win.
#define HELL_EMIT_CAP 2048
__global__ __launch_bounds__(256) void k_hell_tau(int n2, const uint32_t* __restrict__ list2, const float* __restrict__ sq2, int dim, int dim_pad,
                                                  const float* __restrict__ dist_out, VerifyParams vp, float* __restrict__ tau) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n2) return;
    const int lane = lane_id();
    float qn2 = 0.f;
    for (int c = lane; c < dim; c += 64) { const float v = sq2[(size_t)i * dim_pad + c]; qn2 += v * v; }
    qn2 = wave_sum_f(qn2);
    if (lane == 0) {
        const float dk = dist_out[list2[i]];                             // k = 1: the best exact value of stage 1 (NaN: no candidate at all)
        const float eps_s = (17.f * KNN_U * vp.cmax2 + (2.f * vp.dot_rel + 2.f * KNN_U) * sqrtf(qn2 * vp.cmax2) + 2.f * knn_abs_err(vp, qn2) + vp.cn_acc * vp.cmax2) * 1.00001f;
        // not emitted  <=>  score > tau  =>  LB(score) (1 - ku) > dk  (LB as in k_knn_rerank_hell, |score| <= |sqrt q|^2 + |sqrt c|max^2)
        float t = dk * (1.f + 2.f * vp.ku) - qn2 * (1.f - 16.f * KNN_U) + eps_s + 8.f * KNN_U * (qn2 + vp.cmax2 + dk);
        if (!(dk == dk)) t = __builtin_inff();                           // nothing found so far: everything is a candidate (the cap will tell)
        tau[i] = t;
    }
}
// one workgroup per query: its four waves evaluate the fast chi-square of the listed rows, wave 0 then walks the sequential chain
// for the rows within rounding of the smallest fast value (together with stage 1's answer) and writes the winner
__global__ __launch_bounds__(256) void k_hell_eval(int n2, const uint32_t* __restrict__ list2, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ rows, int cap,
                                                   const uint32_t* __restrict__ perm, const float* __restrict__ q, int dim, const float* __restrict__ words, int dim_pad, float ku,
                                                   int32_t* __restrict__ idx_out, float* __restrict__ dist_out, uint32_t* __restrict__ overflow) {
    __shared__ float s_fast[HELL_EMIT_CAP];
    __shared__ __attribute__((aligned(16))) float s_terms[1344];
    const int i = blockIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t qi = list2[i];
    const uint32_t n = cnt[i];
    if (n > (uint32_t)cap) { if (threadIdx.x == 0) atomicAdd(overflow, 1u); return; }
    const float* qp = q + (size_t)qi * dim;
    {
        const int n4 = dim >> 2;                                        // the caller's rows are dim floats apart: whole chunks only when dim % 4 == 0 (checked on the host)
        f32x4 qv[CHI_FAST_CH];
        chi2_fast_load_q(qp, n4, lane, qv);
        for (uint32_t s_ = wv; s_ < n; s_ += 4) {
            const float part = chi2_fast(qv, words + (size_t)perm[rows[(size_t)i * cap + s_]] * dim_pad, n4, lane);
            if (lane == 0) s_fast[s_] = part;
        }
    }
    __syncthreads();
    if (wv != 0) return;
    float mn = __builtin_inff();
    for (uint32_t s_ = lane; s_ < n; s_ += 64) mn = fminf(mn, s_fast[s_]);
    mn = wave_min_f(mn);
    const float win = mn * (1.f + 12.f * ku);                            // fast and chain values differ by <= 3 ku each way (see k_knn_rerank_hell)
    unsigned long long best = ~0ull;
    {
        const int id0 = idx_out[qi];                                     // stage 1's answer stays in the race
        if (id0 >= 0) best = ((unsigned long long)__float_as_uint(dist_out[qi]) << 32) | (unsigned)id0;
    }
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        const uint32_t s_ = s0 + lane;
        const bool in = s_ < n && !(s_fast[s_] > win);
        unsigned long long pend = __ballot(in);
        const int row = in ? (int)perm[rows[(size_t)i * cap + s_]] : -1;
        while (pend) {
            const int src = __ffsll((long long)pend) - 1; pend &= pend - 1;
            const int cid = __shfl(row, src, 64);
            const float d = wave_functor(ISMHIP_METRIC_CHI2, qp, words + (size_t)cid * dim_pad, dim, lane, s_terms);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)cid;
            best = key < best ? key : best;
        }
    }
    if (lane == 0 && best != ~0ull) { idx_out[qi] = (int)(best & 0xffffffffull); dist_out[qi] = __uint_as_float((unsigned)(best >> 32)); }
}

// chi-square, two stages: Hellinger candidates on the matrix cores + exact chi-square re-rank and proof (k_knn_rerank_hell); the
// queries that stage cannot prove are gathered and go through the VALU chi-square kernel (k_knn_chi2) with its own proof and
// exact scan. One 8-byte read-back (negative flag of the batch; later the number of unproven queries) synchronises the call.
int run_knn_chi2_hellinger(ismhip_ctx* ctx, const ismhip_codebook* cb, int nq, const float* q, int k, int32_t* idx_out, float* dist_out, bool& taken) {
    taken = false;
    float* sq = (float*)ism_scratch(ctx, SCR_KNN_QSQRT, (size_t)nq * cb->dim_pad * sizeof(float) + 16);
    if (!sq) return ISMHIP_ERR_NOMEM;
    uint32_t* flag = (uint32_t*)(sq + (size_t)nq * cb->dim_pad);
    ISM_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    const size_t tot = (size_t)nq * cb->dim_pad;
    hipLaunchKernelGGL(k_sqrt_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, q, nq, cb->dim, cb->dim, cb->dim_pad, sq, flag);
    ISM_CHECK_LAUNCH(ctx, "k_sqrt_rows");
    uint32_t neg = 0;
    ISM_HIP(ctx, hipMemcpyAsync(&neg, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (neg) return ISMHIP_OK;                                          // not histogram data: the caller takes the VALU kernel
    taken = true;
    KnnStage1 s1{nullptr, nullptr};
    int rc = run_knn<4>(ctx, cb, ISMHIP_METRIC_CHI2, nq, q, k, idx_out, dist_out, &s1, nullptr, false, false, sq);
    if (rc != ISMHIP_OK) return rc;
    uint32_t n2u = 0;
    ISM_HIP(ctx, hipMemcpyAsync(&n2u, s1.flag_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->knn_stage2_queries = n2u;
    const int n2 = (int)n2u;
    if (n2 == 0) { ctx->knn_stats[0] = ctx->knn_stats[1] = 0; return ISMHIP_OK; }
    float* q2 = (float*)ism_scratch(ctx, SCR_KNN_Q2, (size_t)n2 * cb->dim * sizeof(float));
    uint32_t* list2 = (uint32_t*)ism_scratch(ctx, SCR_KNN_LIST2, (size_t)n2 * (sizeof(uint32_t) + (size_t)k * (sizeof(int32_t) + sizeof(float))));
    if (!q2 || !list2) return ISMHIP_ERR_NOMEM;
    int32_t* idx2 = (int32_t*)(list2 + n2);
    float* dist2 = (float*)(idx2 + (size_t)n2 * k);
    hipLaunchKernelGGL(k_knn_gather_flagged, dim3(n2), dim3(256), 0, ctx->stream, s1.qrec, n2, q, cb->dim, q2, list2);
    ISM_CHECK_LAUNCH(ctx, "k_knn_gather_flagged");
    if (k == 1 && ctx->knn_hell_emit) {
        // k = 1: list every row that can still beat stage 1's answer and evaluate those (see k_hell_tau)
        uint32_t over = 0;
        {
        TimerScope t2(ctx, "knn_stage2");
        const ismhip_codebook* xb = cb->chi_shadow;
        const int dp = cb->dim_pad, cap = HELL_EMIT_CAP;
        const int n2p = (n2 + 127) / 128 * 128;
        const size_t b_sq2 = (size_t)n2 * dp * 4, b_tau = (size_t)n2p * 4, b_cnt = (size_t)n2p * 4 + 64, b_rows = (size_t)n2 * cap * 4, b_img = (size_t)n2p * cb->ld16 * 2;
        char* buf = (char*)ism_scratch(ctx, SCR_KNN_HELL_EMIT, b_sq2 + b_tau + b_cnt + b_rows + b_img + 64);
        if (!buf) return ISMHIP_ERR_NOMEM;
        float* sq2 = (float*)buf; float* tau = (float*)(buf + b_sq2); uint32_t* cnt = (uint32_t*)(buf + b_sq2 + b_tau);
        uint32_t* sc = cnt + n2p;                                       // [0..2] f16 scalars of the gathered batch, [8] overflow counter
        uint32_t* rows = (uint32_t*)(buf + b_sq2 + b_tau + b_cnt); u16* qimg = (u16*)(buf + b_sq2 + b_tau + b_cnt + b_rows);
        ISM_HIP(ctx, hipMemsetAsync(cnt, 0, b_cnt, ctx->stream));
        hipLaunchKernelGGL(k_knn_gather_flagged, dim3(n2), dim3(256), 0, ctx->stream, s1.qrec, n2, (const float*)sq, dp, sq2, list2);
        ISM_CHECK_LAUNCH(ctx, "k_knn_gather_flagged");
        hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, ctx->stream, (const float*)sq2, n2, cb->dim, dp, sc);
        ISM_CHECK_LAUNCH(ctx, "k_absmax");
        const size_t tot = (size_t)n2p * cb->ld16;
        hipLaunchKernelGGL(k_to_f16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)sq2, n2, cb->dim, dp, n2p, cb->ld16, sc, xb->f16_scale, qimg);
        ISM_CHECK_LAUNCH(ctx, "k_to_f16");
        VerifyParams vp;
        vp.ku = 1.01f * (float)dp * KNN_U;
        vp.dot_rel = 2.002f * 4.8828125e-04f + 1.01f * (float)dp * 1.1920929e-07f;
        vp.cmax2 = xb->max_norm2; vp.dabs_c = 6.103515625e-05f / xb->f16_scale; vp.dabs_q = (const float*)(sc + 2);
        vp.sqrt_dim = sqrtf((float)dp); vp.cn_acc = 0.f;
        hipLaunchKernelGGL(k_hell_tau, dim3((n2 + 3) / 4), dim3(256), 0, ctx->stream, n2, (const uint32_t*)list2, (const float*)sq2, cb->dim, dp, (const float*)dist_out, vp, tau);
        ISM_CHECK_LAUNCH(ctx, "k_hell_tau");
        {
            const void* kern = (const void*)k_knn_l2_mfma16<4, 2, 2, 2, 2, 1, 64, 1>;
            const size_t lds = (size_t)2 * (128 + 128) * 64 * sizeof(u16) + 128 * sizeof(float);
            if (!ctx->attr_done.count(kern)) { ISM_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ctx->attr_done.insert(kern); }
            const int n_qt = n2p / 128, n_mt = cb->n_words_pad / 128;
            int nsp = std::max(1, std::min(n_mt / 2, (2048 + 8 * ((n_qt + 7) / 8) - 1) / (8 * ((n_qt + 7) / 8))));
            int tps = (n_mt + nsp - 1) / nsp; nsp = (n_mt + tps - 1) / tps;
            const u16* wh = xb->words_f16; const u16* wl = nullptr; const float* word_norm = xb->word_norm; const float* osc = (const float*)(sc + 1);
            int n_tiles_m = n_mt, ld16 = cb->ld16, k_steps = (cb->dim + 15) / 16, nq_ = n2, ncand = cb->n_words, nb = 0, cap_ = cap;
            const u16* qh_ = qimg; const u16* ql_ = nullptr; float* nf = nullptr; int* ni = nullptr; const float* tau_ = tau;
            void* args[] = {&wh, &wl, &word_norm, &n_tiles_m, &ld16, &k_steps, &qh_, &ql_, &nq_, &osc, &tps, &nsp, &nf, &ni, &ncand, &nf, &nb, &tau_, &cnt, &rows, &cap_};
            ISM_HIP(ctx, hipLaunchKernel(kern, dim3(8 * ((n_qt + 7) / 8) * nsp), dim3(256), args, lds, ctx->stream));
            ISM_CHECK_LAUNCH(ctx, "k_knn_l2_mfma16<emit>");
        }
        hipLaunchKernelGGL(k_hell_eval, dim3(n2), dim3(256), 0, ctx->stream, n2, (const uint32_t*)list2, (const uint32_t*)cnt, (const uint32_t*)rows, cap,
                           (const uint32_t*)xb->shadow_perm, q, cb->dim, (const float*)cb->words, dp, vp.ku, idx_out, dist_out, sc + 8);
        ISM_CHECK_LAUNCH(ctx, "k_hell_eval");
        ISM_HIP(ctx, hipMemcpyAsync(&over, sc + 8, 4, hipMemcpyDeviceToHost, ctx->stream));
        ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        ctx->knn_stats[0] = over; ctx->knn_stats[1] = 0;
        if (over == 0) return ISMHIP_OK;                                // every list fitted: done
        // some query has more than HELL_EMIT_CAP rows below its best value: the VALU kernel answers for all gathered queries
    }
    {
        TimerScope t2(ctx, "knn_stage2");
        rc = k > 2 ? run_knn<4>(ctx, cb, ISMHIP_METRIC_CHI2, n2, q2, k, idx2, dist2, nullptr, "knn_chi2_valu")
                   : run_knn<2>(ctx, cb, ISMHIP_METRIC_CHI2, n2, q2, k, idx2, dist2, nullptr, "knn_chi2_valu");
        if (rc != ISMHIP_OK) return rc;
    }
    hipLaunchKernelGGL(k_knn_scatter_results, dim3((n2 * k + 255) / 256), dim3(256), 0, ctx->stream, list2, n2, k, idx2, dist2, idx_out, dist_out);
    ISM_CHECK_LAUNCH(ctx, "k_knn_scatter_results");
    return ISMHIP_OK;
}

}  // namespace

// bf16 hi/lo and scaled-f16 images of the codebook for k_knn_l2_mfma16 (called once from ismhip_codebook_create)
int ism_codebook_split_bf16(ismhip_ctx* ctx, ismhip_codebook* cb, uint32_t absmax_bits) {
    cb->ld16 = (cb->dim + 63) / 64 * 64;
    const size_t tot = (size_t)cb->n_words_pad * cb->ld16;
    if (hipMalloc((void**)&cb->words_bf16_hi, tot * 3 * sizeof(u16) + 16) != hipSuccess) return ism_set_err(ctx, ISMHIP_ERR_NOMEM, "codebook bf16/f16 images");
    cb->words_bf16_lo = cb->words_bf16_hi + tot;
    cb->words_f16 = cb->words_bf16_lo + tot;
    uint32_t* sc = (uint32_t*)(cb->words_f16 + tot);
    hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, cb->words, cb->n_words_pad, cb->dim_pad, cb->dim_pad,
                       cb->n_words_pad, cb->ld16, cb->words_bf16_hi, cb->words_bf16_lo);
    ISM_CHECK_LAUNCH(ctx, "k_split_bf16");
    cb->f16_scale = f16_scale_for(absmax_bits);
    ISM_HIP(ctx, hipMemcpyAsync(sc, &absmax_bits, 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_to_f16, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, cb->words, cb->n_words_pad, cb->dim_pad, cb->dim_pad,
                       cb->n_words_pad, cb->ld16, sc, 1.0f, cb->words_f16);
    ISM_CHECK_LAUNCH(ctx, "k_to_f16");
    {
        const int nk = ((cb->dim + 15) / 16 + 1) / 2, n_tiles = cb->n_words_pad / 256;
        const size_t tt = (size_t)n_tiles * nk * 8192;
        if (hipMalloc((void**)&cb->words_f16t, tt * sizeof(u16)) != hipSuccess) return ism_set_err(ctx, ISMHIP_ERR_NOMEM, "codebook tiled f16 image");
        hipLaunchKernelGGL(k_to_f16_tiled, dim3((unsigned)((tt + 255) / 256)), dim3(256), 0, ctx->stream, cb->words, cb->n_words_pad, cb->dim_pad, cb->dim_pad,
                           n_tiles, nk, sc, 1.0f, cb->words_f16t);
        ISM_CHECK_LAUNCH(ctx, "k_to_f16_tiled");
    }
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));       // absmax_bits is a stack variable of the caller's frame
    return ISMHIP_OK;
}

#ifdef ISM_KNN_DBG_VARIANTS
extern "C" int ismhip_debug_knn_counters(unsigned long long* out256, int reset) {
    if (out256 && hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_knn_dbg), 2048) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[256]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_dbg), z, 2048) != hipSuccess) return -1; }
    return 0;
}
#endif
extern "C" {

int ismhip_knn(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q, int k,
               int32_t* idx_out, float* dist_out) {
    if (!ctx || !cb || !q || !idx_out || !dist_out || nq < 0 || k <= 0 || (metric != ISMHIP_METRIC_L2SQ && metric != ISMHIP_METRIC_CHI2))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "knn: bad argument");
    // k <= 4 is what the candidate slots (T <= 4 per slot) are sized for. Up to KNN_MAX_K the same kernels serve: the re-rank takes
    // the k best of the <= 64 candidates, the proof asks every slot's dropped-score bound to clear the k-th exact distance -- which
    // fails whenever one slot held more than T of the true k best -- and the exact scan of those slots finishes the query.
    if (k > KNN_MAX_K) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "knn: k > 16 not built");
    if (nq == 0) return ISMHIP_OK;
    TimerScope ts(ctx, "knn");
    // T = candidates kept per slot. The bf16x3 candidate scores carry a larger error bound, so more are kept (T = 4): the proof
    // then compares against the 5th best of every slot and almost never fails.
    // (short descriptors take the exact-f32 contraction, whose scores are as good as exact: two candidates per slot are plenty)
    const bool wide = k > 2 || (metric == ISMHIP_METRIC_L2SQ && cb->words_bf16_hi && ctx->knn_mode <= 1 && !(cb->dim <= 64 && ctx->knn_mode == 0));
    // default for big squared-L2 launches with k <= 2 (every shipped configuration): the two-stage search (see run_knn_two_stage)
    if (metric == ISMHIP_METRIC_L2SQ && k <= 2 && ctx->knn_t == 0 && ctx->knn_mode == 0 && ctx->knn_two_stage && cb->words_f16t && nq >= 4096 &&
        cb->n_words_pad >= 4096 && !ctx->knn_small_tile && !ctx->knn_no_ring && (cb->dim > 64 || cb->pca.m > 0))
        return run_knn_two_stage(ctx, cb, nq, q, k, idx_out, dist_out);
    // chi-square on histogram data: Hellinger candidates on the matrix cores (run_knn_chi2_hellinger); a batch with a negative element
    // keeps the VALU kernel
    if (metric == ISMHIP_METRIC_CHI2 && cb->chi_shadow && ctx->knn_hellinger && ctx->knn_mode == 0 && ctx->knn_t == 0 && nq >= 256 && cb->n_words >= 1024 && cb->dim % 4 == 0) {
        bool taken = false;
        const int rc = run_knn_chi2_hellinger(ctx, cb, nq, q, k, idx_out, dist_out, taken);
        if (rc != ISMHIP_OK || taken) return rc;
    }
    if (ctx->knn_t == 1 && k <= 1) return run_knn<1>(ctx, cb, metric, nq, q, k, idx_out, dist_out);
    if (ctx->knn_t == 3 && k <= 3) return run_knn<3>(ctx, cb, metric, nq, q, k, idx_out, dist_out);
    if (ctx->knn_t == 2 && k <= 2) return run_knn<2>(ctx, cb, metric, nq, q, k, idx_out, dist_out);
    return wide ? run_knn<4>(ctx, cb, metric, nq, q, k, idx_out, dist_out) : run_knn<2>(ctx, cb, metric, nq, q, k, idx_out, dist_out);
}

int ismhip_knn_ratio(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q,
                     float ratio_threshold, int32_t* idx_out, float* dist_out) {
    if (!ctx || !cb || !q || !idx_out || !dist_out || nq < 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "knn_ratio: bad argument");
    if (nq == 0) return ISMHIP_OK;
    int32_t* idx2 = (int32_t*)ism_scratch(ctx, SCR_QNORM, (size_t)nq * 2 * (sizeof(int32_t) + sizeof(float)));
    if (!idx2) return ISMHIP_ERR_NOMEM;
    float* d2 = (float*)(idx2 + (size_t)nq * 2);
    int rc = ismhip_knn(ctx, cb, metric, nq, q, 2, idx2, d2);
    if (rc != ISMHIP_OK) return rc;
    hipLaunchKernelGGL(k_ratio, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, nq, ratio_threshold, idx2, d2, idx_out, dist_out);
    ISM_CHECK_LAUNCH(ctx, "k_ratio");
    return ISMHIP_OK;
}

int ismhip_knn_rule(ismhip_ctx* ctx, const ismhip_codebook* cb, int metric, int nq, const float* q,
                    float ratio_threshold, int32_t* idx_out, float* dist_out) {
    if (!ctx || !cb || !q || !idx_out || !dist_out || nq < 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "knn_rule: bad argument");
    if (nq == 0) return ISMHIP_OK;
    int32_t* idx3 = (int32_t*)ism_scratch(ctx, SCR_QNORM, (size_t)nq * 3 * (sizeof(int32_t) + sizeof(float)));
    if (!idx3) return ISMHIP_ERR_NOMEM;
    float* d3 = (float*)(idx3 + (size_t)nq * 3);
    int rc = ismhip_knn(ctx, cb, metric, nq, q, 3, idx3, d3);
    if (rc != ISMHIP_OK) return rc;
    hipLaunchKernelGGL(k_rule, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, nq, ratio_threshold, idx3, d3, cb->word_class, idx_out, dist_out);
    ISM_CHECK_LAUNCH(ctx, "k_rule");
    return ISMHIP_OK;
}

}  // extern "C"
