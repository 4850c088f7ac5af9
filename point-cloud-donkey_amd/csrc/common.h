// common.h — internal definitions shared by the HIP translation units of libismhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "../../include/ismhip.h"

#define ISM_WAVE 64

// ---- per-object uniform grid (device-resident) ------------------------------------------------
#define ISM_GRID_MAXDIM 32                      // cells per axis (cell edge doubles until the object fits)
#define ISM_GRID_MAXCELLS (ISM_GRID_MAXDIM * ISM_GRID_MAXDIM * ISM_GRID_MAXDIM)
#define ISM_GRID_STRIDE (ISM_GRID_MAXCELLS + 1) // cell_start entries reserved per object

// The grid is ANISOTROPIC: a ball query walks one contiguous x-run of points per (y,z) cell row, so the y/z edge sets the
// number of rows (per-row set-up, ragged row ends) while the x edge only sets how tightly the chord of the ball clips a row.
// x cells are therefore ISM_GRID_XFRAC times finer than y/z cells: fewer, longer rows AND less over-fetch at their ends.
#define ISM_GRID_XFRAC 3
struct GridMeta {
    float minv[3];
    float cell[3];      // edge actually used per axis (>= requested)
    float inv_cell[3];
    int   dim[3];
    float centroid[3];
    uint32_t n_finite;
};

struct ismhip_cloud {
    int n_obj = 0;
    uint32_t n_pts = 0;
    uint32_t max_pts = 0;                 // largest object
    std::vector<uint32_t> pt_off_h;
    uint32_t* pt_off = nullptr;           // [n_obj+1]
    // caller's arrays (borrowed, original order)
    const float *x = nullptr, *y = nullptr, *z = nullptr, *nx = nullptr, *ny = nullptr, *nz = nullptr;
    const uint32_t* rgba = nullptr;
    // cell-sorted packed copies (owned): one 16-byte record per point and array, so that a candidate costs ONE coalesced
    // global_load_dwordx4 (and a queued neighbour one 16-byte gather for its normal) instead of three dword loads
    float4* sp4 = nullptr;                // (x, y, z, bits of the object-local original index)
    float4* sn4 = nullptr;                // (nx, ny, nz, 0)
    float4* slab4 = nullptr;              // normalised CIELab (L, a, b, 0), only with rgba
    uint32_t* cell_of_pt = nullptr;       // scratch: cell id per original point
    uint32_t* rank_of_pt = nullptr;       // scratch: arrival rank inside the cell
    uint32_t* members = nullptr;          // scratch: object-local original indices grouped by cell in arrival order
    GridMeta* meta = nullptr;             // [n_obj]
    uint32_t* cell_start = nullptr;       // [n_obj * ISM_GRID_STRIDE]
    float requested_cell = 0.f;
    // capacities of the owned allocations (clouds are recycled through ismhip_ctx::cloud_pool)
    size_t cap_pts = 0; int cap_obj = 0; bool cap_color = false;
};

// A rotated, truncated f16 image of a codebook. Rows of R = the m leading eigenvectors of the codebook's second-moment matrix
// (orthonormal up to the error folded into inv_sig2), so |R (q - c)|^2 <= sigma_max(R)^2 |q - c|^2: a score over the leading m rotated
// coordinates is a LOWER bound of the functor value.
struct PcaImage {
    int m = 0;                       // leading rotated coordinates kept (multiple of 32); 0: not built
    float* R = nullptr;              // [m x dim_pad] fp32, row j = j-th basis vector (zero in the padding columns)
    unsigned short* f16t = nullptr;  // f16 image of R c (scale sc) in the ring kernel's streaming layout [tile][m/32][256][32]
    float* cn_scaled = nullptr;      // [n_words_pad + 256] |c^|^2 / out_scale of that image (-inf for padding rows): the ring kernel's C operand
    float* osc = nullptr;            // device scalars: [0] out_scale = -2 / (sq sc)
    float sq = 1.f, sc = 1.f;        // power-of-two f16 scales of the query / codebook images (the query scale is FIXED per codebook)
    float cmax2 = 0.f;               // max |c^|^2 over the real rows (c^ = image / sc)
    float inv_sig2 = 1.f;            // 1 / (upper bound of sigma_max(R)^2), rounded down
    float d_rel = 0.f;               // |x^ - R x|_2 <= d_rel |x|_2 + dq_abs (queries) / dc_abs (codewords): rotation + f16 rounding
    float dq_abs = 0.f, dc_abs = 0.f;
    float resid2 = 0.f;              // mean second moment per codeword OUTSIDE the leading m coordinates (the pre-pass relaxes its start thresholds by it)
    float energy = 0.f;              // share of the codebook's second moment in the leading m coordinates (diagnostic)
};

struct ismhip_codebook {
    int n_words = 0, dim = 0, dim_pad = 0, n_votes = 0, max_votes = 0, n_classes = 0;
    float* words = nullptr;          // [n_words_pad * dim_pad] row-major, zero padded (MFMA tile friendly)
    int n_words_pad = 0;
    float* word_norm = nullptr;      // [n_words_pad] squared L2 norm (+inf for padding rows)
    unsigned short* words_bf16_hi = nullptr;   // [n_words_pad * dim_pad] RN_bf16(word)            (one allocation holds hi then lo)
    unsigned short* words_bf16_lo = nullptr;   // [n_words_pad * dim_pad] RN_bf16(word - hi)
    unsigned short* words_f16 = nullptr;       // [n_words_pad * dim_pad] RN_f16(word * f16_scale)  (same allocation)
    unsigned short* words_f16t = nullptr;      // the f16 image in k_knn_l2_ring's streaming layout (k_to_f16_tiled), own allocation
    int ld16 = 0;                    // row stride (halves) of the 16-bit images: dim rounded up to 64, zero padded
    float f16_scale = 1.f;           // power of two: largest |element| * f16_scale in [2^13, 2^14)
    bool words_nonneg = false;       // no negative (or NaN) element: k_knn_chi2 may drop the functor's "sum > 0" test (see there)
    float max_norm2 = 0.f;           // max squared norm over the real rows (bounds the fp32 contraction error of kNN)
    float* word_weight = nullptr;    // [n_words]
    uint32_t* vote_off = nullptr;    // [n_words+1]
    float* vote_xyz = nullptr;       // [n_votes*3]
    float* vote_weight = nullptr;    // [n_votes]
    float* vote_class_weight = nullptr;
    uint32_t* vote_class = nullptr;
    uint32_t* vote_instance = nullptr;
    float* vote_bbox_quat = nullptr; // [n_votes*4]
    float* vote_bbox_size = nullptr; // [n_votes*3]
    float* class_sigma = nullptr;    // [n_classes]
    uint32_t* word_class = nullptr;  // [n_words] Codeword::getClassId
    // chi-square candidates on the matrix cores (Hellinger lower bound, k_knn_rerank_hell): a shadow codebook that owns the 16-bit
    // images, norms and scales of sqrt(words); its fp32 words are not kept. Only for codebooks without negative / NaN elements.
    ismhip_codebook* chi_shadow = nullptr;
    uint32_t* shadow_perm = nullptr; // (in the shadow) [n_words] codebook row of every shadow row
    // ---- rotated, truncated images of the squared-L2 search (pca.hip; m == 0: not built) ---------------------------------------------
    // pca: the stage-1 image (leading coordinates that hold 97 % of the second moment); pca2: a longer one (99.7 %) for stage 2, the
    // queries whose stage-1 proof failed -- both cut from the same eigenbasis
    PcaImage pca, pca2;
};

struct TimerAcc {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0;
    int64_t launches = 0;
};

struct ismhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool timers_on = false;
    std::map<std::string, TimerAcc> timers;
    std::vector<hipEvent_t> event_pool;
    // grow-only scratch slots
    struct Slot { void* p = nullptr; size_t bytes = 0; };
    std::map<int, Slot> scratch;
    // LUTs for RGB->Lab (device)
    float* lut_srgb = nullptr;   // [256]
    float* lut_sxyz = nullptr;   // [4000]
    // destroyed clouds keep their device allocations here for the next ismhip_cloud_create (no hipMalloc/hipFree per batch)
    std::vector<ismhip_cloud*> cloud_pool;
    uint32_t* truncated_d = nullptr;  // device counter: maxima dropped by the per-class / per-object caps of find_maxima / hough3d_maxima (ismhip_sync reports and clears it)
    uint32_t knn_stats[2] = {0, 0};   // last ismhip_knn: {queries, (query,slot) items} sent to the exact fallback (valid with timers on, after a sync)
    std::set<const void*> attr_done;  // kernels whose MaxDynamicSharedMemorySize attribute has been raised on THIS ctx's device
    uint32_t knn_stage2_queries = 0;  // last two-stage ismhip_knn: queries the T = 2 stage could not prove (searched again with T = 4)
    bool knn_two_stage = true;   // env ISMHIP_KNN_TWOSTAGE=0: single-stage T = 4 search (A/B runs)
    bool knn_small_tile = false; // env ISMHIP_KNN_TILE128=1: keep the bf16x3 kernel on its 128x128 tile (A/B runs)
    bool knn_join = true;        // env ISMHIP_KNN_JOIN=0: every workgroup of the ring kernel sweeps its split from the first tile (A/B runs); default: joined streams
    bool knn_qpanel2 = true;     // env ISMHIP_KNN_QPANEL2=0: stage 1 on <= 160 rotated coordinates WITHOUT the 256-query panel resident in LDS (A/B runs; default on: 27.5 -> 25.4 ms per bench launch)
    bool knn_qpanel = false;     // env ISMHIP_KNN_QPANEL=1: the ring kernel on 256 x 128 tiles with the query panel resident in LDS (A/B runs)
    bool knn_half = false;       // env ISMHIP_KNN_HALF=1: the ring kernel on 128 x 256 tiles, two workgroups per CU (A/B runs)
    bool knn_ring32 = false;     // env ISMHIP_KNN_RING32=1: the ring kernel on the 32x32x16 MFMA shape instead of 16x16x32 (A/B runs)
    int knn_splits = 0;          // env ISMHIP_KNN_SPLITS: force the number of codebook splits of the squared-L2 candidate kernels (A/B runs)
    int knn_t = 0;               // env ISMHIP_KNN_T = 2 | 3: candidates kept per slot (default 4 on the 16-bit paths); fewer = cheaper epilogue, more unproven slots
    int knn_dbg = 0;             // env ISMHIP_KNN_DBG: timing experiments on k_knn_l2_ring (1 no epilogue, 2 no MFMA, 3 no DMA); results invalid
    bool knn_no_ring = false;    // env ISMHIP_KNN_NORING=1: f16 candidates by the register-staged kernel instead of the LDS-DMA ring (A/B runs)
    bool knn_kb32 = false;       // env ISMHIP_KNN_KB32=1: f16 candidates with 32-deep LDS slices instead of 64 (A/B runs)
    bool xcd_map = true;         // env ISMHIP_XCD_MAP=0: per-object kernels on the plain object-major block order instead of the XCD-local map (A/B runs)
    float grid_xfrac = 0.f;      // env ISMHIP_GRID_XFRAC: x cells this many times finer than y/z cells (default ISM_GRID_XFRAC; A/B runs)
    int shot_var = 0;            // env ISMHIP_SHOT_VAR=2: k_shot on the contiguous candidate sweep instead of 16 interleaved segments (A/B runs; same histogram)
    bool knn_hell_emit = true;   // env ISMHIP_KNN_HELL_EMIT=0: queries the Hellinger proof leaves open go to the VALU kernel instead of the list-and-evaluate stage (A/B runs)
    bool knn_hellinger = true;   // env ISMHIP_KNN_HELLINGER=0: chi-square candidates by the VALU kernel k_knn_chi2 only (A/B runs)
    int knn_t1 = 2;              // env ISMHIP_KNN_T1 = 1 | 2: candidates kept per lane slot in stage 1 of the two-stage search (A/B runs)
    int knn_pre_step = 32;       // env ISMHIP_KNN_PRE_STEP: the sampling pre-pass sweeps every n-th codeword tile (measured with the resident query panel, kNN ms per bench launch at gamma 1.0: 8 -> 36.4, 16 -> 33.6, 32 -> 33.2; no pre-pass 34.7)
    float knn_pre_gamma = 1.0f;  // env ISMHIP_KNN_PRE_GAMMA: relaxation of the pre-pass start thresholds in units of the truncated second moment (A/B runs)
    bool knn_prepass = true;     // env ISMHIP_KNN_PREPASS=0: stage 1 starts every candidate list cold instead of from the sampled pre-pass threshold (A/B runs)
    int knn_pca_m = -1;          // env ISMHIP_KNN_PCA_M: leading rotated coordinates of the stage-1 image (0 = no rotated image, -1 = chosen from the spectrum)
    bool knn_stage2_t4 = false;       // env ISMHIP_KNN_STAGE2_T4=1: partial stage-2 chunks on the 256-query tiles of full ones (A/B runs)
    bool codebook_light = false;      // set around ismhip_codebook_create by ism_knn_only_codebook: skip the rotated image and the chi-square shadow
    int knn_pca_m2 = -1;         // env ISMHIP_KNN_PCA_M2: coordinates of the stage-2 image (0 = stage 2 on all dimensions, -1 = chosen from the spectrum)
    uint32_t knn_pca_launches = 0;    // squared-L2 searches whose stage 1 ran on the rotated image (tests / bench)
    int knn_mode = 0;            // env ISMHIP_KNN_MODE = f16 (0, default) | bf16x3 (1) | f32 (2): squared-L2 candidate kernel (A/B runs, tests)
};

enum ScratchSlot {
    SCR_KP_OFF = 1, SCR_TIE_LIST, SCR_TIE_REC, SCR_TIE_KEYS, SCR_COUNTERS, SCR_KNN_CAND_IDX, SCR_KNN_CAND_VAL,
    SCR_QNORM, SCR_FPFH_FLAG, SCR_FPFH_LIST, SCR_FPFH_SPFH, SCR_FPFH_LOOKUP, SCR_SLOT_OFF, SCR_CLASS_BW,
    SCR_COMPACT_KEEP, SCR_COMPACT_POS, SCR_OBJ_COUNT, SCR_QPAD, SCR_LRF_COV, SCR_KNN_FLAGS, SCR_KNN_QSPLIT, SCR_MAX_REC, SCR_QNORM2, SCR_KNN_Q2, SCR_KNN_LIST2, SCR_TRAIN, SCR_TRAIN2, SCR_MAX_WORK, SCR_KMEANS, SCR_KNN_CLOCK, SCR_PCA, SCR_KNN_THR0, SCR_KNN_QSQRT, SCR_KNN_HELL_EMIT
};

int  ism_set_err(ismhip_ctx* ctx, int code, const std::string& msg);
void* ism_scratch(ismhip_ctx* ctx, int slot, size_t bytes);   // nullptr on failure (error recorded)

#define ISM_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return ism_set_err((ctx), ISMHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define ISM_CHECK_LAUNCH(ctx, name)                                                                 \
    do {                                                                                            \
        hipError_t e__ = hipGetLastError();                                                         \
        if (e__ != hipSuccess)                                                                      \
            return ism_set_err((ctx), ISMHIP_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e__)); \
    } while (0)

// RAII-ish timer scope: records events around a group of launches on the ctx stream
struct TimerScope {
    ismhip_ctx* ctx; const char* name; hipEvent_t a = nullptr, b = nullptr;
    TimerScope(ismhip_ctx* c, const char* n);
    ~TimerScope();
};

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// squared distance exactly as FLANN L2_Simple<float> / the oracle: sequential float accumulate.
// The library is compiled with -ffp-contract=off, so no FMA is formed here.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float r = 0.f, d;
    d = ax - bx; r += d * d;
    d = ay - by; r += d * d;
    d = az - bz; r += d * d;
    return r;
}

// Wave-wide sums on the DPP data path. hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-crossbar round trip of ~100+
// cycles per step, 6 dependent steps per sum); the DPP sequence below is 6 register-to-register moves + 1 v_readlane.
// All 64 lanes must be active at the call site (they are: every use is wave-uniform). Steps: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_shr:4, row_shr:8, row_bcast:15, row_bcast:31 -> the total lands in lane 63 (lanes without a source add 0).
template <int CTRL>
__device__ __forceinline__ int dpp_mov0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ float wave_sum_f(float v) {
    v += __int_as_float(dpp_mov0<0xb1>(__float_as_int(v)));
    v += __int_as_float(dpp_mov0<0x4e>(__float_as_int(v)));
    v += __int_as_float(dpp_mov0<0x114>(__float_as_int(v)));
    v += __int_as_float(dpp_mov0<0x118>(__float_as_int(v)));
    v += __int_as_float(dpp_mov0<0x142>(__float_as_int(v)));
    v += __int_as_float(dpp_mov0<0x143>(__float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_mov0<0xb1>(v);
    v += dpp_mov0<0x4e>(v);
    v += dpp_mov0<0x114>(v);
    v += dpp_mov0<0x118>(v);
    v += dpp_mov0<0x142>(v);
    v += dpp_mov0<0x143>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
// the same data path for wave-wide minima (lanes without a source see the identity) and for an inclusive prefix sum
// (row_shr 1,2,4,8 scan the 16-lane rows; row_bcast:15 / :31 with row masks 0xa / 0xc carry the row totals forward)
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int ident, int v) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_min_f(float v) {        // NaN-free inputs or NaN-last semantics of fminf
    const int inf = 0x7f800000;
    v = fminf(v, __int_as_float(dpp_movi<0xb1>(inf, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_movi<0x4e>(inf, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_movi<0x114>(inf, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_movi<0x118>(inf, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_movi<0x142>(inf, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_movi<0x143>(inf, __float_as_int(v))));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
    const unsigned lo = (unsigned)dpp_movi<CTRL>(-1, (int)(unsigned)(v & 0xffffffffull)), hi = (unsigned)dpp_movi<CTRL>(-1, (int)(unsigned)(v >> 32));
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xb1>(v); v = dpp_min_u64<0x4e>(v); v = dpp_min_u64<0x114>(v);
    v = dpp_min_u64<0x118>(v); v = dpp_min_u64<0x142>(v); v = dpp_min_u64<0x143>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    int x = (int)v;
    x += dpp_mov0<0x111>(x); x += dpp_mov0<0x112>(x); x += dpp_mov0<0x114>(x); x += dpp_mov0<0x118>(x);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov0_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_mov0<CTRL>((int)(b & 0xffffffffll)), hi = dpp_mov0<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_mov0_d<0xb1>(v);
    v += dpp_mov0_d<0x4e>(v);
    v += dpp_mov0_d<0x114>(v);
    v += dpp_mov0_d<0x118>(v);
    v += dpp_mov0_d<0x142>(v);
    v += dpp_mov0_d<0x143>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// XCD-local block map for the per-object kernels. The hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (blocks b and b + 8 share one, MI355X_MICROARCH "Workgroup dispatch"), each with its own 4 MiB L2. With a (blocks, objects) 2-D
// grid the blocks of ONE object land on all eight XCDs and every L2 pulls every object's cloud from beyond (measured round 1:
// 8x the compulsory fetch). Here the grid is 1-D, 8 * nbx * ceil(n_obj / 8) blocks: the blocks with id = xcd (mod 8) walk the
// objects xcd, xcd + 8, ... one after the other, so an object's ~400 KB cloud is fetched once, into one L2. Speed only: the
// result does not depend on the placement. Returns false for the padding blocks of the last object group.
// Fewer than 8 objects (single-object detect()): plain object-major order, so that all XCDs work.
__device__ __forceinline__ bool xcd_object_block(int nbx, int n_obj, int& o, int& bx) {
    if (n_obj < 8) { o = blockIdx.x / nbx; bx = blockIdx.x % nbx; return true; }
    const int slot = blockIdx.x >> 3;
    o = (slot / nbx) * 8 + (blockIdx.x & 7);
    bx = slot % nbx;
    return o < n_obj;
}
static inline unsigned xcd_object_grid(unsigned nbx, int n_obj) { return n_obj < 8 ? nbx * (unsigned)n_obj : 8u * nbx * (unsigned)((n_obj + 7) / 8); }

// cell coordinate of a value along one axis, clamped (monotone in v)
__device__ __forceinline__ int cell_coord(float v, float minv, float inv_cell, int dim) {
    int c = (int)floorf((v - minv) * inv_cell);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

// Conservative cell range of the ball (q, r) along every axis. Returns false when the ball misses the grid.
struct CellRange { int lo[3], hi[3]; };
__device__ __forceinline__ bool ball_cells(const GridMeta& m, float qx, float qy, float qz, float r, CellRange& cr) {
    const float q[3] = {qx, qy, qz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pad = r * 1e-5f + fabsf(q[a]) * 4e-7f + 1e-30f;   // absorbs the rounding of q±r and of the binning
        const float lo = (q[a] - r - pad - m.minv[a]) * m.inv_cell[a];
        const float hi = (q[a] + r + pad - m.minv[a]) * m.inv_cell[a];
        int l = (int)floorf(lo), h = (int)floorf(hi);
        if (h < 0 || l > m.dim[a] - 1) return false;
        cr.lo[a] = l < 0 ? 0 : l;
        cr.hi[a] = h > m.dim[a] - 1 ? m.dim[a] - 1 : h;
    }
    return true;
}

// Clips the x cell range of one (gy, gz) cell row to the chord of the ball (q, r) in that row. Conservative: the padding
// absorbs the float rounding of the cell bounds and of the point binning. Returns false when the row misses the ball.
__device__ __forceinline__ bool row_cells(const GridMeta& m, const CellRange& cr, int gy, int gz,
                                          float qx, float qy, float qz, float r, int& lo, int& hi) {
    const float pad = r * 2e-5f + (fabsf(qx) + fabsf(qy) + fabsf(qz)) * 1e-6f + 1e-30f;
    const float y0 = m.minv[1] + (float)gy * m.cell[1], z0 = m.minv[2] + (float)gz * m.cell[2];
    const float dy = fmaxf(fmaxf(y0 - qy, qy - (y0 + m.cell[1])) - pad, 0.f);
    const float dz = fmaxf(fmaxf(z0 - qz, qz - (z0 + m.cell[2])) - pad, 0.f);
    const float rr = r + pad;
    const float rem = rr * rr - (dy * dy + dz * dz);
    if (!(rem > 0.f)) return false;
    const float hc = sqrtf(rem) + pad;
    int l = (int)floorf((qx - hc - m.minv[0]) * m.inv_cell[0]), h = (int)floorf((qx + hc - m.minv[0]) * m.inv_cell[0]);
    lo = l < cr.lo[0] ? cr.lo[0] : l;
    hi = h > cr.hi[0] ? cr.hi[0] : h;
    return lo <= hi;
}

// ---- flattened ball traversal ---------------------------------------------------------------------------------------------
// A ball query touches (2R/cell+1)^2 cell rows; walking them one after the other costs, per row, the chord arithmetic on all 64
// lanes, two dependent cell_start loads and a mostly half-empty wave of points. Here the lanes first take one row EACH (chord,
// start, length: one round trip for up to 64 rows), a wave prefix sum lays the non-empty rows end to end, and the wave then sweeps
// the concatenated candidate list 64 at a time with every lane busy.
// Candidate -> row without a search: every non-empty row marks its LAST candidate in a bit mask (one ds_or_b64 per row batch);
// for the 64 candidates of a block the row of lane l is (rows that ended before the block: a running popcount) + (marks below l:
// v_mbcnt on the block's 64-bit word, which is wave-uniform and read one block ahead), and ONE per-lane LDS read of the row's
// (start - offset) turns the flat index into the sorted point index. Round 1 walked a per-lane cursor through a prefix table:
// three to four dependent LDS round trips per block.
#define ISM_ROWS_CAP 4096                  // candidates per window of the mark mask (64 words); longer row batches take several windows
struct WaveRows {
    unsigned long long last[ISM_ROWS_CAP / 64];   // bit i of last[b]: candidate 64 b + i (window-relative) is the last one of its row
    int delta[64];                                // per non-empty row: first sorted index - flat offset
    uint32_t rows_before[ISM_ROWS_CAP / 64];      // NG > 1 only: rows that ended before block b (window-relative)
};
// L(i, valid) loads and returns the point record (issued one sweep AHEAD of its use, so the next 64 candidates are in flight
// while the current ones are processed); F(record, i, valid) consumes it. i = object-local sorted point index; a lane without a
// candidate gets i = 0 (a valid index: the sweep only runs when a row holds points), so L may load unconditionally -- a conditional
// load costs an exec-mask branch and a zero fill of the record per block -- and F must not use the record when valid is false.
// NG = 1: the wave sweeps the flat list 64 consecutive candidates at a time (fully coalesced: 1 KB per load instruction).
// NG = 8: the list is cut into 8 contiguous segments and every group of 8 lanes walks its own segment (128 B per group and
//         load: still whole cache lines). The 64 candidates a wave looks at together then come from 8 DISTANT parts of the ball
//         instead of one short stretch of one cell row -- for k_shot this is what matters: neighbours that are adjacent in the
//         cell-sorted order fall into the same (sector, cosine) bin, and same-address LDS atomics of one wave instruction
//         serialise (measured: SHOT got SLOWER with finer x cells, i.e. with FEWER candidates but a more coherent order).
template <int NG = 1, bool IL = false, class L, class F>
__device__ __forceinline__ void ball_for_each(const GridMeta& m, const uint32_t* __restrict__ cs, const CellRange& cr,
                                              float qx, float qy, float qz, float r, int lane, WaveRows& wr, L&& load, F&& f) {
    const int ny = cr.hi[1] - cr.lo[1] + 1, nrows = ny * (cr.hi[2] - cr.lo[2] + 1);
    for (int r0 = 0; r0 < nrows; r0 += 64) {
        const int j = r0 + lane;
        uint32_t s = 0, len = 0;
        if (j < nrows) {
            const int gz = cr.lo[2] + j / ny, gy = cr.lo[1] + j % ny;
            int xl, xh;
            if (row_cells(m, cr, gy, gz, qx, qy, qz, r, xl, xh)) {
                const int rb = (gz * m.dim[1] + gy) * m.dim[0];
                s = cs[rb + xl]; len = cs[rb + xh + 1] - s;
            }
        }
        const uint32_t end = wave_incl_scan_u32(len);                            // flat offset one past this row's last candidate
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
        if (total == 0) continue;
        const unsigned long long ne = __ballot(len != 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // the previous batch's table reads are done
        if (len != 0) wr.delta[__popcll(ne & ((1ull << lane) - 1ull))] = (int)s - (int)(end - len);
        for (uint32_t w0 = 0; w0 < total; w0 += ISM_ROWS_CAP) {                  // one window unless a row batch holds > 4096 candidates
            const uint32_t wn = min(total - w0, (uint32_t)ISM_ROWS_CAP);
            wr.last[lane] = 0ull;                                               // LDS traffic of one wave is ordered: no barrier needed
            if (len != 0 && end > w0 && end <= w0 + wn) atomicOr(&wr.last[(end - 1 - w0) >> 6], 1ull << ((end - 1 - w0) & 63));
            // The marks are read back by OTHER lanes. The hardware keeps the LDS operations of a wave in order, but the compiler
            // reasons per thread: without this (instruction-free) wavefront fence it forwards the 0 a lane has just stored to
            // that lane's own read of last[lane], as if no other lane could have written in between.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t row_base = (uint32_t)__popcll(__ballot(len != 0 && end <= w0));   // rows that ended before this window
            if (NG == 1) {
                auto mask_of = [&](uint32_t b) -> unsigned long long {             // wave-uniform word -> SGPR pair
                    const unsigned long long v = wr.last[b];
                    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull));
                };
                auto locate = [&](unsigned long long mk, uint32_t rb, uint32_t flat, bool v) -> uint32_t {
                    if (!v) return 0u;
                    const uint32_t row = rb + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    return (uint32_t)(wr.delta[row] + (int)flat);
                };
                unsigned long long mk = mask_of(0);
                bool vn = (uint32_t)lane < wn;
                uint32_t in = locate(mk, row_base, w0 + lane, vn);
                auto rn = load(in, vn);
                for (uint32_t c = 0; c < wn; c += 64) {
                    const bool v = vn; const uint32_t i = in; const auto rec = rn;
                    if (c + 64 < wn) {
                        row_base += (uint32_t)__popcll(mk);
                        mk = mask_of((c >> 6) + 1);
                        vn = c + 64 + lane < wn;
                        in = locate(mk, row_base, w0 + c + 64 + lane, vn);
                        rn = load(in, vn);
                    }
                    f(rec, i, v);
                }
            } else {
                constexpr int GL = 64 / NG;                                      // lanes per group
                {   // rows that ended before every 64-candidate block: exclusive prefix of the blocks' mark counts
                    const uint32_t pc = (uint32_t)__popcll(wr.last[lane]);
                    wr.rows_before[lane] = row_base + wave_incl_scan_u32(pc) - pc;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                const uint32_t seg = ((wn + NG - 1) / NG + GL - 1) / GL * GL;   // segment length, a multiple of the group width
                // IL: lane l walks segment l % NG (the 8 regions alternate from lane to lane) instead of segment l / GL
                const uint32_t grp = IL ? (uint32_t)(lane % NG) : (uint32_t)(lane / GL), sub = IL ? (uint32_t)(lane / NG) : (uint32_t)(lane % GL);
                const uint32_t g0 = grp * seg + sub;
                const uint32_t gend = min((grp + 1u) * seg, wn);
                auto locate = [&](uint32_t flat, bool v) -> uint32_t {          // flat: window-relative
                    if (!v) return 0u;
                    const unsigned long long mk = wr.last[flat >> 6];
                    const uint32_t row = wr.rows_before[flat >> 6] + (uint32_t)__popcll(mk & ((1ull << (flat & 63u)) - 1ull));
                    return (uint32_t)(wr.delta[row] + (int)(w0 + flat));
                };
                bool vn = g0 < gend;
                uint32_t in = locate(g0, vn);
                auto rn = load(in, vn);
                for (uint32_t c = 0; c < seg; c += GL) {
                    const bool v = vn; const uint32_t i = in; const auto rec = rn;
                    if (c + GL < seg) {
                        vn = g0 + c + GL < gend;
                        in = locate(g0 + c + GL, vn);
                        rn = load(in, vn);
                    }
                    f(rec, i, v);
                }
            }
        }
    }
}
#endif
