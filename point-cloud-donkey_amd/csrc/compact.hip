// compact.hip — stable stream compaction of feature rows.
// Reference: Features::operator() drops keypoints with a non-finite LRF before description (features/features.cpp:66-76)
// and ImplicitShapeModel::removeNaNFeatures drops features whose descriptor holds a NaN (implicit_shape_model.cpp:1276-1308).
// Both filters are order preserving; here they are applied together after description (a NaN LRF yields a NaN descriptor,
// shot.hip), which leaves the same rows in the same order.
#include "common.h"

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {

// keep[k] = 1 when LRF (if given) is finite in its first component of every axis and no descriptor element is NaN
__global__ __launch_bounds__(256) void k_keep(int nkp, int dim, const float* __restrict__ desc, const float* __restrict__ lrf,
                                              uint32_t* __restrict__ keep) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nkp) return;
    const int lane = lane_id();
    bool bad = false;
    for (int i = lane; i < dim; i += 64) bad |= isnan(desc[(size_t)k * dim + i]);
    if (lrf && lane < 3) bad |= !isfinite(lrf[(size_t)k * 9 + lane * 3]);
    const unsigned long long m = __ballot(bad);
    if (lane == 0) keep[k] = m == 0ull ? 1u : 0u;
}

// the same decision from ONE element per row: rows that are NaN as a whole (what shot.hip / fpfh.hip write for an invalid frame, an
// empty neighbourhood or a zero norm) -- the 1.3 GB descriptor matrix of a 908-object batch is not read again just to be tested
__global__ void k_keep_rows(int nkp, int dim, const float* __restrict__ desc, const float* __restrict__ lrf, uint32_t* __restrict__ keep) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkp) return;
    bool bad = isnan(desc[(size_t)k * dim]);
    if (lrf) bad |= !isfinite(lrf[(size_t)k * 9]) || !isfinite(lrf[(size_t)k * 9 + 3]) || !isfinite(lrf[(size_t)k * 9 + 6]);
    keep[k] = bad ? 0u : 1u;
}
__global__ void k_iota(uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// one block per object: exclusive scan of keep[] over the object's rows -> dst position inside the object, count per object
__global__ __launch_bounds__(256) void k_scan_obj(const uint32_t* __restrict__ kp_off, const uint32_t* __restrict__ keep,
                                                  uint32_t* __restrict__ pos, uint32_t* __restrict__ obj_count) {
    const int o = blockIdx.x;
    const uint32_t b = kp_off[o], e = kp_off[o + 1];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = b; base < e; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < e ? keep[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off, 64); if (lane_id() >= off) incl += t; }
        const int w = threadIdx.x >> 6;
        if (lane_id() == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += s_w[k];
        const uint32_t carry = s_carry;
        if (i < e) pos[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) obj_count[o] = s_carry;
}

__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ kp_off, const uint32_t* __restrict__ new_off,
                                                const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, int dim,
                                                const float* __restrict__ desc, const float* __restrict__ lrf,
                                                const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                                                float* __restrict__ desc_o, float* __restrict__ lrf_o,
                                                float* __restrict__ kx_o, float* __restrict__ ky_o, float* __restrict__ kz_o,
                                                uint32_t* __restrict__ src_o) {
    const int o = blockIdx.y;
    const uint32_t k = kp_off[o] + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= kp_off[o + 1] || !keep[k]) return;
    const uint32_t d = new_off[o] + pos[k];
    const int lane = lane_id();
    for (int i = lane; i < dim; i += 64) desc_o[(size_t)d * dim + i] = desc[(size_t)k * dim + i];
    if (lrf && lrf_o && lane < 9) lrf_o[(size_t)d * 9 + lane] = lrf[(size_t)k * 9 + lane];
    if (lane == 0) { kx_o[d] = kx[k]; ky_o[d] = ky[k]; kz_o[d] = kz[k]; if (src_o) src_o[d] = k; }
}

}  // namespace

static int compact_features(ismhip_ctx* ctx, int n_obj, const uint32_t* kp_offsets_h, int dim,
                            const float* desc, const float* lrf9,
                            const float* kpx, const float* kpy, const float* kpz,
                            float* desc_out, float* lrf9_out,
                            float* kpx_out, float* kpy_out, float* kpz_out,
                            uint32_t* src_index_out, uint32_t* keep_offsets_h_out, bool whole_rows, int* all_kept_out) {
    if (all_kept_out) *all_kept_out = 0;
    if (!ctx || n_obj <= 0 || !kp_offsets_h || dim <= 0 || !desc || !kpx || !kpy || !kpz || !desc_out || !kpx_out || !kpy_out ||
        !kpz_out || !keep_offsets_h_out)
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "compact_features: bad argument");
    const uint32_t nkp = kp_offsets_h[n_obj];
    uint32_t maxk = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (kp_offsets_h[o + 1] < kp_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "compact_features: offsets not monotone");
        maxk = std::max(maxk, kp_offsets_h[o + 1] - kp_offsets_h[o]);
    }
    keep_offsets_h_out[0] = 0;
    if (nkp == 0) { for (int o = 0; o < n_obj; ++o) keep_offsets_h_out[o + 1] = 0; return ISMHIP_OK; }
    uint32_t* ko = ism_upload_offsets(ctx, SCR_KP_OFF, kp_offsets_h, n_obj + 1);
    uint32_t* keep = (uint32_t*)ism_scratch(ctx, SCR_COMPACT_KEEP, (size_t)nkp * 4);
    uint32_t* pos = (uint32_t*)ism_scratch(ctx, SCR_COMPACT_POS, (size_t)nkp * 4);
    uint32_t* cnt = (uint32_t*)ism_scratch(ctx, SCR_OBJ_COUNT, (size_t)(2 * n_obj + 2) * 4);
    if (!ko || !keep || !pos || !cnt) return ISMHIP_ERR_NOMEM;
    if (whole_rows) hipLaunchKernelGGL(k_keep_rows, dim3((nkp + 255) / 256), dim3(256), 0, ctx->stream, (int)nkp, dim, desc, lrf9, keep);
    else hipLaunchKernelGGL(k_keep, dim3((nkp + 3) / 4), dim3(256), 0, ctx->stream, (int)nkp, dim, desc, lrf9, keep);
    ISM_CHECK_LAUNCH(ctx, "k_keep");
    hipLaunchKernelGGL(k_scan_obj, dim3(n_obj), dim3(256), 0, ctx->stream, ko, keep, pos, cnt);
    ISM_CHECK_LAUNCH(ctx, "k_scan_obj");
    std::vector<uint32_t> cnt_h(n_obj);
    ISM_HIP(ctx, hipMemcpyAsync(cnt_h.data(), cnt, (size_t)n_obj * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int o = 0; o < n_obj; ++o) keep_offsets_h_out[o + 1] = keep_offsets_h_out[o] + cnt_h[o];
    if (all_kept_out && keep_offsets_h_out[n_obj] == nkp) {
        // nothing to drop: the caller keeps using its input arrays (no 2.6 GB copy of the descriptor matrix onto itself)
        *all_kept_out = 1;
        if (src_index_out) {
            hipLaunchKernelGGL(k_iota, dim3((nkp + 255) / 256), dim3(256), 0, ctx->stream, nkp, src_index_out);
            ISM_CHECK_LAUNCH(ctx, "k_iota");
        }
        return ISMHIP_OK;
    }
    uint32_t* new_off = cnt + n_obj;
    ISM_HIP(ctx, hipMemcpyAsync(new_off, keep_offsets_h_out, (size_t)(n_obj + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather, dim3((maxk + 3) / 4, n_obj), dim3(256), 0, ctx->stream, ko, new_off, keep, pos, dim, desc, lrf9,
                       kpx, kpy, kpz, desc_out, lrf9_out, kpx_out, kpy_out, kpz_out, src_index_out);
    ISM_CHECK_LAUNCH(ctx, "k_gather");
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));   // keep_offsets_h_out is read by the caller right away; new_off copy must land
    return ISMHIP_OK;
}

extern "C" int ismhip_compact_features(ismhip_ctx* ctx, int n_obj, const uint32_t* kp_offsets_h, int dim,
                                       const float* desc, const float* lrf9,
                                       const float* kpx, const float* kpy, const float* kpz,
                                       float* desc_out, float* lrf9_out,
                                       float* kpx_out, float* kpy_out, float* kpz_out,
                                       uint32_t* src_index_out, uint32_t* keep_offsets_h_out) {
    return compact_features(ctx, n_obj, kp_offsets_h, dim, desc, lrf9, kpx, kpy, kpz, desc_out, lrf9_out, kpx_out, kpy_out, kpz_out, src_index_out,
                            keep_offsets_h_out, false, nullptr);
}
extern "C" int ismhip_compact_descriptor_rows(ismhip_ctx* ctx, int n_obj, const uint32_t* kp_offsets_h, int dim,
                                              const float* desc, const float* lrf9,
                                              const float* kpx, const float* kpy, const float* kpz,
                                              float* desc_out, float* lrf9_out,
                                              float* kpx_out, float* kpy_out, float* kpz_out,
                                              uint32_t* src_index_out, uint32_t* keep_offsets_h_out, int* all_kept_out) {
    if (!all_kept_out) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "compact_descriptor_rows: all_kept_out is NULL");
    return compact_features(ctx, n_obj, kp_offsets_h, dim, desc, lrf9, kpx, kpy, kpz, desc_out, lrf9_out, kpx_out, kpy_out, kpz_out, src_index_out,
                            keep_offsets_h_out, true, all_kept_out);
}

// ---- ImplicitShapeModel::filterNormals (implicit_shape_model.cpp:1034-1075): the points whose estimated normal holds a NaN leave the
//      cloud, order kept. The arrays stay in HBM: flag, per-object scan (k_scan_obj), gather into the output arrays.
namespace {
__global__ void k_keep_normals(uint32_t n, const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = (isnan(nx[i]) || isnan(ny[i]) || isnan(nz[i])) ? 0u : 1u;
}
__global__ __launch_bounds__(256) void k_gather_points(const uint32_t* __restrict__ off, const uint32_t* __restrict__ new_off, const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ pos, ismhip_point_arrays in, ismhip_point_arrays out) {
    const int o = blockIdx.y;
    const uint32_t i = off[o] + blockIdx.x * 256 + threadIdx.x;
    if (i >= off[o + 1] || !keep[i]) return;
    const uint32_t d = new_off[o] + pos[i];
    out.x[d] = in.x[i]; out.y[d] = in.y[i]; out.z[d] = in.z[i];
    out.nx[d] = in.nx[i]; out.ny[d] = in.ny[i]; out.nz[d] = in.nz[i];
    if (in.rgba && out.rgba) out.rgba[d] = in.rgba[i];
}
}  // namespace

extern "C" int ismhip_filter_normals(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h, const ismhip_point_arrays* in,
                                     const ismhip_point_arrays* out, uint32_t* pt_offsets_h_out) {
    if (!ctx || n_obj <= 0 || !pt_offsets_h || !in || !out || !pt_offsets_h_out || !in->x || !in->y || !in->z || !in->nx || !in->ny || !in->nz ||
        !out->x || !out->y || !out->z || !out->nx || !out->ny || !out->nz || (in->rgba && !out->rgba))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "filter_normals: bad argument");
    if (in->x == out->x || in->y == out->y || in->z == out->z || in->nx == out->nx || in->ny == out->ny || in->nz == out->nz || (in->rgba && in->rgba == out->rgba))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "filter_normals: the output arrays must not alias the inputs");
    const uint32_t n = pt_offsets_h[n_obj];
    uint32_t maxn = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (pt_offsets_h[o + 1] < pt_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "filter_normals: offsets not monotone");
        maxn = std::max(maxn, pt_offsets_h[o + 1] - pt_offsets_h[o]);
    }
    pt_offsets_h_out[0] = 0;
    if (n == 0) { for (int o = 0; o < n_obj; ++o) pt_offsets_h_out[o + 1] = 0; return ISMHIP_OK; }
    uint32_t* po = ism_upload_offsets(ctx, SCR_KP_OFF, pt_offsets_h, n_obj + 1);
    uint32_t* keep = (uint32_t*)ism_scratch(ctx, SCR_COMPACT_KEEP, (size_t)n * 4);
    uint32_t* pos = (uint32_t*)ism_scratch(ctx, SCR_COMPACT_POS, (size_t)n * 4);
    uint32_t* cnt = (uint32_t*)ism_scratch(ctx, SCR_OBJ_COUNT, (size_t)(2 * n_obj + 2) * 4);
    if (!po || !keep || !pos || !cnt) return ISMHIP_ERR_NOMEM;
    hipLaunchKernelGGL(k_keep_normals, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, in->nx, in->ny, in->nz, keep);
    ISM_CHECK_LAUNCH(ctx, "k_keep_normals");
    hipLaunchKernelGGL(k_scan_obj, dim3(n_obj), dim3(256), 0, ctx->stream, po, keep, pos, cnt);
    ISM_CHECK_LAUNCH(ctx, "k_scan_obj");
    std::vector<uint32_t> cnt_h(n_obj);
    ISM_HIP(ctx, hipMemcpyAsync(cnt_h.data(), cnt, (size_t)n_obj * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int o = 0; o < n_obj; ++o) pt_offsets_h_out[o + 1] = pt_offsets_h_out[o] + cnt_h[o];
    uint32_t* new_off = cnt + n_obj;
    ISM_HIP(ctx, hipMemcpyAsync(new_off, pt_offsets_h_out, (size_t)(n_obj + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather_points, dim3((maxn + 255) / 256, n_obj), dim3(256), 0, ctx->stream, po, new_off, keep, pos, *in, *out);
    ISM_CHECK_LAUNCH(ctx, "k_gather_points");
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host copy of the offsets is pageable memory read by the H2D above
    return ISMHIP_OK;
}

// ---- partial descriptors: Codebook::castVotes with UsePartialShot (codebook/codebook.cpp:416-475) keeps the histograms of a subset
//      of the 32 SHOT signatures (getSignatureMask, :952-1036); on the device that is a column gather of the descriptor matrix.
namespace {
__global__ void k_gather_columns(size_t total, int dim_in, int n_cols, const float* __restrict__ src, const int32_t* __restrict__ cols, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / n_cols; const int c = (int)(i % n_cols);
    dst[i] = src[row * dim_in + cols[c]];
}
}  // namespace

extern "C" int ismhip_gather_columns(ismhip_ctx* ctx, int n_rows, int dim_in, const float* src, int n_cols, const int32_t* cols_h, float* dst) {
    if (!ctx || n_rows < 0 || dim_in <= 0 || n_cols <= 0 || !src || !cols_h || !dst) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "gather_columns: bad argument");
    for (int c = 0; c < n_cols; ++c) if (cols_h[c] < 0 || cols_h[c] >= dim_in) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "gather_columns: column out of range");
    if (n_rows == 0) return ISMHIP_OK;
    int32_t* cols = (int32_t*)ism_scratch(ctx, SCR_COMPACT_POS, (size_t)n_cols * 4);
    if (!cols) return ISMHIP_ERR_NOMEM;
    ISM_HIP(ctx, hipMemcpyAsync(cols, cols_h, (size_t)n_cols * 4, hipMemcpyHostToDevice, ctx->stream));
    const size_t total = (size_t)n_rows * n_cols;
    hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, total, dim_in, n_cols, src, cols, dst);
    ISM_CHECK_LAUNCH(ctx, "k_gather_columns");
    return ISMHIP_OK;
}
