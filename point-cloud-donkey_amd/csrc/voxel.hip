// voxel.hip — voxel-grid keypoints on the device (the step in front of the hot path).
// Reference seam: KeypointsVoxelGrid::iComputeKeypoints (keypoints/keypoints_voxel_grid.cpp:30-46) -> pcl::VoxelGrid<PointXYZRGB>
// with a cubic leaf: the keypoints of an object are the centroids (xyz and rgb averaged, downsample_all_data) of its occupied
// voxels, emitted in ascending voxel index  i0 + i1*div0 + i2*div0*div1,  i_a = floor(p_a / leaf) - floor(min_a / leaf)
// (SURVEY Appendix A.8).
//
// PCL sorts (voxel, point) pairs and sums floats along the sorted run. Here every object owns a dense voxel table
// (count + three 64-bit fixed-point sums + three integer colour sums, built with integer atomics: order independent and
// deterministic, within 2^-40 of the object's largest coordinate of the exact centroid -- the reference's own float
// accumulation error is larger); a per-object scan over the table then emits the occupied voxels in index order.
// Memory-bound and small next to the descriptors: N points read once, sum(div0*div1*div2) table entries written and read once.
#include "common.h"

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {

struct VoxMeta {
    float minv[3], maxabs;      // bbox minimum of the finite points, largest |coordinate|
    float maxv[3];
    int any;                    // object has at least one finite point
};
struct VoxObj {                 // filled by the host from VoxMeta
    int minb[3], div[3];
    unsigned long long base;    // first table entry of the object
    unsigned long long n_vox;
    float fix_scale, fix_inv;   // power of two: |coordinate| * fix_scale < 2^40
};

__device__ __forceinline__ bool fin3(float a, float b, float c) { return isfinite(a) && isfinite(b) && isfinite(c); }

// one workgroup per object: bbox of the finite points
__global__ __launch_bounds__(256) void k_vox_bbox(const uint32_t* __restrict__ pt_off, const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ z, VoxMeta* __restrict__ meta) {
    const int o = blockIdx.x;
    const uint32_t b = pt_off[o], e = pt_off[o + 1];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = b + threadIdx.x; i < e; i += 256) {
        const float px = x[i], py = y[i], pz = z[i];
        if (!fin3(px, py, pz)) continue;
        mn[0] = fminf(mn[0], px); mx[0] = fmaxf(mx[0], px);
        mn[1] = fminf(mn[1], py); mx[1] = fmaxf(mx[1], py);
        mn[2] = fminf(mn[2], pz); mx[2] = fmaxf(mx[2], pz);
    }
    __shared__ float s_mn[3][256], s_mx[3][256];
    for (int a = 0; a < 3; ++a) { s_mn[a][threadIdx.x] = mn[a]; s_mx[a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; ++a) {
                s_mn[a][threadIdx.x] = fminf(s_mn[a][threadIdx.x], s_mn[a][threadIdx.x + s]);
                s_mx[a][threadIdx.x] = fmaxf(s_mx[a][threadIdx.x], s_mx[a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        VoxMeta m;
        float ma = 0.f;
        for (int a = 0; a < 3; ++a) { m.minv[a] = s_mn[a][0]; m.maxv[a] = s_mx[a][0]; ma = fmaxf(ma, fmaxf(fabsf(s_mn[a][0]), fabsf(s_mx[a][0]))); }
        m.any = s_mn[0][0] <= s_mx[0][0];
        m.maxabs = ma;
        meta[o] = m;
    }
}

// table entry layout (SoA over all objects): cnt u32 | sx, sy, sz i64 fixed point | cr, cg, cb u32
struct VoxTable { uint32_t* cnt; unsigned long long *sx, *sy, *sz; uint32_t *cr, *cg, *cb; };

// thread per point: voxel index exactly as pcl::VoxelGrid computes it, integer accumulation
__global__ __launch_bounds__(256) void k_vox_accum(int n_obj, const uint32_t* __restrict__ pt_off, const float* __restrict__ x,
                                                   const float* __restrict__ y, const float* __restrict__ z, const uint32_t* __restrict__ rgba,
                                                   float inv_leaf, const VoxObj* __restrict__ objs, VoxTable t) {
    const int o = blockIdx.y;
    const uint32_t i = pt_off[o] + blockIdx.x * 256 + threadIdx.x;
    if (i >= pt_off[o + 1]) return;
    const float px = x[i], py = y[i], pz = z[i];
    if (!fin3(px, py, pz)) return;
    const VoxObj ob = objs[o];
    const long long i0 = (long long)(floorf(px * inv_leaf) - (float)ob.minb[0]);
    const long long i1 = (long long)(floorf(py * inv_leaf) - (float)ob.minb[1]);
    const long long i2 = (long long)(floorf(pz * inv_leaf) - (float)ob.minb[2]);
    const unsigned long long v = ob.base + (unsigned long long)(i0 + i1 * ob.div[0] + i2 * (long long)ob.div[0] * ob.div[1]);
    atomicAdd(&t.cnt[v], 1u);
    atomicAdd(&t.sx[v], (unsigned long long)(long long)rintf(px * ob.fix_scale));     // two's complement: signed sums through the unsigned add
    atomicAdd(&t.sy[v], (unsigned long long)(long long)rintf(py * ob.fix_scale));
    atomicAdd(&t.sz[v], (unsigned long long)(long long)rintf(pz * ob.fix_scale));
    if (rgba) {
        const uint32_t c = rgba[i];
        atomicAdd(&t.cr[v], (c >> 16) & 0xffu); atomicAdd(&t.cg[v], (c >> 8) & 0xffu); atomicAdd(&t.cb[v], c & 0xffu);
    }
}

// one workgroup per object: sweep its table in index order, emit the occupied voxels behind pt_off[o] (a staging area that
// cannot overflow: an object has at most as many occupied voxels as points), count them
__global__ __launch_bounds__(256) void k_vox_emit(const uint32_t* __restrict__ pt_off, const VoxObj* __restrict__ objs, VoxTable t, bool color,
                                                  float* __restrict__ kx, float* __restrict__ ky, float* __restrict__ kz, uint32_t* __restrict__ krgba,
                                                  uint32_t* __restrict__ obj_count) {
    const int o = blockIdx.x;
    const VoxObj ob = objs[o];
    const uint32_t out0 = pt_off[o];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (unsigned long long b = 0; b < ob.n_vox; b += 256) {
        const unsigned long long v = b + threadIdx.x;
        const uint32_t c = v < ob.n_vox ? t.cnt[ob.base + v] : 0u;
        const uint32_t f = c ? 1u : 0u;
        uint32_t incl = f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t pre = s_carry;
        for (int j = 0; j < w; ++j) pre += s_w[j];
        if (f) {
            const uint32_t dst = out0 + pre + incl - 1;
            const float fc = (float)c;
            // the reference divides the float sum by the float count; the fixed-point sum converts exactly to double first
            kx[dst] = (float)((double)(long long)t.sx[ob.base + v] * (double)ob.fix_inv) / fc;
            ky[dst] = (float)((double)(long long)t.sy[ob.base + v] * (double)ob.fix_inv) / fc;
            kz[dst] = (float)((double)(long long)t.sz[ob.base + v] * (double)ob.fix_inv) / fc;
            if (krgba) {
                uint32_t col = 0u;
                if (color) {   // static_cast<uint8_t>(float sum / float count): the 8-bit sums are exact in float
                    const uint32_t r = (uint32_t)((float)t.cr[ob.base + v] / fc) & 0xffu, g = (uint32_t)((float)t.cg[ob.base + v] / fc) & 0xffu,
                                   bl = (uint32_t)((float)t.cb[ob.base + v] / fc) & 0xffu;
                    col = (r << 16) | (g << 8) | bl;
                }
                krgba[dst] = col;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) obj_count[o] = s_carry;
}

// final packing: object o's staged run (at pt_off[o]) moves to kp_off[o]
__global__ __launch_bounds__(256) void k_vox_pack(const uint32_t* __restrict__ pt_off, const uint32_t* __restrict__ kp_off,
                                                  const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const uint32_t* __restrict__ srgba,
                                                  float* __restrict__ kx, float* __restrict__ ky, float* __restrict__ kz, uint32_t* __restrict__ krgba) {
    const int o = blockIdx.y;
    const uint32_t n = kp_off[o + 1] - kp_off[o];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = pt_off[o] + i, d = kp_off[o] + i;
    kx[d] = sx[s]; ky[d] = sy[s]; kz[d] = sz[s];
    if (krgba) krgba[d] = srgba[s];
}

}  // namespace

extern "C" int ismhip_voxel_keypoints(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h,
                                      const float* x, const float* y, const float* z, const uint32_t* rgba, float leaf,
                                      uint32_t capacity, float* kx, float* ky, float* kz, uint32_t* krgba, uint32_t* kp_offsets_h_out) {
    if (!ctx || n_obj <= 0 || !pt_offsets_h || !x || !y || !z || !kx || !ky || !kz || !kp_offsets_h_out || !(leaf > 0.f))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "voxel_keypoints: bad argument");
    if (pt_offsets_h[0] != 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "voxel_keypoints: offsets must start at 0");
    uint32_t maxn = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (pt_offsets_h[o + 1] < pt_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "voxel_keypoints: offsets not monotone");
        maxn = std::max(maxn, pt_offsets_h[o + 1] - pt_offsets_h[o]);
    }
    const uint32_t n_pts = pt_offsets_h[n_obj];
    for (int o = 0; o <= n_obj; ++o) kp_offsets_h_out[o] = 0;
    if (n_pts == 0) return ISMHIP_OK;
    ISM_HIP(ctx, hipSetDevice(ctx->device));
    TimerScope ts(ctx, "voxel_keypoints");
    uint32_t* po = ism_upload_offsets(ctx, SCR_KP_OFF, pt_offsets_h, n_obj + 1);
    if (!po) return ISMHIP_ERR_HIP;
    VoxMeta* meta = (VoxMeta*)ism_scratch(ctx, SCR_COUNTERS, (size_t)n_obj * (sizeof(VoxMeta) + sizeof(VoxObj) + sizeof(uint32_t)) + 64);
    if (!meta) return ISMHIP_ERR_NOMEM;
    hipLaunchKernelGGL(k_vox_bbox, dim3(n_obj), dim3(256), 0, ctx->stream, po, x, y, z, meta);
    ISM_CHECK_LAUNCH(ctx, "k_vox_bbox");
    std::vector<VoxMeta> mh(n_obj);
    ISM_HIP(ctx, hipMemcpyAsync(mh.data(), meta, (size_t)n_obj * sizeof(VoxMeta), hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // table geometry per object, exactly as pcl::VoxelGrid::applyFilter derives it (min/max bounding-box cells, divisions)
    const float inv_leaf = 1.0f / leaf;
    std::vector<VoxObj> oh(n_obj);
    unsigned long long total = 0;
    for (int o = 0; o < n_obj; ++o) {
        VoxObj ob{};
        ob.base = total;
        if (mh[o].any) {
            unsigned long long nv = 1;
            for (int a = 0; a < 3; ++a) {
                const double lo = std::floor((double)(mh[o].minv[a] * inv_leaf)), hi = std::floor((double)(mh[o].maxv[a] * inv_leaf));
                if (!(std::fabs(lo) < 2e9 && std::fabs(hi) < 2e9)) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "voxel_keypoints: leaf too small for the coordinates (PCL: index overflow)");
                ob.minb[a] = (int)lo; ob.div[a] = (int)(hi - lo) + 1;
                nv *= (unsigned long long)ob.div[a];
                if (nv > (1ull << 31)) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "voxel_keypoints: leaf too small for the object (PCL: index overflow)");
            }
            ob.n_vox = nv;
            int ex = 0; (void)std::frexp((double)mh[o].maxabs, &ex);      // maxabs < 2^ex
            const int k = 40 - ex;
            ob.fix_scale = std::ldexp(1.0f, std::max(-120, std::min(120, k)));
            ob.fix_inv = 1.0f / ob.fix_scale;
        }
        total += ob.n_vox;
        oh[o] = ob;
    }
    if (total > (1ull << 28)) return ism_set_err(ctx, ISMHIP_ERR_UNSUPPORTED, "voxel_keypoints: more than 2^28 voxels in one batch not built");
    VoxObj* objs = (VoxObj*)(meta + n_obj);
    uint32_t* obj_count = (uint32_t*)(objs + n_obj);
    ISM_HIP(ctx, hipMemcpyAsync(objs, oh.data(), (size_t)n_obj * sizeof(VoxObj), hipMemcpyHostToDevice, ctx->stream));
    const size_t entry = 4 + 3 * 8 + 3 * 4;
    const size_t tot8 = ((size_t)total + 1) & ~(size_t)1;                 // keeps the 64-bit arrays 8-byte aligned
    unsigned char* tab = (unsigned char*)ism_scratch(ctx, SCR_FPFH_SPFH, tot8 * entry + (size_t)n_pts * 16 + 64);
    if (!tab) return ISMHIP_ERR_NOMEM;
    ISM_HIP(ctx, hipMemsetAsync(tab, 0, tot8 * entry, ctx->stream));
    VoxTable t;
    t.sx = (unsigned long long*)tab; t.sy = t.sx + tot8; t.sz = t.sy + tot8;
    t.cnt = (uint32_t*)(t.sz + tot8); t.cr = t.cnt + tot8; t.cg = t.cr + tot8; t.cb = t.cg + tot8;
    float* stx = (float*)(t.cb + tot8); float* sty = stx + n_pts; float* stz = sty + n_pts; uint32_t* stc = (uint32_t*)(stz + n_pts);
    hipLaunchKernelGGL(k_vox_accum, dim3((maxn + 255) / 256, n_obj), dim3(256), 0, ctx->stream, n_obj, po, x, y, z, rgba, inv_leaf, objs, t);
    ISM_CHECK_LAUNCH(ctx, "k_vox_accum");
    hipLaunchKernelGGL(k_vox_emit, dim3(n_obj), dim3(256), 0, ctx->stream, po, objs, t, rgba != nullptr, stx, sty, stz, krgba ? stc : nullptr, obj_count);
    ISM_CHECK_LAUNCH(ctx, "k_vox_emit");
    std::vector<uint32_t> ch(n_obj);
    ISM_HIP(ctx, hipMemcpyAsync(ch.data(), obj_count, (size_t)n_obj * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t maxk = 0;
    for (int o = 0; o < n_obj; ++o) { kp_offsets_h_out[o + 1] = kp_offsets_h_out[o] + ch[o]; maxk = std::max(maxk, ch[o]); }
    if (kp_offsets_h_out[n_obj] > capacity) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "voxel_keypoints: output capacity too small (n_points always suffices)");
    if (maxk == 0) return ISMHIP_OK;
    uint32_t* ko = (uint32_t*)ism_scratch(ctx, SCR_SLOT_OFF, (size_t)(n_obj + 1) * 4);
    if (!ko) return ISMHIP_ERR_NOMEM;
    ISM_HIP(ctx, hipMemcpyAsync(ko, kp_offsets_h_out, (size_t)(n_obj + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_vox_pack, dim3((maxk + 255) / 256, n_obj), dim3(256), 0, ctx->stream, po, ko, stx, sty, stz, stc, kx, ky, kz, krgba);
    ISM_CHECK_LAUNCH(ctx, "k_vox_pack");
    ISM_HIP(ctx, hipStreamSynchronize(ctx->stream));     // kp_offsets_h_out / the staging copies above were pageable host memory
    return ISMHIP_OK;
}
