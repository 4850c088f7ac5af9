// grid.hip — search surface: per-object uniform grid + cell-sorted SoA copy of the cloud.
// Replaces the pcl::search::KdTree the reference builds per object (implicit_shape_model.cpp:823-831):
// an exact fixed-radius search over a grid is equivalent up to neighbour order.
//
// HBM layout: points of all objects are concatenated SoA (x|y|z|nx|ny|nz, 4-byte floats) as the caller hands them over. The
// sorted copy orders every object's points by cell id (x fastest, original index inside a cell) and packs them as 16-byte
// records (x,y,z,index | nx,ny,nz,0 | L,a,b,0), so the 2r-wide x-run of cells a query ball touches in one (y,z) row is ONE
// contiguous span read with one coalesced global_load_dwordx4 per candidate.
#include "common.h"
#include <cfloat>

namespace {

__global__ __launch_bounds__(256) void k_bbox_meta(const uint32_t* __restrict__ pt_off,
                                                   const float* __restrict__ x, const float* __restrict__ y,
                                                   const float* __restrict__ z, float req_cell, float x_frac,
                                                   GridMeta* __restrict__ meta, uint32_t* __restrict__ cell_start) {
    const int o = blockIdx.x;
    const uint32_t b = pt_off[o], e = pt_off[o + 1];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    double s[3] = {0, 0, 0};
    uint32_t cnt = 0;
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const float px = x[i], py = y[i], pz = z[i];
        if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue;
        mn[0] = fminf(mn[0], px); mx[0] = fmaxf(mx[0], px);
        mn[1] = fminf(mn[1], py); mx[1] = fmaxf(mx[1], py);
        mn[2] = fminf(mn[2], pz); mx[2] = fmaxf(mx[2], pz);
        s[0] += px; s[1] += py; s[2] += pz; cnt++;
    }
    __shared__ float s_mn[4][3], s_mx[4][3];
    __shared__ double s_s[4][3];
    __shared__ uint32_t s_c[4];
    __shared__ int s_ncell;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
        s[a] = wave_sum_d(s[a]);
    }
    cnt = (uint32_t)wave_sum_i((int)cnt);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
        for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; s_s[w][a] = s[a]; }
        s_c[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        GridMeta m;
        uint32_t c = 0; double ss[3] = {0, 0, 0};
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int k = 0; k < 4; ++k) {
            c += s_c[k];
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], s_mn[k][a]); hi[a] = fmaxf(hi[a], s_mx[k][a]); ss[a] += s_s[k][a]; }
        }
        if (c == 0) { for (int a = 0; a < 3; ++a) { lo[a] = 0.f; hi[a] = 0.f; } }
        // requested edge = the y/z edge; x cells are ISM_GRID_XFRAC times finer. An axis that would need more than
        // ISM_GRID_MAXDIM cells gets the smallest edge that fits.
        int ncell = 1;
        for (int a = 0; a < 3; ++a) {
            float cell = req_cell > 0.f ? req_cell : 1.f;
            if (a == 0) cell *= x_frac;
            if (!((hi[a] - lo[a]) / cell < (float)(ISM_GRID_MAXDIM - 1))) cell = fmaxf(cell, (hi[a] - lo[a]) / ((float)ISM_GRID_MAXDIM - 1.5f));
            m.cell[a] = cell; m.inv_cell[a] = 1.0f / cell;
            m.minv[a] = lo[a];
            int d = (int)floorf((hi[a] - lo[a]) * m.inv_cell[a]) + 1;
            d = d < 1 ? 1 : (d > ISM_GRID_MAXDIM ? ISM_GRID_MAXDIM : d);
            m.dim[a] = d; ncell *= d;
            m.centroid[a] = c ? (float)(ss[a] / (double)c) : 0.f;   // pcl::compute3DCentroid (double accumulate)
        }
        m.n_finite = c;
        meta[o] = m;
        s_ncell = ncell;
    }
    __syncthreads();
    uint32_t* cs = cell_start + (size_t)o * ISM_GRID_STRIDE;
    for (int i = threadIdx.x; i <= s_ncell; i += blockDim.x) cs[i] = 0u;
}

// counts points per cell; remembers each point's cell and its arrival rank inside the cell
__global__ __launch_bounds__(256) void k_count(const uint32_t* __restrict__ pt_off, const float* __restrict__ x,
                                               const float* __restrict__ y, const float* __restrict__ z,
                                               const GridMeta* __restrict__ meta, uint32_t* __restrict__ cell_start,
                                               uint32_t* __restrict__ cell_of_pt, uint32_t* __restrict__ rank_of_pt) {
    const int o = blockIdx.y;
    const uint32_t b = pt_off[o], e = pt_off[o + 1];
    const uint32_t i = b + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    const GridMeta m = meta[o];
    const float px = x[i], py = y[i], pz = z[i];
    if (!(isfinite(px) && isfinite(py) && isfinite(pz))) { cell_of_pt[i] = 0xffffffffu; return; }
    const int cx = cell_coord(px, m.minv[0], m.inv_cell[0], m.dim[0]);
    const int cy = cell_coord(py, m.minv[1], m.inv_cell[1], m.dim[1]);
    const int cz = cell_coord(pz, m.minv[2], m.inv_cell[2], m.dim[2]);
    const uint32_t c = (uint32_t)((cz * m.dim[1] + cy) * m.dim[0] + cx);
    cell_of_pt[i] = c;
    rank_of_pt[i] = atomicAdd(&cell_start[(size_t)o * ISM_GRID_STRIDE + c], 1u);
}

// exclusive scan of the per-cell counts of one object (<= 32768 cells), in place; entry [ncell] = total
__global__ __launch_bounds__(1024) void k_scan(const GridMeta* __restrict__ meta, uint32_t* __restrict__ cell_start) {
    const int o = blockIdx.x;
    const GridMeta m = meta[o];
    const int ncell = m.dim[0] * m.dim[1] * m.dim[2];
    uint32_t* cs = cell_start + (size_t)o * ISM_GRID_STRIDE;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < ncell ? cs[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane_id() >= off) incl += t;
        }
        const int w = threadIdx.x >> 6;
        if (lane_id() == 63) s_wave[w] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int k = 0; k < w; ++k) wave_off += s_wave[k];
        const uint32_t carry = s_carry;
        if (i < ncell) cs[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) cs[ncell] = s_carry;
}

// members[b + cell_start[c] + arrival rank] = object-local index: the points of every cell, grouped (arrival order)
__global__ __launch_bounds__(256) void k_members(const uint32_t* __restrict__ pt_off, const uint32_t* __restrict__ cell_start,
                                                 const uint32_t* __restrict__ cell_of_pt, const uint32_t* __restrict__ rank_of_pt,
                                                 uint32_t* __restrict__ members) {
    const int o = blockIdx.y;
    const uint32_t b = pt_off[o], e = pt_off[o + 1];
    const uint32_t i = b + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    const uint32_t c = cell_of_pt[i];
    if (c == 0xffffffffu) return;
    members[b + cell_start[(size_t)o * ISM_GRID_STRIDE + c] + rank_of_pt[i]] = i - b;
}

// SingleObjectHelper::getModelRadius (single_object_mode_helper.cpp:15-27): max over the points of |p - centroid| (Eigen norm: sqrt of
// the float sum x^2 + y^2 + z^2); max is order-free, points with a non-finite coordinate never compare greater
__global__ __launch_bounds__(256) void k_cloud_radii(const uint32_t* __restrict__ pt_off, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                     const float* __restrict__ centroid, float* __restrict__ radius) {
    __shared__ float s_m[4];
    const int o = blockIdx.x;
    const float cx = centroid[o * 3], cy = centroid[o * 3 + 1], cz = centroid[o * 3 + 2];
    float m = 0.f;
    for (uint32_t i = pt_off[o] + threadIdx.x; i < pt_off[o + 1]; i += 256) {
        const float dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if (d > m) m = d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) radius[o] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// Scatter into the cell-sorted packed arrays. The position of a point inside its cell is its rank BY ORIGINAL INDEX among the
// cell's members (a stable counting sort), not the atomic arrival rank of k_count: the sorted copy, and with it the order of
// every floating-point accumulation over a neighbourhood (LRF covariance, FPFH sums), is the same from run to run.
template <bool COLOR>
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ pt_off,
                                                 const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                 const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                                 const uint32_t* __restrict__ rgba,
                                                 const float* __restrict__ lut_srgb, const float* __restrict__ lut_sxyz,
                                                 const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ cell_of_pt,
                                                 const uint32_t* __restrict__ members,
                                                 float4* __restrict__ sp4, float4* __restrict__ sn4, float4* __restrict__ slab4) {
    const int o = blockIdx.y;
    const uint32_t b = pt_off[o], e = pt_off[o + 1];
    const uint32_t i = b + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e) return;
    const uint32_t c = cell_of_pt[i];
    if (c == 0xffffffffu) return;
    const uint32_t* cs = cell_start + (size_t)o * ISM_GRID_STRIDE;
    const uint32_t m0 = cs[c], m1 = cs[c + 1], me = i - b;
    uint32_t rank = 0;
    for (uint32_t t = m0; t < m1; ++t) rank += members[b + t] < me;
    const uint32_t d = b + m0 + rank;
    sp4[d] = make_float4(x[i], y[i], z[i], __uint_as_float(me));
    sn4[d] = make_float4(nx[i], ny[i], nz[i], 0.f);
    if (COLOR) {
        // RGB2CIELAB, reference: features/features_short_cshot.cpp:651-687 (PCL cshot.hpp); normalised L/100, a/120, b/120
        const uint32_t c4 = rgba[i];
        const float fr = lut_srgb[(c4 >> 16) & 0xff], fg = lut_srgb[(c4 >> 8) & 0xff], fb = lut_srgb[c4 & 0xff];
        const float X = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
        const float Y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
        const float Z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
        float vx = X / 0.95047f, vy = Y, vz = Z / 1.08883f;
        int ix = (int)(vx * 4000), iy = (int)(vy * 4000), iz = (int)(vz * 4000);
        ix = ix < 0 ? 0 : (ix > 3999 ? 3999 : ix); iy = iy < 0 ? 0 : (iy > 3999 ? 3999 : iy); iz = iz < 0 ? 0 : (iz > 3999 ? 3999 : iz);
        vx = lut_sxyz[ix]; vy = lut_sxyz[iy]; vz = lut_sxyz[iz];
        float L = 116.0f * vy - 16.0f; if (L > 100) L = 100.0f;
        float A = 500.0f * (vx - vy); if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
        float B = 200.0f * (vy - vz); if (B > 120) B = 120.0f; else if (B < -120) B = -120.0f;
        slab4[d] = make_float4(L / 100.0f, A / 120.0f, B / 120.0f, 0.f);
    }
}

__global__ void k_centroids(const GridMeta* __restrict__ meta, int n_obj, float* __restrict__ out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_obj) return;
    out[o * 3 + 0] = meta[o].centroid[0]; out[o * 3 + 1] = meta[o].centroid[1]; out[o * 3 + 2] = meta[o].centroid[2];
}

__global__ __launch_bounds__(256) void k_center_dist(const GridMeta* __restrict__ meta, const uint32_t* __restrict__ kp_off,
                                                     const float* __restrict__ kx, const float* __restrict__ ky,
                                                     const float* __restrict__ kz, float* __restrict__ out) {
    const int o = blockIdx.y;
    const uint32_t k = kp_off[o] + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= kp_off[o + 1]) return;
    const float dx = kx[k] - meta[o].centroid[0], dy = ky[k] - meta[o].centroid[1], dz = kz[k] - meta[o].centroid[2];
    out[k] = sqrtf(dx * dx + dy * dy + dz * dz);
}

}  // namespace

// uploads a small host offsets array into a scratch slot; returns device pointer (nullptr on failure)
uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n) {
    uint32_t* d = (uint32_t*)ism_scratch(ctx, slot, (size_t)n * sizeof(uint32_t));
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, off_h, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        ism_set_err(ctx, ISMHIP_ERR_HIP, "offset upload failed");
        return nullptr;
    }
    return d;
}

extern "C" {

int ismhip_cloud_create(ismhip_ctx* ctx, int n_obj, const uint32_t* pt_offsets_h,
                        const float* x, const float* y, const float* z,
                        const float* nx, const float* ny, const float* nz,
                        const uint32_t* rgba, float cell_size, ismhip_cloud** out) {
    if (!ctx || !out || n_obj <= 0 || !pt_offsets_h || !x || !y || !z || !nx || !ny || !nz || !(cell_size > 0.f))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cloud_create: bad argument");
    *out = nullptr;
    for (int o = 0; o < n_obj; ++o)
        if (pt_offsets_h[o + 1] < pt_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cloud_create: offsets not monotone");
    if (pt_offsets_h[0] != 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cloud_create: offsets must start at 0");
    ISM_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n_pts_new = pt_offsets_h[n_obj];
    const size_t np = n_pts_new ? n_pts_new : 1;
    // recycle a destroyed cloud whose allocations are large enough (steady-state batches never touch hipMalloc)
    ismhip_cloud* c = nullptr;
    for (size_t i = 0; i < ctx->cloud_pool.size(); ++i) {
        ismhip_cloud* p = ctx->cloud_pool[i];
        if (p->cap_pts >= np && p->cap_obj >= n_obj && (p->cap_color || !rgba)) { c = p; ctx->cloud_pool.erase(ctx->cloud_pool.begin() + i); break; }
    }
    const bool fresh = c == nullptr;
    if (fresh) c = new ismhip_cloud();
    c->n_obj = n_obj;
    c->pt_off_h.assign(pt_offsets_h, pt_offsets_h + n_obj + 1);
    c->n_pts = n_pts_new;
    c->max_pts = 0;
    for (int o = 0; o < n_obj; ++o) c->max_pts = std::max(c->max_pts, pt_offsets_h[o + 1] - pt_offsets_h[o]);
    c->x = x; c->y = y; c->z = z; c->nx = nx; c->ny = ny; c->nz = nz; c->rgba = rgba;
    c->requested_cell = cell_size;
    auto fail = [&](int code, const char* msg) { c->cap_pts = 0; ismhip_cloud_destroy(ctx, c); return ism_set_err(ctx, code, msg); };
    if (fresh) {
        const size_t capp = np + np / 8;
        const int n_arr = rgba ? 3 : 2;
        float4* block = nullptr;
        if (hipMalloc((void**)&block, capp * sizeof(float4) * n_arr) != hipSuccess) return fail(ISMHIP_ERR_NOMEM, "cloud_create: hipMalloc sorted arrays");
        c->sp4 = block; c->sn4 = block + capp;
        if (rgba) c->slab4 = block + 2 * capp;
        if (hipMalloc((void**)&c->members, capp * 4) != hipSuccess || hipMalloc((void**)&c->cell_of_pt, capp * 4) != hipSuccess ||
            hipMalloc((void**)&c->rank_of_pt, capp * 4) != hipSuccess || hipMalloc((void**)&c->pt_off, (size_t)(n_obj + 1) * 4) != hipSuccess ||
            hipMalloc((void**)&c->meta, (size_t)n_obj * sizeof(GridMeta)) != hipSuccess ||
            hipMalloc((void**)&c->cell_start, (size_t)n_obj * ISM_GRID_STRIDE * 4) != hipSuccess)
            return fail(ISMHIP_ERR_NOMEM, "cloud_create: hipMalloc grid");
        c->cap_pts = capp; c->cap_obj = n_obj; c->cap_color = rgba != nullptr;
    }
    if (hipMemcpyAsync(c->pt_off, c->pt_off_h.data(), (size_t)(n_obj + 1) * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(ISMHIP_ERR_HIP, "cloud_create: offsets copy");
    {
        TimerScope ts(ctx, "grid");
        // non-finite points are dropped: an object's sorted span holds its n_finite points first, the tail is never read
        hipLaunchKernelGGL(k_bbox_meta, dim3(n_obj), dim3(256), 0, ctx->stream, c->pt_off, x, y, z, cell_size, ctx->grid_xfrac > 0.f ? 1.0f / ctx->grid_xfrac : 1.0f / (float)ISM_GRID_XFRAC, c->meta, c->cell_start);
        const dim3 g((c->max_pts + 255) / 256 ? (c->max_pts + 255) / 256 : 1, n_obj);
        hipLaunchKernelGGL(k_count, g, dim3(256), 0, ctx->stream, c->pt_off, x, y, z, c->meta, c->cell_start, c->cell_of_pt, c->rank_of_pt);
        hipLaunchKernelGGL(k_scan, dim3(n_obj), dim3(1024), 0, ctx->stream, c->meta, c->cell_start);
        hipLaunchKernelGGL(k_members, g, dim3(256), 0, ctx->stream, c->pt_off, c->cell_start, c->cell_of_pt, c->rank_of_pt, c->members);
        if (rgba)
            hipLaunchKernelGGL(k_scatter<true>, g, dim3(256), 0, ctx->stream, c->pt_off, x, y, z, nx, ny, nz, rgba, ctx->lut_srgb, ctx->lut_sxyz,
                               c->cell_start, c->cell_of_pt, c->members, c->sp4, c->sn4, c->slab4);
        else
            hipLaunchKernelGGL(k_scatter<false>, g, dim3(256), 0, ctx->stream, c->pt_off, x, y, z, nx, ny, nz, rgba, ctx->lut_srgb, ctx->lut_sxyz,
                               c->cell_start, c->cell_of_pt, c->members, c->sp4, c->sn4, c->slab4);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ISMHIP_ERR_HIP, hipGetErrorString(e));
    *out = c;
    return ISMHIP_OK;
}

static void cloud_free(ismhip_cloud* c) {
    if (c->sp4) (void)hipFree(c->sp4);
    if (c->members) (void)hipFree(c->members);
    if (c->cell_of_pt) (void)hipFree(c->cell_of_pt);
    if (c->rank_of_pt) (void)hipFree(c->rank_of_pt);
    if (c->pt_off) (void)hipFree(c->pt_off);
    if (c->meta) (void)hipFree(c->meta);
    if (c->cell_start) (void)hipFree(c->cell_start);
    delete c;
}

int ismhip_cloud_destroy(ismhip_ctx* ctx, ismhip_cloud* c) {
    if (!c) return ISMHIP_ERR_INVALID;
    // stream order protects the buffers: a recycled cloud is only rewritten by later work on the same stream
    if (ctx && c->cap_pts > 0 && ctx->cloud_pool.size() < 4) { ctx->cloud_pool.push_back(c); return ISMHIP_OK; }
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    cloud_free(c);
    return ISMHIP_OK;
}

void ism_cloud_pool_release(ismhip_ctx* ctx) {
    for (ismhip_cloud* c : ctx->cloud_pool) cloud_free(c);
    ctx->cloud_pool.clear();
}

int ismhip_cloud_centroids(ismhip_ctx* ctx, const ismhip_cloud* cloud, float* centroid_out) {
    if (!ctx || !cloud || !centroid_out) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cloud_centroids: bad argument");
    hipLaunchKernelGGL(k_centroids, dim3((cloud->n_obj + 63) / 64), dim3(64), 0, ctx->stream, cloud->meta, cloud->n_obj, centroid_out);
    ISM_CHECK_LAUNCH(ctx, "k_centroids");
    return ISMHIP_OK;
}

int ismhip_cloud_radii(ismhip_ctx* ctx, const ismhip_cloud* cloud, const float* centroid, float* radius_out) {
    if (!ctx || !cloud || !centroid || !radius_out) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "cloud_radii: bad argument");
    hipLaunchKernelGGL(k_cloud_radii, dim3(cloud->n_obj), dim3(256), 0, ctx->stream, cloud->pt_off, cloud->x, cloud->y, cloud->z, centroid, radius_out);
    ISM_CHECK_LAUNCH(ctx, "k_cloud_radii");
    return ISMHIP_OK;
}

int ismhip_center_dist(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                       const float* kpx, const float* kpy, const float* kpz, float* out) {
    if (!ctx || !cloud || !kp_offsets_h || !kpx || !kpy || !kpz || !out) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "center_dist: bad argument");
    uint32_t* ko = ism_upload_offsets(ctx, SCR_KP_OFF, kp_offsets_h, cloud->n_obj + 1);
    if (!ko) return ISMHIP_ERR_HIP;
    uint32_t maxk = 0;
    for (int o = 0; o < cloud->n_obj; ++o) maxk = std::max(maxk, kp_offsets_h[o + 1] - kp_offsets_h[o]);
    if (maxk == 0) return ISMHIP_OK;
    hipLaunchKernelGGL(k_center_dist, dim3((maxk + 255) / 256, cloud->n_obj), dim3(256), 0, ctx->stream, cloud->meta, ko, kpx, kpy, kpz, out);
    ISM_CHECK_LAUNCH(ctx, "k_center_dist");
    return ISMHIP_OK;
}

}  // extern "C"
