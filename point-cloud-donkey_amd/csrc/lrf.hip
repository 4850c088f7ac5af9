// lrf.hip — SHOT local reference frames.
// Reference seam: Features::computeSHOTReferenceFrames (features/features.cpp:238-252) ->
// pcl::SHOTLocalReferenceFrameEstimationOMP; arithmetic as third_party/pcl_shot_na_lrf/shot_na_lrf.hpp:48-178
// with upstream's z-sign rule (vij . v3 >= 0), SURVEY Appendix A.1.
//
// Roofline: HBM-bound gather, algorithmic bytes = sum_k M'_k * 12 + K * 48.
// Four small kernels instead of one fat one (the fused version needed 245 VGPRs -> 2 waves/SIMD and sat 51 % of its
// wave cycles in s_waitcnt):
//   k_lrf_cov  : wave per keypoint, streams the clipped candidate runs (coalesced SoA loads of the cell-sorted cloud),
//                FP64 weighted covariance per lane + wave reduction -> 8 doubles per keypoint
//   k_lrf_eig  : thread per keypoint, cyclic Jacobi 3x3 (eigen3.h) -> x / z axis candidates
//   k_lrf_sign : wave per keypoint, re-streams the runs, counts the sign votes, writes the frame or queues a tie
//   k_lrf_tie  : work-queue kernel for sign ties (~3 % of keypoints): the 5 median neighbours BY DISTANCE decide
//                (shot_na_lrf.hpp:139-151); rank selection by bitwise bisection over 64-bit (d^2, index) keys
#include "common.h"
#include "eigen3.h"

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {


struct TieRec {
    double v1[3], v3[3];
    uint32_t kp, obj;
    int valid;
    int tie_x, tie_z;     // 1 = undecided
};

struct CloudView {
    const uint32_t* pt_off; const GridMeta* meta; const uint32_t* cell_start;
    const float4* sp4;        // cell-sorted (x, y, z, original index bits)
    const float *x, *y, *z;   // original order (tie kernel)
    int n_obj, nbx;           // XCD-local block map (common.h)
};

__device__ __forceinline__ void write_lrf(float* out, const double v1[3], const double v3[3]) {
    const float xf[3] = {(float)v1[0], (float)v1[1], (float)v1[2]};
    const float zf[3] = {(float)v3[0], (float)v3[1], (float)v3[2]};
    out[0] = xf[0]; out[1] = xf[1]; out[2] = xf[2];
    out[3] = zf[1] * xf[2] - zf[2] * xf[1];     // y = z x x, in float (shot_na_lrf.hpp:170)
    out[4] = zf[2] * xf[0] - zf[0] * xf[2];
    out[5] = zf[0] * xf[1] - zf[1] * xf[0];
    out[6] = zf[0]; out[7] = zf[1]; out[8] = zf[2];
}

// sqrt(double(d2)) to full double precision from the float estimate + one Newton step (v_sqrt_f64 is slow)
__device__ __forceinline__ double sqrt_f32_as_f64(float d2) {
    if (d2 <= 0.f) return 0.0;
    const float s0f = sqrtf(d2);
    const double s0 = (double)s0f, x = (double)d2;
    return s0 + (x - s0 * s0) * (double)(0.5f / s0f);
}

// cov_out[k*8 + 0..6] = c00,c01,c02,c11,c12,c22,sum ; [7] = number of valid neighbours (as double), -1 = no search possible
__global__ __launch_bounds__(256) void k_lrf_cov(CloudView cv, const uint32_t* __restrict__ kp_off,
                                                 const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                                                 float radius, float r2, double* __restrict__ cov_out) {
    int o, bx;
    if (!xcd_object_block(cv.nbx, cv.n_obj, o, bx)) return;
    const uint32_t k = kp_off[o] + bx * 4 + (threadIdx.x >> 6);
    if (k >= kp_off[o + 1]) return;
    const int lane = lane_id();
    const float cx = kx[k], cy = ky[k], cz = kz[k];
    const GridMeta m = cv.meta[o];
    const uint32_t* cs = cv.cell_start + (size_t)o * ISM_GRID_STRIDE;
    const uint32_t base = cv.pt_off[o];
    CellRange cr;
    if (!(isfinite(cx) && isfinite(cy) && isfinite(cz)) || !ball_cells(m, cx, cy, cz, radius, cr)) {
        if (lane == 0) cov_out[(size_t)k * 8 + 7] = -1.0;
        return;
    }
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, sum = 0;
    int valid = 0;
    const double rd = (double)radius;
    __shared__ WaveRows s_rows[4];
    ball_for_each(m, cs, cr, cx, cy, cz, radius, lane, s_rows[threadIdx.x >> 6],
                  [&](uint32_t t, bool) { return cv.sp4[base + t]; },
                  [&](const float4& p, uint32_t, bool v) {
        if (!v) return;
        const float px = p.x, py = p.y, pz = p.z;
        const float d2 = sqdist3(px, py, pz, cx, cy, cz);
        if (d2 < r2 && !(px == cx && py == cy && pz == cz)) {
            const double vx = (double)(px - cx), vy = (double)(py - cy), vz = (double)(pz - cz);
            const double w = rd - sqrt_f32_as_f64(d2);
            const double wx = w * vx, wy = w * vy, wz = w * vz;
            c00 = fma(wx, vx, c00); c01 = fma(wx, vy, c01); c02 = fma(wx, vz, c02);
            c11 = fma(wy, vy, c11); c12 = fma(wy, vz, c12); c22 = fma(wz, vz, c22);
            sum += w; valid++;
        }
    });
    c00 = wave_sum_d(c00); c01 = wave_sum_d(c01); c02 = wave_sum_d(c02);
    c11 = wave_sum_d(c11); c12 = wave_sum_d(c12); c22 = wave_sum_d(c22);
    sum = wave_sum_d(sum); valid = wave_sum_i(valid);
    if (lane == 0) {
        double* c = cov_out + (size_t)k * 8;
        c[0] = c00; c[1] = c01; c[2] = c02; c[3] = c11; c[4] = c12; c[5] = c22; c[6] = sum; c[7] = (double)valid;
    }
}

// axes_out[k*6 + 0..2] = x candidate (largest eigenvalue), [3..5] = z candidate (smallest); invalid frames are written as NaN
__global__ __launch_bounds__(256) void k_lrf_eig(uint32_t nkp, const double* __restrict__ cov, double* __restrict__ axes_out,
                                                 float* __restrict__ lrf_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkp) return;
    const double* c = cov + (size_t)k * 8;
    const double valid = c[7];
    bool bad = valid < 5.0;                                                   // shot_na_lrf.hpp:80-86
    double w[3], V[3][3];
    if (!bad) {
        const double sum = c[6];
        double A[3][3] = {{c[0] / sum, c[1] / sum, c[2] / sum}, {c[1] / sum, c[3] / sum, c[4] / sum}, {c[2] / sum, c[4] / sum, c[5] / sum}};
        eigen_sym3(A, w, V);
        bad = !(isfinite(w[0]) && isfinite(w[1]) && isfinite(w[2]));
    }
    double* a = axes_out + (size_t)k * 6;
    if (bad) {
        for (int i = 0; i < 9; ++i) lrf_out[(size_t)k * 9 + i] = __builtin_nanf("");
        a[0] = __builtin_nan("");
        return;
    }
    a[0] = V[0][2]; a[1] = V[1][2]; a[2] = V[2][2];
    a[3] = V[0][0]; a[4] = V[1][0]; a[5] = V[2][0];
}

__global__ __launch_bounds__(256, 4) void k_lrf_sign(CloudView cv, const uint32_t* __restrict__ kp_off,
                                                  const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ kz,
                                                  float radius, float r2, const double* __restrict__ cov, const double* __restrict__ axes,
                                                  float* __restrict__ lrf_out, uint32_t* __restrict__ tie_count, TieRec* __restrict__ tie_rec) {
    int o, bx;
    if (!xcd_object_block(cv.nbx, cv.n_obj, o, bx)) return;
    const uint32_t k = kp_off[o] + bx * 4 + (threadIdx.x >> 6);
    if (k >= kp_off[o + 1]) return;
    const double* a = axes + (size_t)k * 6;
    if (isnan(a[0]) || cov[(size_t)k * 8 + 7] < 5.0) return;             // frame already written as NaN
    const int lane = lane_id();
    const int valid = (int)cov[(size_t)k * 8 + 7];
    double v1[3] = {a[0], a[1], a[2]}, v3[3] = {a[3], a[4], a[5]};
    const float cx = kx[k], cy = ky[k], cz = kz[k];
    const GridMeta m = cv.meta[o];
    const uint32_t* cs = cv.cell_start + (size_t)o * ISM_GRID_STRIDE;
    const uint32_t base = cv.pt_off[o];
    CellRange cr;
    ball_cells(m, cx, cy, cz, radius, cr);
    int plusT = 0, plusN = 0;
    __shared__ WaveRows s_rows[4];
    ball_for_each(m, cs, cr, cx, cy, cz, radius, lane, s_rows[threadIdx.x >> 6],
                  [&](uint32_t t, bool) { return cv.sp4[base + t]; },
                  [&](const float4& p, uint32_t, bool v) {
        if (!v) return;
        const float px = p.x, py = p.y, pz = p.z;
        const float d2 = sqdist3(px, py, pz, cx, cy, cz);
        if (d2 < r2 && !(px == cx && py == cy && pz == cz)) {
            const double vx = (double)(px - cx), vy = (double)(py - cy), vz = (double)(pz - cz);
            if (vx * v1[0] + vy * v1[1] + vz * v1[2] >= 0) plusT++;
            if (vx * v3[0] + vy * v3[1] + vz * v3[2] >= 0) plusN++;
        }
    });
    plusT = 2 * wave_sum_i(plusT) - valid;
    plusN = 2 * wave_sum_i(plusN) - valid;
    if (plusT < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
    if (plusN < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
    if (plusT == 0 || plusN == 0) {
        if (lane == 0) {
            const uint32_t slot = atomicAdd(tie_count, 1u);
            TieRec r;
            for (int i = 0; i < 3; ++i) { r.v1[i] = v1[i]; r.v3[i] = v3[i]; }
            r.kp = k; r.obj = (uint32_t)o; r.valid = valid; r.tie_x = plusT == 0; r.tie_z = plusN == 0;
            tie_rec[slot] = r;
        }
        return;   // finished by k_lrf_tie
    }
    if (lane == 0) write_lrf(lrf_out + (size_t)k * 9, v1, v3);
}

#define TIE_LDS_KEYS 8192     // keys of one neighbourhood in LDS (64 KB, one wave per workgroup); larger ones go to a global scratch row
#define TIE_REG_KEYS 20      // keys per lane held in registers by the fast selection path (neighbourhoods up to 1280 points)
__global__ __launch_bounds__(64) void k_lrf_tie(CloudView cv, const float* __restrict__ kx, const float* __restrict__ ky,
                                                 const float* __restrict__ kz, float radius, float r2,
                                                 float* __restrict__ lrf_out, const uint32_t* __restrict__ tie_count,
                                                 const TieRec* __restrict__ tie_rec, unsigned long long* __restrict__ keys,
                                                 uint32_t key_cap) {
    // keys live in LDS when the neighbourhood fits (the common case), else in a per-wave global scratch row
    __shared__ unsigned long long s_keys[1][TIE_LDS_KEYS];
    __shared__ WaveRows s_rows[1];
    const int lane = lane_id();
    const uint32_t gw = blockIdx.x;
    const uint32_t nw = gridDim.x;
    const uint32_t n_tie = *tie_count;
    unsigned long long* gkeys = keys + (size_t)gw * key_cap;
    for (uint32_t t = gw; t < n_tie; t += nw) {
        const TieRec r = tie_rec[t];
        const int o = (int)r.obj;
        const uint32_t k = r.kp;
        const float cx = kx[k], cy = ky[k], cz = kz[k];
        const GridMeta m = cv.meta[o];
        const uint32_t* cs = cv.cell_start + (size_t)o * ISM_GRID_STRIDE;
        const uint32_t base = cv.pt_off[o];
        CellRange cr;
        ball_cells(m, cx, cy, cz, radius, cr);
        // r.valid neighbours will be collected (same test as k_lrf_cov): choose the key store up front (wave-uniform)
        unsigned long long* mykeys = r.valid <= TIE_LDS_KEYS ? s_keys[threadIdx.x >> 6] : gkeys;
        const uint32_t cap = r.valid <= TIE_LDS_KEYS ? (uint32_t)TIE_LDS_KEYS : key_cap;
        uint32_t n = 0;
        ball_for_each(m, cs, cr, cx, cy, cz, radius, lane, s_rows[threadIdx.x >> 6],
                      [&](uint32_t i, bool) { return cv.sp4[base + i]; },
                      [&](const float4& p, uint32_t, bool v) {
            bool pass = false; float d2 = 0.f; uint32_t orig = 0;
            if (v) {
                d2 = sqdist3(p.x, p.y, p.z, cx, cy, cz);
                pass = d2 < r2 && !(p.x == cx && p.y == cy && p.z == cz);
                orig = __float_as_uint(p.w);
            }
            const unsigned long long mask = __ballot(pass);
            if (pass) {
                const uint32_t pos = n + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cap) mykeys[pos] = ((unsigned long long)__float_as_uint(d2) << 32) | orig;
            }
            n += __popcll(mask);
        });
        __threadfence_block();
        if (n > cap) n = cap;   // cannot happen: n == r.valid <= cap
        // rank selection by bitwise bisection: the element of rank t is the largest v with #(key < v) <= t.
        // Keys are (positive float bits << 32 | index < key_cap): only bits 62..32 and the low index bits can be set.
        const int median = (int)n / 2;
        const int tsel = median - 2;
        int idx_bits = 1; while ((1u << idx_bits) < key_cap) ++idx_bits;
        unsigned long long five[5];
        if (n <= 64 * TIE_REG_KEYS) {
            // the common case: the keys fit the register file (TIE_REG_KEYS per lane), every round is compares + one DPP wave sum
            unsigned long long kr[TIE_REG_KEYS];
#pragma unroll
            for (int j = 0; j < TIE_REG_KEYS; ++j) kr[j] = (uint32_t)(j * 64 + lane) < n ? mykeys[j * 64 + lane] : ~0ull;
            unsigned long long sel = 0ull;
            for (int bit = 62; bit >= 0; --bit) {
                if (bit < 32 && bit >= idx_bits) continue;
                const unsigned long long cand = sel | (1ull << bit);
                int c = 0;
#pragma unroll
                for (int j = 0; j < TIE_REG_KEYS; ++j) c += kr[j] < cand;
                c = wave_sum_i(c);
                if (c <= tsel) sel = cand;
            }
            five[0] = sel;
            for (int j = 1; j < 5; ++j) {
                unsigned long long mn = ~0ull;
#pragma unroll
                for (int x = 0; x < TIE_REG_KEYS; ++x) { const unsigned long long kk = kr[x]; if (kk > five[j - 1] && kk < mn) mn = kk; }
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) { const unsigned long long tt = __shfl_xor(mn, o2, 64); mn = tt < mn ? tt : mn; }
                five[j] = mn;
            }
        } else {
            unsigned long long sel = 0ull;
            for (int bit = 62; bit >= 0; --bit) {
                if (bit < 32 && bit >= idx_bits) continue;
                const unsigned long long cand = sel | (1ull << bit);
                int c = 0;
                uint32_t i = lane;
                for (; i + 192 < n; i += 256) {          // four independent LDS reads in flight
                    const unsigned long long k0 = mykeys[i], k1 = mykeys[i + 64], k2 = mykeys[i + 128], k3 = mykeys[i + 192];
                    c += (k0 < cand) + (k1 < cand) + (k2 < cand) + (k3 < cand);
                }
                for (; i < n; i += 64) c += mykeys[i] < cand;
                c = wave_sum_i(c);
                if (c <= tsel) sel = cand;
            }
            // sel = key of rank median-2; the next four follow by successive minima above the previous one
            five[0] = sel;
            for (int j = 1; j < 5; ++j) {
                unsigned long long mn = ~0ull;
                for (uint32_t i = lane; i < n; i += 64) { const unsigned long long kk = mykeys[i]; if (kk > five[j - 1] && kk < mn) mn = kk; }
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) { const unsigned long long tt = __shfl_xor(mn, o2, 64); mn = tt < mn ? tt : mn; }
                five[j] = mn;
            }
        }
        int cntx = 0, cntz = 0;
        if (lane < 5) {
            unsigned long long ki = five[0];
            if (lane == 1) ki = five[1]; else if (lane == 2) ki = five[2]; else if (lane == 3) ki = five[3]; else if (lane == 4) ki = five[4];
            const uint32_t orig = (uint32_t)(ki & 0xffffffffull);
            const double vx = (double)(cv.x[base + orig] - cx), vy = (double)(cv.y[base + orig] - cy), vz = (double)(cv.z[base + orig] - cz);
            if (vx * r.v1[0] + vy * r.v1[1] + vz * r.v1[2] > 0) cntx++;
            if (vx * r.v3[0] + vy * r.v3[1] + vz * r.v3[2] > 0) cntz++;
        }
        cntx = wave_sum_i(cntx); cntz = wave_sum_i(cntz);
        double v1[3] = {r.v1[0], r.v1[1], r.v1[2]}, v3[3] = {r.v3[0], r.v3[1], r.v3[2]};
        if (r.tie_x && cntx < 3) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
        if (r.tie_z && cntz < 3) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
        if (lane == 0) write_lrf(lrf_out + (size_t)k * 9, v1, v3);
    }
}

}  // namespace

extern "C" int ismhip_shot_lrf(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                               const float* kpx, const float* kpy, const float* kpz, float radius, float* lrf9_out) {
    if (!ctx || !cloud || !kp_offsets_h || !kpx || !kpy || !kpz || !lrf9_out || !(radius > 0.f))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "shot_lrf: bad argument");
    const int n_obj = cloud->n_obj;
    uint32_t maxk = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (kp_offsets_h[o + 1] < kp_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "shot_lrf: offsets not monotone");
        maxk = std::max(maxk, kp_offsets_h[o + 1] - kp_offsets_h[o]);
    }
    if (kp_offsets_h[0] != 0) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "shot_lrf: offsets must start at 0");
    const uint32_t nkp = kp_offsets_h[n_obj];
    if (nkp == 0 || maxk == 0) return ISMHIP_OK;
    uint32_t* ko = ism_upload_offsets(ctx, SCR_KP_OFF, kp_offsets_h, n_obj + 1);
    if (!ko) return ISMHIP_ERR_HIP;
    uint32_t* tie_count = (uint32_t*)ism_scratch(ctx, SCR_COUNTERS, 64);
    TieRec* tie_rec = (TieRec*)ism_scratch(ctx, SCR_TIE_REC, (size_t)nkp * sizeof(TieRec));
    double* cov = (double*)ism_scratch(ctx, SCR_LRF_COV, (size_t)nkp * 14 * sizeof(double));
    const int tie_blocks = 1024;  // one wave per workgroup
    const uint32_t key_cap = cloud->max_pts ? cloud->max_pts : 1;
    unsigned long long* keys = (unsigned long long*)ism_scratch(ctx, SCR_TIE_KEYS, (size_t)tie_blocks * key_cap * 8);
    if (!tie_count || !tie_rec || !keys || !cov) return ISMHIP_ERR_NOMEM;
    double* axes = cov + (size_t)nkp * 8;
    CloudView cv{cloud->pt_off, cloud->meta, cloud->cell_start, cloud->sp4, cloud->x, cloud->y, cloud->z,
                 ctx->xcd_map ? n_obj : 0, (int)((maxk + 3) / 4)};
    const float r2 = (float)((double)radius * (double)radius);   // PCL: static_cast<float>(radius*radius) with double radius
    TimerScope ts(ctx, "lrf");
    ISM_HIP(ctx, hipMemsetAsync(tie_count, 0, 4, ctx->stream));
    const dim3 grid(ctx->xcd_map ? xcd_object_grid((maxk + 3) / 4, n_obj) : ((maxk + 3) / 4) * (unsigned)n_obj);
    hipLaunchKernelGGL(k_lrf_cov, grid, dim3(256), 0, ctx->stream, cv, ko, kpx, kpy, kpz, radius, r2, cov);
    ISM_CHECK_LAUNCH(ctx, "k_lrf_cov");
    hipLaunchKernelGGL(k_lrf_eig, dim3((nkp + 255) / 256), dim3(256), 0, ctx->stream, nkp, cov, axes, lrf9_out);
    ISM_CHECK_LAUNCH(ctx, "k_lrf_eig");
    hipLaunchKernelGGL(k_lrf_sign, grid, dim3(256), 0, ctx->stream, cv, ko, kpx, kpy, kpz, radius, r2, cov, axes, lrf9_out, tie_count, tie_rec);
    ISM_CHECK_LAUNCH(ctx, "k_lrf_sign");
    hipLaunchKernelGGL(k_lrf_tie, dim3(tie_blocks), dim3(64), 0, ctx->stream, cv, kpx, kpy, kpz, radius, r2, lrf9_out, tie_count, tie_rec, keys, key_cap);
    ISM_CHECK_LAUNCH(ctx, "k_lrf_tie");
    return ISMHIP_OK;
}


// ---- normals from the SHOT frame (the step in front of the path) -------------------------------------------
// Reference seam: ImplicitShapeModel::computeNormals with ConsistentNormalsMethod 2, the default (implicit_shape_model.cpp:
// 1014-1018) -> NormalOrientation::processSHOTLRF (utils/normal_orientation.cpp:48-110): a SHOT frame with radius NormalRadius at
// EVERY point of the cloud, normal = inverted z axis. The arrays hold the PCA normals (flipped towards the origin) when this runs
// (implicit_shape_model.cpp:1016), so a point whose frame is invalid (< 5 neighbours) KEEPS its PCA normal. The reference then means
// to recompute the k invalid ones but indexes its loop with the loop counter (normal_orientation.cpp:96-106): points 0..k-1 of the
// NaN-free cloud get NormalEstimation::computePointNormal's normal instead -- the same PCA normal WITHOUT the viewpoint flip (sign =
// pcl::eigen33's) -- whether their frame was valid or not. Reproduced as is: one workgroup per object, finite points in input order.
namespace {
__global__ __launch_bounds__(256) void k_normals_from_lrf(const uint32_t* __restrict__ pt_off, const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z, const float* __restrict__ lrf, const uint8_t* __restrict__ rawflip,
                                                          float* __restrict__ nx, float* __restrict__ ny, float* __restrict__ nz) {
    const uint32_t base = pt_off[blockIdx.x], n = pt_off[blockIdx.x + 1] - base;
    __shared__ uint32_t s_cnt[4], s_inv;
    auto finite_pt = [&](uint32_t i) { return isfinite(x[base + i]) && isfinite(y[base + i]) && isfinite(z[base + i]); };
    auto frame_ok = [&](uint32_t i) { const float* f = lrf + (size_t)(base + i) * 9; return isfinite(f[0]) && isfinite(f[3]) && isfinite(f[6]); };
    if (threadIdx.x == 0) s_inv = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) mine += (finite_pt(i) && !frame_ok(i)) ? 1u : 0u;
    mine = (uint32_t)wave_sum_i((int)mine);
    if (lane_id() == 0 && mine) atomicAdd(&s_inv, mine);
    __syncthreads();
    const uint32_t n_inv = s_inv;
    uint32_t rank0 = 0;                                                         // finite points in front of this chunk
    for (uint32_t c = 0; c < n; c += 256) {
        const uint32_t i = c + threadIdx.x;
        const bool fin = i < n && finite_pt(i);
        const unsigned long long b = __ballot(fin);
        if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t r = rank0 + (uint32_t)__popcll(b & ((1ull << lane_id()) - 1ull));
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) r += s_cnt[w];
        const uint32_t chunk = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
        if (fin) {
            const uint32_t g = base + i;
            if (r < n_inv) { if (rawflip[g]) { nx[g] = -nx[g]; ny[g] = -ny[g]; nz[g] = -nz[g]; } }
            else if (frame_ok(i)) { const float* f = lrf + (size_t)g * 9; nx[g] = -f[6]; ny[g] = -f[7]; nz[g] = -f[8]; }
        }
        rank0 += chunk;
    }
}
// the cloud's cell-sorted normal copies follow (sorted position s of object o holds the original point in sp4[s].w)
__global__ __launch_bounds__(256) void k_sorted_normals(const uint32_t* __restrict__ pt_off, const GridMeta* __restrict__ meta, const float4* __restrict__ sp4,
                                                        const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                                        float4* __restrict__ sn4) {
    const int o = blockIdx.y;
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= meta[o].n_finite) return;
    const uint32_t base = pt_off[o], src = base + __float_as_uint(sp4[base + s].w);
    sn4[base + s] = make_float4(nx[src], ny[src], nz[src], 0.f);
}
}  // namespace

// ---- PCA normals: ImplicitShapeModel::computeNormals, ConsistentNormalsMethod 0 and 1 (implicit_shape_model.cpp:969-1011) ->
// pcl::NormalEstimationOMPWithEigVals (third_party/pcl_normal_3d_omp_with_eigenvalues/normal_3d_omp_with_eigenvalues.hpp:61-144,
// .h:112-180): neighbours within NormalRadius (the point itself included), fewer than 3 -> NaN; normal = eigenvector of the smallest
// eigenvalue of the neighbourhood covariance, flipped towards the viewpoint. Method 0: viewpoint (0,0,0). Method 1: the cloud is
// shifted by its centroid, normals are flipped towards the origin and then inverted, i.e. they point AWAY from the centroid.
// One wave per (cell-sorted) point: FP64 moments of (p - q) per lane + wave reduction (the reference's single-pass float formula
// E[xx] - E[x]E[x] on absolute coordinates loses ~4 digits on unit-sized objects; this is the better-conditioned evaluation of the
// same covariance), Jacobi solver of eigen3.h, results scattered to the caller's (original-order) arrays and into the sorted copy.
namespace {
__device__ __forceinline__ void cross3(const double* u, const double* v, double* w) {
    w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0];
}
__global__ __launch_bounds__(256) void k_pca_normals(CloudView cv, float radius, float r2, int orientation,
                                                     float* __restrict__ nx, float* __restrict__ ny, float* __restrict__ nz, float4* __restrict__ sn4,
                                                     uint8_t* __restrict__ rawflip) {
    int o, bx;
    if (!xcd_object_block(cv.nbx, cv.n_obj, o, bx)) return;
    const GridMeta m = cv.meta[o];
    const uint32_t sidx = (uint32_t)bx * 4 + (threadIdx.x >> 6);
    if (sidx >= m.n_finite) return;
    const int lane = lane_id();
    const uint32_t base = cv.pt_off[o];
    const float4 q = cv.sp4[base + sidx];
    const uint32_t* cs = cv.cell_start + (size_t)o * ISM_GRID_STRIDE;
    CellRange cr;
    double sx = 0, sy = 0, sz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    int cnt = 0;
    __shared__ WaveRows s_rows[4];
    if (ball_cells(m, q.x, q.y, q.z, radius, cr))
        ball_for_each(m, cs, cr, q.x, q.y, q.z, radius, lane, s_rows[threadIdx.x >> 6],
                      [&](uint32_t t, bool) { return cv.sp4[base + t]; },
                      [&](const float4& p, uint32_t, bool v) {
            if (!v) return;
            if (sqdist3(p.x, p.y, p.z, q.x, q.y, q.z) < r2) {
                const double dx = (double)p.x - (double)q.x, dy = (double)p.y - (double)q.y, dz = (double)p.z - (double)q.z;
                sx += dx; sy += dy; sz += dz;
                xx = fma(dx, dx, xx); xy = fma(dx, dy, xy); xz = fma(dx, dz, xz); yy = fma(dy, dy, yy); yz = fma(dy, dz, yz); zz = fma(dz, dz, zz);
                cnt++;
            }
        });
    sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
    xx = wave_sum_d(xx); xy = wave_sum_d(xy); xz = wave_sum_d(xz); yy = wave_sum_d(yy); yz = wave_sum_d(yz); zz = wave_sum_d(zz);
    cnt = wave_sum_i(cnt);
    if (lane != 0) return;
    const uint32_t orig = base + __float_as_uint(q.w);
    float n0 = __builtin_nanf(""), n1 = n0, n2 = n0;
    if (cnt >= 3) {
        const double inv = 1.0 / (double)cnt, mx = sx * inv, my = sy * inv, mz = sz * inv;
        double A[3][3] = {{xx * inv - mx * mx, xy * inv - mx * my, xz * inv - mx * mz}, {xy * inv - mx * my, yy * inv - my * my, yz * inv - my * mz},
                          {xz * inv - mx * mz, yz * inv - my * mz, zz * inv - mz * mz}};
        double w[3], V[3][3];
        eigen_sym3(A, w, V);
        n0 = (float)V[0][0]; n1 = (float)V[1][0]; n2 = (float)V[2][0];
        bool raw_neg = false;
        if (rawflip) {
            // the sign pcl::eigen33 gives its eigenvector (no viewpoint flip: NormalEstimation::computePointNormal): the largest of
            // the three cross products of rows of (A / scale - lambda I)
            double sc = 0;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sc = fmax(sc, fabs(A[r][c]));
            if (!(sc > 0)) sc = 1.0;
            double B[3][3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[r][c] = A[r][c] / sc - (r == c ? w[0] / sc : 0.0);
            double c1[3], c2[3], c3[3];
            cross3(B[0], B[1], c1); cross3(B[0], B[2], c2); cross3(B[1], B[2], c3);
            const double l1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2], l2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2], l3 = c3[0] * c3[0] + c3[1] * c3[1] + c3[2] * c3[2];
            const double* cb = (l1 >= l2 && l1 >= l3) ? c1 : (l2 >= l1 && l2 >= l3) ? c2 : c3;
            raw_neg = cb[0] * V[0][0] + cb[1] * V[1][0] + cb[2] * V[2][0] < 0;
        }
        // flipNormalTowardsViewpointMod (.h:162-180): flip when (viewpoint - point) . n < 0
        const float vx = (orientation == 1 ? m.centroid[0] : 0.f) - q.x, vy = (orientation == 1 ? m.centroid[1] : 0.f) - q.y, vz = (orientation == 1 ? m.centroid[2] : 0.f) - q.z;
        bool neg = false;
        if (vx * n0 + vy * n1 + vz * n2 < 0) neg = !neg;
        if (orientation == 1) neg = !neg;                                      // implicit_shape_model.cpp:996-1002
        if (neg) { n0 = -n0; n1 = -n1; n2 = -n2; }
        if (rawflip) rawflip[orig] = (neg != raw_neg) ? 1 : 0;                 // 1: the stored normal is the negated eigen33 vector
    }
    nx[orig] = n0; ny[orig] = n1; nz[orig] = n2;
    sn4[base + sidx] = make_float4(n0, n1, n2, 0.f);
}
}  // namespace

static int pca_normals(ismhip_ctx* ctx, ismhip_cloud* cloud, float radius, int orientation, float* nx_out, float* ny_out, float* nz_out, uint8_t* rawflip) {
    const uint32_t n = cloud->n_pts;
    if (n == 0 || cloud->max_pts == 0) return ISMHIP_OK;
    const int n_obj = cloud->n_obj;
    const unsigned nbx = (cloud->max_pts + 3) / 4;
    CloudView cv{cloud->pt_off, cloud->meta, cloud->cell_start, cloud->sp4, cloud->x, cloud->y, cloud->z, ctx->xcd_map ? n_obj : 0, (int)nbx};
    const float r2 = (float)((double)radius * (double)radius);
    TimerScope ts(ctx, "normals_pca");
    // points that are not finite never enter the sorted copy: their normals are NaN
    ISM_HIP(ctx, hipMemsetAsync(nx_out, 0xff, (size_t)n * 4, ctx->stream));
    ISM_HIP(ctx, hipMemsetAsync(ny_out, 0xff, (size_t)n * 4, ctx->stream));
    ISM_HIP(ctx, hipMemsetAsync(nz_out, 0xff, (size_t)n * 4, ctx->stream));
    const dim3 grid(ctx->xcd_map ? xcd_object_grid(nbx, n_obj) : nbx * (unsigned)n_obj);
    hipLaunchKernelGGL(k_pca_normals, grid, dim3(256), 0, ctx->stream, cv, radius, r2, orientation, nx_out, ny_out, nz_out, cloud->sn4, rawflip);
    ISM_CHECK_LAUNCH(ctx, "k_pca_normals");
    cloud->nx = nx_out; cloud->ny = ny_out; cloud->nz = nz_out;
    return ISMHIP_OK;
}
extern "C" int ismhip_estimate_normals_pca(ismhip_ctx* ctx, ismhip_cloud* cloud, float radius, int orientation, float* nx_out, float* ny_out, float* nz_out) {
    if (!ctx || !cloud || !nx_out || !ny_out || !nz_out || !(radius > 0.f) || (orientation != 0 && orientation != 1))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "estimate_normals_pca: bad argument");
    return pca_normals(ctx, cloud, radius, orientation, nx_out, ny_out, nz_out, nullptr);
}

extern "C" int ismhip_estimate_normals(ismhip_ctx* ctx, ismhip_cloud* cloud, float radius, float* nx_out, float* ny_out, float* nz_out) {
    if (!ctx || !cloud || !nx_out || !ny_out || !nz_out || !(radius > 0.f)) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "estimate_normals: bad argument");
    const uint32_t n = cloud->n_pts;
    if (n == 0) return ISMHIP_OK;
    float* lrf = (float*)ism_scratch(ctx, SCR_FPFH_SPFH, (size_t)n * 9 * sizeof(float) + n);
    if (!lrf) return ISMHIP_ERR_NOMEM;
    uint8_t* rawflip = (uint8_t*)(lrf + (size_t)n * 9);
    int rc = pca_normals(ctx, cloud, radius, 0, nx_out, ny_out, nz_out, rawflip);
    if (rc != ISMHIP_OK) return rc;
    rc = ismhip_shot_lrf(ctx, cloud, cloud->pt_off_h.data(), cloud->x, cloud->y, cloud->z, radius, lrf);
    if (rc != ISMHIP_OK) return rc;
    hipLaunchKernelGGL(k_normals_from_lrf, dim3(cloud->n_obj), dim3(256), 0, ctx->stream, cloud->pt_off, cloud->x, cloud->y, cloud->z, lrf, rawflip,
                       nx_out, ny_out, nz_out);
    ISM_CHECK_LAUNCH(ctx, "k_normals_from_lrf");
    hipLaunchKernelGGL(k_sorted_normals, dim3((cloud->max_pts + 255) / 256, cloud->n_obj), dim3(256), 0, ctx->stream, cloud->pt_off, cloud->meta, cloud->sp4,
                       nx_out, ny_out, nz_out, cloud->sn4);
    ISM_CHECK_LAUNCH(ctx, "k_sorted_normals");
    cloud->nx = nx_out; cloud->ny = ny_out; cloud->nz = nz_out;
    return ISMHIP_OK;
}
