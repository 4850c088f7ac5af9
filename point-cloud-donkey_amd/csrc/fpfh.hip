// fpfh.hip — FPFH-33 descriptors. Reference seam: FeaturesFPFH::iComputeDescriptors
// (features/features_fpfh.cpp:27-72) -> pcl::FPFHEstimationOMP<..., FPFHSignature33>; arithmetic restated from
// PCL 1.10 computePairFeatures / computePointSPFHSignature / weightPointSPFHSignature.
//
// Three launches, one wavefront per unit of work:
//   k_fpfh_mark : per keypoint, flag every surface point inside its ball (the union U of A.4) and count M_k
//   k_spfh      : per flagged surface point p, SPFH(p) = 3x11 histogram of Darboux angles to its neighbours
//   k_fpfh_sum  : per keypoint, FPFH = sum_nb SPFH(nb)/d^2, every 11-bin block rescaled to 100
// Roofline (HBM gather model): sum_{p in U} M_p*24 + |U|*132 + sum_k M_k*136 + K*132 bytes.
#include "common.h"
#include <cstdlib>

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {

struct FpfhArgs {
    const uint32_t* pt_off; const GridMeta* meta; const uint32_t* cell_start;
    const float4 *sp4, *sn4;
    const uint32_t* kp_off; const float *kx, *ky, *kz;
    float radius, r2;
    uint8_t* flag;      // [n_pts] sorted index space
    float* spfh;        // [n_pts*33]
    float* desc; uint32_t* count;
    uint32_t max_pts;
    int n_obj, nbx_kp;           // XCD-local block map (common.h) of k_fpfh_mark / k_fpfh_sum: an object's SPFH rows (2.2 MB at 16384 points) stay in ONE L2
    int dbg;            // env ISMHIP_FPFH_DBG (timing experiments, results invalid): 1 = k_spfh without the pair features, 2 = every pair by the exact arithmetic, 3 = fast arithmetic only (A/B runs: 21.1 / 21.1 / 23.0 ms per 128 objects -- the pair arithmetic is not what bounds k_spfh)
};

__global__ __launch_bounds__(256) void k_fpfh_mark(FpfhArgs a) {
    int o, bx;
    if (!xcd_object_block(a.nbx_kp, a.n_obj, o, bx)) return;
    const uint32_t k = a.kp_off[o] + bx * 4 + (threadIdx.x >> 6);
    if (k >= a.kp_off[o + 1]) return;
    const int lane = lane_id();
    const float cx = a.kx[k], cy = a.ky[k], cz = a.kz[k];
    const GridMeta m = a.meta[o];
    CellRange cr;
    uint32_t total = 0;
    if (isfinite(cx) && isfinite(cy) && isfinite(cz) && ball_cells(m, cx, cy, cz, a.radius, cr)) {
        const uint32_t* cs = a.cell_start + (size_t)o * ISM_GRID_STRIDE;
        const uint32_t base = a.pt_off[o];
        for (int gz = cr.lo[2]; gz <= cr.hi[2]; ++gz)
            for (int gy = cr.lo[1]; gy <= cr.hi[1]; ++gy) {
                int xl, xh;
                if (!row_cells(m, cr, gy, gz, cx, cy, cz, a.radius, xl, xh)) continue;
                const int rb = (gz * m.dim[1] + gy) * m.dim[0];
                const uint32_t s = cs[rb + xl], e = cs[rb + xh + 1];
                for (uint32_t t = s + lane; t < e; t += 64) {
                    const float4 q = a.sp4[base + t];
                    const float d2 = sqdist3(q.x, q.y, q.z, cx, cy, cz);
                    if (d2 < a.r2) { a.flag[base + t] = 1; total++; }
                }
            }
    }
    total = (uint32_t)wave_sum_i((int)total);
    if (a.count && lane == 0) a.count[k] = total;
}

// pcl::computePairFeatures in float; returns false when the pair is skipped
__device__ __forceinline__ bool pair_features(float px, float py, float pz, float pnx, float pny, float pnz,
                                              float qx, float qy, float qz, float qnx, float qny, float qnz,
                                              float& f1, float& f2, float& f3) {
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float f4 = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (f4 == 0.0f) return false;
    float ax = pnx, ay = pny, az = pnz, bx = qnx, by = qny, bz = qnz;
    const float angle1 = ((ax * dx + ay * dy) + az * dz) / f4;
    const float angle2 = ((bx * dx + by * dy) + bz * dz) / f4;
    // PCL swaps the roles when acos|a1| > acos|a2|. acos is decreasing with |slope| >= 1, so when the two absolute cosines differ
    // by more than 1e-5 (hundreds of float acosf errors) the order of the acos values is the reverse order of the cosines and no
    // acosf is needed; inside that band (and only there) the reference's own comparison of the two acosf values decides.
    const float c1 = fabsf(angle1), c2 = fabsf(angle2);
    const float gap = c2 - c1;
    const bool swap_roles = fabsf(gap) > 1e-5f ? gap > 0.f : acosf(c1) > acosf(c2);
    if (swap_roles) {
        float t;
        t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; t = az; az = bz; bz = t;
        dx = -dx; dy = -dy; dz = -dz;
        f3 = -angle2;
    } else f3 = angle1;
    float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
    const float vn = sqrtf((vx * vx + vy * vy) + vz * vz);
    if (vn == 0.0f) return false;
    vx /= vn; vy /= vn; vz /= vn;
    const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
    f2 = (vx * bx + vy * by) + vz * bz;
    f1 = atan2f((wx * bx + wy * by) + wz * bz, (ax * bx + ay * by) + az * bz);
    return true;
}

__device__ __forceinline__ int clamp_bin(int h) { return h < 0 ? 0 : (h > 10 ? 10 : h); }

// The same three features by FAST arithmetic (v_rsq_f32 / v_rcp_f32 instead of sqrt + IEEE divisions, a degree-13 odd polynomial
// for the arctangent: max error 6.6e-7 rad over [0, 1], fitted and checked in tests/test_host_cpu.py) and the three bin
// coordinates t = 11 (f + pi) / 2pi, 11 (f + 1) / 2 in float. The histogram only needs floor(t): when every t is at least
// FPFH_GUARD away from the integers 1..10 the fast bins ARE the reference's bins (the fast values differ from the exact ones by
// < 2e-5 in t), otherwise -- and for the degenerate pairs -- the caller evaluates pair_features + the double-precision bin formulas
// exactly as the reference does. `sure` = the fast bins can be used. The role swap (which decides everything downstream) is
// taken exactly as in pair_features.
#define FPFH_GUARD 1e-4f
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);                     // 0/0 -> NaN: the caller's guard test fails and the exact path runs
    const float z = a * a;
    float p = 0.008097294718027115f;
    p = fmaf(p, z, -0.037751708179712296f); p = fmaf(p, z, 0.08475969731807709f); p = fmaf(p, z, -0.13537675142288208f);
    p = fmaf(p, z, 0.19895026087760925f); p = fmaf(p, z, -0.3332797586917877f); p = fmaf(p, z, 0.9999997019767761f);
    float r = a * p;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return y < 0.f ? -r : r;
}
__device__ __forceinline__ bool fpfh_bin_sure(float t) {               // false for NaN
    const float fr = t - floorf(t);
    return (fr > FPFH_GUARD && fr < 1.0f - FPFH_GUARD) || t < 1.0f - FPFH_GUARD || t > 10.0f + FPFH_GUARD;   // only the integers 1..10 separate bins (0 and 11 are clamped away)
}
__device__ __forceinline__ bool pair_bins_fast(float px, float py, float pz, float pnx, float pny, float pnz,
                                               float qx, float qy, float qz, float qnx, float qny, float qnz,
                                               int& h1, int& h2, int& h3) {
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 == 0.0f) return false;                                       // the exact path skips the pair: let it
    const float inv_f4 = __builtin_amdgcn_rsqf(d2);
    float ax = pnx, ay = pny, az = pnz, bx = qnx, by = qny, bz = qnz;
    // the swap test needs the reference's own angle values near a tie: exact division there, reciprocal elsewhere
    const float dot1 = (ax * dx + ay * dy) + az * dz, dot2 = (bx * dx + by * dy) + bz * dz;
    float angle1 = dot1 * inv_f4, angle2 = dot2 * inv_f4;
    const float gapf = fabsf(angle2) - fabsf(angle1);
    if (!(fabsf(gapf) > 1e-4f)) return false;                           // near tie of the two cosines (or NaN): exact path decides the roles
    float f3;
    if (gapf > 0.f) {
        float t;
        t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; t = az; az = bz; bz = t;
        dx = -dx; dy = -dy; dz = -dz;
        f3 = -angle2;
    } else f3 = angle1;
    float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
    const float vn2 = (vx * vx + vy * vy) + vz * vz;
    if (!(vn2 > 1e-30f)) return false;                                  // degenerate (or denormal): exact path
    const float inv_vn = __builtin_amdgcn_rsqf(vn2);
    vx *= inv_vn; vy *= inv_vn; vz *= inv_vn;
    const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
    const float f2 = (vx * bx + vy * by) + vz * bz;
    const float ay_ = (wx * bx + wy * by) + wz * bz, ax_ = (ax * bx + ay * by) + az * bz;
    // the arctangent is only as well conditioned as |(x, y)| is large: the two arguments carry ~3e-7 of fast-arithmetic error, which
    // is 1.5e-5 rad = 2.6e-5 of a bin at |(x, y)| = 0.02 (unit normals: |(x, y)|^2 = 1 - f2^2); closer to the pole the exact path runs
    if (!((ax_ * ax_ + ay_ * ay_) > 4e-4f)) return false;
    const float f1 = fast_atan2(ay_, ax_);
    const float t1 = 11.0f * ((f1 + 3.14159265358979323846f) * 0.15915494309189535f);
    const float t2 = 11.0f * ((f2 + 1.0f) * 0.5f), t3 = 11.0f * ((f3 + 1.0f) * 0.5f);
    if (!(fpfh_bin_sure(t1) && fpfh_bin_sure(t2) && fpfh_bin_sure(t3))) return false;
    h1 = clamp_bin((int)floorf(t1)); h2 = clamp_bin((int)floorf(t2)); h3 = clamp_bin((int)floorf(t3));
    return true;
}

__global__ __launch_bounds__(256) void k_spfh(FpfhArgs a) {
    __shared__ WaveRows s_rows[4];
    __shared__ float4 s_queue[4][128];      // queued neighbours: x, y, z, sorted index (bits)
    __shared__ float4 s_queue_n[4][128];    //                    their normals
    const int o = blockIdx.y;      // plain object-major order: the XCD-local map made this kernel 6 % SLOWER (A/B on one box, 823 objects: 127.0 vs 119.9 ms)
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const uint32_t base = a.pt_off[o];
    const uint32_t n = a.pt_off[o + 1] - base;
    const uint32_t p = blockIdx.x * 4 + wv;
    if (p >= n || !a.flag[base + p]) return;
    const float4 pp = a.sp4[base + p], pn = a.sn4[base + p];
    const float px = pp.x, py = pp.y, pz = pp.z;
    const float pnx = pn.x, pny = pn.y, pnz = pn.z;
    const GridMeta m = a.meta[o];
    CellRange cr;
    ball_cells(m, px, py, pz, a.radius, cr);
    const uint32_t* cs = a.cell_start + (size_t)o * ISM_GRID_STRIDE;
    const float d_pi = 1.0f / (2.0f * 3.14159265358979323846f);
    uint32_t total = 0, qn_ = 0, qh = 0;
    // The pair features cost several hundred instructions per neighbour (two sqrt, five divisions, atan2f, three double-precision
    // bin formulas), and only about half of the candidates the sweep visits lie inside the ball: as in k_shot, the neighbours that
    // pass are queued (128-entry circular LDS queue, ballot + prefix popcount) and a FULL wave of them does the arithmetic
    // (measured per 128 objects x 16384 points, M ~ 900: k_spfh 20.0 -> see DESIGN.md §5).
    // The 33 bin counts of this point live in wave-uniform registers and grow by popcount(ballot(bin == b)): 33 compares + scalar
    // popcounts per 64 pairs on the wave's own issue slots. Three ds_add_u32 per pair into the 33-word LDS histogram were the
    // bottleneck of this kernel (11 addresses per block: the LDS unit of the CU serialises the same-address lanes, measured ~83
    // cycles per wave instruction, 12 of 23 ms per 128 objects -- the pair arithmetic itself, exact or fast, made no difference).
    int c1[11], c2[11], c3[11];
#pragma unroll
    for (int b = 0; b < 11; ++b) { c1[b] = 0; c2[b] = 0; c3[b] = 0; }
    auto pair_of = [&](bool act, uint32_t t, float qx, float qy, float qz, const float4& qn) {
        int h1 = -1, h2 = -1, h3 = -1;                                  // -1: no deposit (inactive lane, skipped pair)
        if (act && a.dbg != 1) {
            bool have = a.dbg != 2 && pair_bins_fast(px, py, pz, pnx, pny, pnz, qx, qy, qz, qn.x, qn.y, qn.z, h1, h2, h3);
            if (!have && a.dbg != 3) {                                  // ~0.1 % of the pairs (a few per cent of the waves): the reference's arithmetic
                float f1, f2, f3;
                h1 = h2 = h3 = -1;
                if (pair_features(px, py, pz, pnx, pny, pnz, qx, qy, qz, qn.x, qn.y, qn.z, f1, f2, f3)) {
                    // the three bin formulas are evaluated in double as in PCL (float operands, double constants)
                    h1 = clamp_bin((int)floor(11 * (((double)f1 + 3.14159265358979323846) * (double)d_pi)));
                    h2 = clamp_bin((int)floor(11 * (((double)f2 + 1.0) * 0.5)));
                    h3 = clamp_bin((int)floor(11 * (((double)f3 + 1.0) * 0.5)));
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 11; ++b) {
            c1[b] += __popcll(__ballot(h1 == b)); c2[b] += __popcll(__ballot(h2 == b)); c3[b] += __popcll(__ballot(h3 == b));
        }
    };
    float4* sq = s_queue[wv]; float4* sqn = s_queue_n[wv];
    struct PN { float4 p, n; };
    // flattened ball traversal (common.h): the candidate rows laid end to end, every lane busy. The sweep loads position AND normal of
    // every candidate one block ahead of their use (the normal of a candidate that fails the radius test is wasted bandwidth, but a
    // dependent 16-byte gather per queued neighbour, issued when it is needed, was latency the wave could not hide)
    ball_for_each(m, cs, cr, px, py, pz, a.radius, lane, s_rows[wv],
                  [&](uint32_t t, bool) { PN r; r.p = a.sp4[base + t]; r.n = a.sn4[base + t]; return r; },
                  [&](const PN& rec, uint32_t t, bool v) {
        const float4 qq = rec.p;
        bool pass = false;
        if (v) pass = sqdist3(qq.x, qq.y, qq.z, px, py, pz) < a.r2;
        const unsigned long long mask = __ballot(pass);
        total += __popcll(mask);
        const bool enq = pass && t != p;                                     // the point itself is a neighbour (it counts) but no pair
        const unsigned long long emask = __ballot(enq);
        if (enq) {
            const uint32_t pos = (qh + qn_ + __popcll(emask & ((1ull << lane) - 1ull))) & 127u;
            sq[pos] = make_float4(qq.x, qq.y, qq.z, __uint_as_float(t)); sqn[pos] = rec.n;
        }
        qn_ += __popcll(emask);
        if (qn_ >= 64) {
            const uint32_t at = (qh + lane) & 127u;                          // LDS traffic of one wave is ordered: no barrier needed
            const float4 e = sq[at];
            pair_of(true, __float_as_uint(e.w), e.x, e.y, e.z, sqn[at]);
            qh = (qh + 64) & 127u; qn_ -= 64;
        }
    });
    if (qn_ > 0) {
        const bool act = (uint32_t)lane < qn_;
        const uint32_t at = (qh + lane) & 127u;
        const float4 e = sq[at];
        pair_of(act, act ? __float_as_uint(e.w) : 0u, e.x, e.y, e.z, sqn[at]);
    }
    const float hist_incr = 100.0f / (float)(total - 1u);
    int mine = 0;                                                       // lane b < 33 takes bin b
#pragma unroll
    for (int b = 0; b < 11; ++b) { mine = lane == b ? c1[b] : mine; mine = lane == 11 + b ? c2[b] : mine; mine = lane == 22 + b ? c3[b] : mine; }
    // a bin that received nothing stays 0 even when hist_incr = 100/0 (point alone in its ball), as in the reference's += loop
    if (lane < 33) a.spfh[(size_t)(base + p) * 33 + lane] = mine ? (float)mine * hist_incr : 0.f;
}

__global__ __launch_bounds__(256) void k_fpfh_sum(FpfhArgs a) {
    __shared__ uint32_t s_qi[4][64];
    __shared__ float s_qw[4][64];
    int o, bx;
    if (!xcd_object_block(a.nbx_kp, a.n_obj, o, bx)) return;
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const uint32_t k = a.kp_off[o] + bx * 4 + wv;
    if (k >= a.kp_off[o + 1]) return;
    float* out = a.desc + (size_t)k * 33;
    const float cx = a.kx[k], cy = a.ky[k], cz = a.kz[k];
    const GridMeta m = a.meta[o];
    CellRange cr;
    if (!(isfinite(cx) && isfinite(cy) && isfinite(cz)) || !ball_cells(m, cx, cy, cz, a.radius, cr)) {
        if (lane < 33) out[lane] = __builtin_nanf("");
        return;
    }
    const uint32_t* cs = a.cell_start + (size_t)o * ISM_GRID_STRIDE;
    const uint32_t base = a.pt_off[o];
    float acc = 0.f;          // lane b < 33 owns bin b
    uint32_t total = 0;
    for (int gz = cr.lo[2]; gz <= cr.hi[2]; ++gz)
        for (int gy = cr.lo[1]; gy <= cr.hi[1]; ++gy) {
            int xl, xh;
            if (!row_cells(m, cr, gy, gz, cx, cy, cz, a.radius, xl, xh)) continue;
            const int rb = (gz * m.dim[1] + gy) * m.dim[0];
            const uint32_t s = cs[rb + xl], e = cs[rb + xh + 1];
            for (uint32_t t0 = s; t0 < e; t0 += 64) {
                const uint32_t i = t0 + lane;
                bool pass = false; float d2 = 0.f;
                if (i < e) { const float4 q = a.sp4[base + i]; d2 = sqdist3(q.x, q.y, q.z, cx, cy, cz); pass = d2 < a.r2; }
                const unsigned long long mask = __ballot(pass);
                total += __popcll(mask);
                const bool use = pass && d2 != 0.f;                      // "minus the query point itself"
                const unsigned long long umask = __ballot(use);
                if (use) {
                    const uint32_t pos = __popcll(umask & ((1ull << lane) - 1ull));
                    s_qi[wv][pos] = base + i; s_qw[wv][pos] = 1.0f / d2;
                }
                __builtin_amdgcn_wave_barrier();
                const int cnt = __popcll(umask);
                if (lane < 33) {
                    // eight independent 132-byte row gathers in flight, then the additions in the neighbours' order (unchanged sum)
                    int j = 0;
                    for (; j + 8 <= cnt; j += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = a.spfh[(size_t)s_qi[wv][j + u] * 33 + lane];
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc += v[u] * s_qw[wv][j + u];
                    }
                    for (; j < cnt; ++j) acc += a.spfh[(size_t)s_qi[wv][j] * 33 + lane] * s_qw[wv][j];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    if (total == 0) {                 // searchForNeighbors == 0 -> NaN histogram
        if (lane < 33) out[lane] = __builtin_nanf("");
        return;
    }
    // per 11-bin block: scale to sum 100 (sum over the block's lanes)
    float sum = 0.f;
    const int blk = lane / 11;
    for (int j = 0; j < 11; ++j) {
        const float v = __shfl(acc, blk * 11 + j, 64);
        sum += v;
    }
    if (sum != 0.f) sum = 100.0f / sum;
    if (lane < 33) out[lane] = acc * sum;
}

}  // namespace

extern "C" int ismhip_fpfh33(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                             const float* kpx, const float* kpy, const float* kpz,
                             float radius, float* desc_out, uint32_t* neighbour_count_out) {
    if (!ctx || !cloud || !kp_offsets_h || !kpx || !kpy || !kpz || !desc_out || !(radius > 0.f))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, "fpfh33: bad argument");
    const int n_obj = cloud->n_obj;
    uint32_t maxk = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (kp_offsets_h[o + 1] < kp_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, "fpfh33: offsets not monotone");
        maxk = std::max(maxk, kp_offsets_h[o + 1] - kp_offsets_h[o]);
    }
    if (maxk == 0) return ISMHIP_OK;
    uint32_t* ko = ism_upload_offsets(ctx, SCR_KP_OFF, kp_offsets_h, n_obj + 1);
    if (!ko) return ISMHIP_ERR_HIP;
    const size_t np = cloud->n_pts ? cloud->n_pts : 1;
    uint8_t* flag = (uint8_t*)ism_scratch(ctx, SCR_FPFH_FLAG, np);
    float* spfh = (float*)ism_scratch(ctx, SCR_FPFH_SPFH, np * 33 * sizeof(float));
    if (!flag || !spfh) return ISMHIP_ERR_NOMEM;
    FpfhArgs a;
    a.pt_off = cloud->pt_off; a.meta = cloud->meta; a.cell_start = cloud->cell_start;
    a.sp4 = cloud->sp4; a.sn4 = cloud->sn4;
    a.kp_off = ko; a.kx = kpx; a.ky = kpy; a.kz = kpz;
    a.radius = radius; a.r2 = (float)((double)radius * (double)radius);
    a.flag = flag; a.spfh = spfh; a.desc = desc_out; a.count = neighbour_count_out; a.max_pts = cloud->max_pts;
    { const char* e = getenv("ISMHIP_FPFH_DBG"); a.dbg = e ? atoi(e) : 0; }
    TimerScope ts(ctx, "fpfh33");
    ISM_HIP(ctx, hipMemsetAsync(flag, 0, np, ctx->stream));
    // measured round 3 (configs[4], 823 objects per launch, plain (blocks, objects) grid): k_fpfh_sum pulled 74.8 GB per launch from
    // beyond the L2s for 1.8 GB of SPFH rows -- every object's rows were wanted by all eight XCDs at once. A/B on one box:
    // k_fpfh_sum 23.4 -> 21.0 ms with the XCD-local map
    a.n_obj = ctx->xcd_map ? n_obj : 0; a.nbx_kp = (int)((maxk + 3) / 4);
    const unsigned g_kp = ctx->xcd_map ? xcd_object_grid((unsigned)a.nbx_kp, n_obj) : (unsigned)a.nbx_kp * (unsigned)n_obj;
    hipLaunchKernelGGL(k_fpfh_mark, dim3(g_kp), dim3(256), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_fpfh_mark");
    hipLaunchKernelGGL(k_spfh, dim3((cloud->max_pts + 3) / 4, n_obj), dim3(256), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_spfh");
    hipLaunchKernelGGL(k_fpfh_sum, dim3(g_kp), dim3(256), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, "k_fpfh_sum");
    return ISMHIP_OK;
}
