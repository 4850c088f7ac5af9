// shot.hip — SHOT-352 and CSHOT-1344 descriptors, one 64-lane wavefront per keypoint.
// Reference seams: FeaturesSHOT::iComputeDescriptors (features/features_shot.cpp:28-81) ->
// pcl::SHOTEstimationOMP<..., SHOT352>; FeaturesCSHOT::iComputeDescriptors (features/features_cshot.cpp:28-103)
// -> pcl::SHOTColorEstimationOMP<..., SHOT1344>. Arithmetic restated from PCL 1.10 computePointSHOT /
// interpolateSingleChannel / interpolateDoubleChannel / normalizeHistogram.
//
// Roofline: HBM-bound gather. Algorithmic bytes per keypoint = M_k * 24 + 12 + 36 + 352*4 (SHOT) or
// M_k * 28 + 12 + 36 + 4 + 1344*4 (CSHOT), M_k = radius neighbours.
//
// Structure per wave: stream the candidate x-runs of the query ball (coalesced SoA loads of the cell-sorted
// cloud); lanes whose point is inside the ball are COMPACTED with a ballot + prefix popcount into a 128-entry
// LDS queue; whenever 64 are queued, all 64 lanes run the per-neighbour math on a full wave (no divergence on
// the radius test) and deposit into a per-wave LDS histogram (64-bit fixed point, ds_add_u64 — see ShotSmem). The wave then
// L2-normalises and writes the row with coalesced stores.
#include "common.h"

uint32_t* ism_upload_offsets(ismhip_ctx* ctx, int slot, const uint32_t* off_h, int n);

namespace {

#define PST_RAD_45f   0.78539816339744830962f
#define PST_RAD_90f   1.57079632679489661923f
#define PST_RAD_135f  2.35619449019234492885f
#define PST_RAD_PI_7_8f 2.7488935718910690836f

struct ShotArgs {
    const uint32_t* pt_off; const GridMeta* meta; const uint32_t* cell_start;
    const float4 *sp4, *sn4, *slab4;
    const uint32_t* kp_off; const float *kx, *ky, *kz; const uint32_t* kp_rgba;
    const float* lrf; float radius, r2, r12sq_f;
    const float *lut_srgb, *lut_sxyz;
    float* desc; uint32_t* count;
    int n_obj, nbx;
};

// The per-wave LDS histogram is kept in 64-bit FIXED POINT (2^-28 units) and updated with ds_add_u64.
// Measured on gfx950 (tools/lds_atomic_bench.hip): ds_add_f32 costs ~195 CU-cycles per wave-instruction whatever the
// addresses, ds_add_u64 17-24, ds_add_u32 15 — with float atomics the kernel spent 64 % of its wave cycles in
// SQ_WAIT_INST_LDS. Every deposit is a non-negative interpolation weight < 8, so round(v * 2^28) fits 32 bits and a bin
// (<= 4 * 2^14 neighbours) needs 46 bits. Integer adds are associative: the histogram no longer depends on the order in
// which neighbours arrive (bitwise reproducible), and its error (<= 2^-29 per deposit) is far below the float
// accumulation error of the reference itself.
#define SHOT_FIX_SCALE 268435456.0f          /* 2^28 */
#define SHOT_FIX_INV   3.7252902984619140625e-09 /* 2^-28 */
typedef unsigned long long shot_bin_t;
// The interpolation weights are CONTINUOUS in distance / inclination / azimuth (all hard bin decisions are taken on the
// signs and squares above them), so these three only need ~1e-7 absolute accuracy -- far inside the 1e-4 parity tolerance --
// and not libm's last ulp: the IEEE division / sqrt expansions and OCML's acosf / atan2f were a quarter of the kernel's VALU.
// acos: Abramowitz & Stegun 4.4.46 (|err| <= 2e-8 before rounding); atan: A&S 4.4.49 (|err| <= 2e-8).
__device__ __forceinline__ float shot_acos(float x) {
    const float a = fabsf(x);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, a, 0.0066700901f); p = __builtin_fmaf(p, a, -0.0170881256f); p = __builtin_fmaf(p, a, 0.0308918810f);
    p = __builtin_fmaf(p, a, -0.0501743046f); p = __builtin_fmaf(p, a, 0.0889789874f); p = __builtin_fmaf(p, a, -0.2145988016f);
    p = __builtin_fmaf(p, a, 1.5707963050f);
    const float r = __builtin_amdgcn_sqrtf(fmaxf(1.0f - a, 0.f)) * p;
    return x < 0.f ? 3.14159265358979323846f - r : r;
}
__device__ __forceinline__ float shot_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);          // mx > 0: the caller excludes x == y == 0
    const float z = a * a;
    float p = 0.0028662257f;
    p = __builtin_fmaf(p, z, -0.0161657367f); p = __builtin_fmaf(p, z, 0.0429096138f); p = __builtin_fmaf(p, z, -0.0752896400f);
    p = __builtin_fmaf(p, z, 0.1065626393f); p = __builtin_fmaf(p, z, -0.1420889944f); p = __builtin_fmaf(p, z, 0.1999355085f);
    p = __builtin_fmaf(p, z, -0.3333314528f); p = __builtin_fmaf(p, z, 1.0f);
    float r = a * p;
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (x < 0.f) r = 3.14159265358979323846f - r;
    return y < 0.f ? -r : r;
}

// step = floor(m*x + c) and off = float((m*x + c0) - step) exactly as the reference's double arithmetic yields them, for a float x
// whose product m*x (m = 5 or 30) is exact in double -- true wherever m*x + c can reach an integer. One fma rounds the exact value
// once; floor of the rounded value differs from the true floor only if the rounding went UP onto an integer, which the sign of a
// second fma (the exact residual, rounded) reveals. c0 = c - 0.5: the offset is measured from the bin centre.
__device__ __forceinline__ void shot_hard_bin(float m, float x, float c, float c0, int& step, float& off) {
    const float t = __builtin_fmaf(m, x, c);
    float k = floorf(t);
    if (t == k && __builtin_fmaf(m, x, c - k) < 0.f) k -= 1.f;
    step = (int)k;
    off = __builtin_fmaf(m, x, c0 - k);
}

template <bool COLOR>
struct ShotSmem {
    static constexpr int D = COLOR ? 1344 : 352;
    shot_bin_t hist[4][D];
    float4 qd[4][128];       // dx, dy, dz, d2 of queued neighbours
    uint32_t qi[4][128];     // sorted index of queued neighbours
    WaveRows rows[4];
};
__device__ __forceinline__ void shot_dep(shot_bin_t* hist, int bin, float v) {
    atomicAdd(&hist[bin], (shot_bin_t)__float2uint_rn(v * SHOT_FIX_SCALE));
}

// Per-neighbour SHOT update. All 64 lanes call it; 'act' marks lanes that hold a neighbour.
template <bool COLOR, int VAR>
__device__ __forceinline__ void shot_neighbour(const ShotArgs& a, shot_bin_t* hist, bool act, uint32_t gi,
                                               float dx, float dy, float dz, float d2,
                                               const float fx[3], const float fy[3], const float fz[3],
                                               float r12, float r14, float r34, float inv_r12, float r12sq_f,
                                               float LRef, float aRef, float bRef) {
    if (!act) return;
    const float4 nrm = a.sn4[gi];
    const float nxv = nrm.x, nyv = nrm.y, nzv = nrm.z;
    if (!(isfinite(nxv) && isfinite(nyv) && isfinite(nzv))) return;              // createBinDistanceShape: NaN normal -> skipped
    float cosd = (nxv * fz[0] + nyv * fz[1]) + nzv * fz[2];
    cosd = fminf(1.0f, fmaxf(-1.0f, cosd));
    // PCL's interpolation is only partly soft: the whole accumulated weight lands in the HARD-assigned (sector, step) bin, so
    // every hard decision must be taken exactly as the reference takes it. The reference computes the cosine bin in double from
    // the float cosine (createBinDistanceShape) and compares the radial shell on squares in double; both are reproduced below
    // without FP64 instructions (shot_hard_bin, r12sq_f).
    const float dist = __builtin_amdgcn_sqrtf(d2);
    if (dist < 1e-15f) return;                                                    // areEquals(distance, 0)
    float xl = (dx * fx[0] + dy * fx[1]) + dz * fx[2];
    float yl = (dx * fy[0] + dy * fy[1]) + dz * fy[2];
    float zl = (dx * fz[0] + dy * fz[1]) + dz * fz[2];
    if (fabsf(yl) < 1e-30f) yl = 0.f;
    if (fabsf(xl) < 1e-30f) xl = 0.f;
    if (fabsf(zl) < 1e-30f) zl = 0.f;
    const int bit4 = ((yl > 0.f) || ((yl == 0.f) && (xl < 0.f))) ? 1 : 0;
    const int bit3 = ((xl > 0.f) || ((xl == 0.f) && (yl > 0.f))) ? !bit4 : bit4;
    int di = ((bit4 << 3) + (bit3 << 2)) << 1;
    const bool same_sign = (xl > 0.f && yl > 0.f) || (xl < 0.f && yl < 0.f);      // x*y > 0 without float underflow
    if (same_sign || xl == 0.f) di += (fabsf(xl) >= fabsf(yl)) ? 0 : 4;
    else di += (fabsf(xl) > fabsf(yl)) ? 4 : 0;
    di += zl > 0.f ? 1 : 0;
    const bool outer = d2 > r12sq_f;                                              // distance > radius1_2: (double)d2 > r^2/4 in double == d2 > largest float <= r^2/4
    di += outer ? 2 : 0;

    int step; float bd;
    shot_hard_bin(5.f, cosd, 5.5f, 5.f, step, bd);                                // bin = floor(((1 + cos) * 10) / 2 + 0.5), bd = offset from its centre
    const int vol = di * 11;
    float w_shape = 1.f - fabsf(bd);
    // Everything below is branch-free: each of the four interpolation directions yields ONE (neighbouring bin, weight) pair
    // chosen by selects -- written as the reference's if/else ladders it compiled into eight divergent deposit sites with
    // partial exec masks and their branch overhead. The float operations (and so the results) are the same: in the "deposit"
    // case of every ladder the reference adds 1 -/+ t with t of the sign that makes it 1 - |t|, and deposits |t|.
    {
        int t = bd > 0.f ? step + 1 : step + 9;                                   // (step + 1) % 10 | (step - 1 + 10) % 10, step in 0..10
        t = t >= 10 ? t - 10 : t;
        shot_dep(hist, vol + t, fabsf(bd));
    }
    int step_c = 0, vol_c = 0; float w_col = 0.f;
    if (COLOR) {
        const float4 lab = a.slab4[gi];
        const float L = lab.x, A = lab.y, B = lab.z;
        float cd = (fabsf(LRef - L) + ((fabsf(aRef - A) + fabsf(bRef - B)) * 0.5f)) / 3.0f;   // feeds a hard bin: exact division
        cd = fminf(1.0f, fmaxf(0.0f, cd));
        float bc;
        shot_hard_bin(30.f, cd, 0.5f, 0.f, step_c, bc);                           // colorDistance (float) * nr_color_bins_, taken in double by the reference
        vol_c = 352 + di * 31;
        w_col = 1.f - fabsf(bc);
        int t = bc > 0.f ? step_c + 1 : step_c + 29;                              // % 30
        t = t >= 30 ? t - 30 : t;
        shot_dep(hist, vol_c + t, fabsf(bc));
    }
    float winc = 0.f;
    // radial: outer shell interpolates towards the inner one below 3r/4, inner shell towards the outer one above r/4
    const float xr = outer ? (dist - r34) * inv_r12 : -((dist - r14) * inv_r12);
    const float wr_ = xr > 0.f ? 0.f : fabsf(xr);
    winc += 1.f - fabsf(xr);
    const int sec_r = outer ? di - 2 : di + 2;
    // elevation
    float ic = zl * __builtin_amdgcn_rcpf(dist);
    ic = fminf(1.0f, fmaxf(-1.0f, ic));
    const float inc = shot_acos(ic);
    const bool lower = !(zl > 0.f);   // inclination > 90 deg, or exactly 90 deg with z <= 0: the same test that picked the sector's elevation bit
    const float ide = (inc - (lower ? PST_RAD_135f : PST_RAD_45f)) * (1.0f / PST_RAD_90f);
    const float xe = lower ? ide : -ide;
    const float we = xe > 0.f ? 0.f : fabsf(xe);
    winc += 1.f - fabsf(xe);
    const int sec_e = lower ? di + 1 : di - 1;
    // azimuth
    float wa = 0.f; int sec_a = di;
    if (yl != 0.f || xl != 0.f) {
        const float az = shot_atan2(yl, xl);
        const int sel = di >> 2;
        float ad = (az - (-PST_RAD_PI_7_8f + PST_RAD_45f * (float)sel)) * (1.0f / PST_RAD_45f);
        ad = fmaxf(-0.5f, fminf(ad, 0.5f));
        wa = fabsf(ad);
        winc += 1.f - wa;
        sec_a = (ad > 0.f ? di + 4 : di - 4) & 31;
    }
    if (VAR & 1) {
        // unconditional deposits (measured SLOWER, 4.44 vs 4.06 ms: more lanes in every atomic = more same-address serialisation)
        shot_dep(hist, sec_r * 11 + step, wr_); shot_dep(hist, sec_e * 11 + step, we); shot_dep(hist, sec_a * 11 + step, wa);
        if (COLOR) { shot_dep(hist, 352 + sec_r * 31 + step_c, wr_); shot_dep(hist, 352 + sec_e * 31 + step_c, we); shot_dep(hist, 352 + sec_a * 31 + step_c, wa); }
    } else {
    if (wr_ != 0.f) { shot_dep(hist, sec_r * 11 + step, wr_); if (COLOR) shot_dep(hist, 352 + sec_r * 31 + step_c, wr_); }
    if (we != 0.f) { shot_dep(hist, sec_e * 11 + step, we); if (COLOR) shot_dep(hist, 352 + sec_e * 31 + step_c, we); }
    if (wa != 0.f) { shot_dep(hist, sec_a * 11 + step, wa); if (COLOR) shot_dep(hist, 352 + sec_a * 31 + step_c, wa); }
    }
    shot_dep(hist, vol + step, w_shape + winc);
    if (COLOR) shot_dep(hist, vol_c + step_c, w_col + winc);
}

__device__ __forceinline__ void rgb2lab_norm(const float* lut_srgb, const float* lut_sxyz, uint32_t c4, float& L, float& A, float& B) {
    const float fr = lut_srgb[(c4 >> 16) & 0xff], fg = lut_srgb[(c4 >> 8) & 0xff], fb = lut_srgb[c4 & 0xff];
    const float X = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
    const float Y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
    const float Z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
    float vx = X / 0.95047f, vy = Y, vz = Z / 1.08883f;
    int ix = (int)(vx * 4000), iy = (int)(vy * 4000), iz = (int)(vz * 4000);
    ix = ix < 0 ? 0 : (ix > 3999 ? 3999 : ix); iy = iy < 0 ? 0 : (iy > 3999 ? 3999 : iy); iz = iz < 0 ? 0 : (iz > 3999 ? 3999 : iz);
    vx = lut_sxyz[ix]; vy = lut_sxyz[iy]; vz = lut_sxyz[iz];
    L = 116.0f * vy - 16.0f; if (L > 100) L = 100.0f;
    A = 500.0f * (vx - vy); if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
    B = 200.0f * (vy - vz); if (B > 120) B = 120.0f; else if (B < -120) B = -120.0f;
    L /= 100.0f; A /= 120.0f; B /= 120.0f;
}

// 6 workgroups (24 waves) per CU: the LDS budget (24.5 KB per workgroup) allows it, so the register allocation must too (<= 80)
template <bool COLOR, int VAR>
__global__ __launch_bounds__(256, COLOR ? 2 : 6) void k_shot(ShotArgs a) {
    constexpr int D = COLOR ? 1344 : 352;
    __shared__ ShotSmem<COLOR> sm;
    int o, bx;
    if (!xcd_object_block(a.nbx, a.n_obj, o, bx)) return;
    const int wv = threadIdx.x >> 6;
    const int lane = lane_id();
    const uint32_t k = a.kp_off[o] + bx * 4 + wv;
    if (k >= a.kp_off[o + 1]) return;          // wave-uniform; no block-level barrier below
    shot_bin_t* hist = sm.hist[wv];
    float* out = a.desc + (size_t)k * D;
    const float cx = a.kx[k], cy = a.ky[k], cz = a.kz[k];
    const float* f = a.lrf + (size_t)k * 9;
    const float fx[3] = {f[0], f[1], f[2]}, fy[3] = {f[3], f[4], f[5]}, fz[3] = {f[6], f[7], f[8]};
    const GridMeta m = a.meta[o];
    CellRange cr;
    const bool ok = isfinite(fx[0]) && isfinite(fy[0]) && isfinite(fz[0]) && isfinite(cx) && isfinite(cy) && isfinite(cz);
    if (!ok || !ball_cells(m, cx, cy, cz, a.radius, cr)) {
        for (int i = lane; i < D; i += 64) out[i] = __builtin_nanf("");
        if (a.count && lane == 0) a.count[k] = 0;
        return;
    }
    for (int i = lane; i < D; i += 64) hist[i] = 0ull;
    float LRef = 0.f, aRef = 0.f, bRef = 0.f;
    if (COLOR) rgb2lab_norm(a.lut_srgb, a.lut_sxyz, a.kp_rgba[k], LRef, aRef, bRef);
    const float r12 = a.radius * 0.5f, r14 = a.radius * 0.25f, r34 = (a.radius * 3.0f) * 0.25f, inv_r12 = 1.0f / r12;
    const float r12sq_f = a.r12sq_f;
    const uint32_t* cs = a.cell_start + (size_t)o * ISM_GRID_STRIDE;
    const uint32_t base = a.pt_off[o];
    uint32_t qn = 0, qh = 0, total = 0;
    // 16 interleaved segments: measured 4.18 (contiguous) -> 3.27 (8 segments) -> 2.96 ms (16 interleaved) per 256 objects; 32 lose to coalescing
    ball_for_each<(VAR & 2) ? 1 : 16, true>(m, cs, cr, cx, cy, cz, a.radius, lane, sm.rows[wv],
                  [&](uint32_t i, bool) { return a.sp4[base + i]; },      // invalid lanes carry index 0 (common.h): no branch, no zero fill
                  [&](const float4& p, uint32_t i, bool v) {
        bool pass = false; float dx = 0, dy = 0, dz = 0, d2 = 0;
        if (v) {
            const float px = p.x, py = p.y, pz = p.z;
            d2 = sqdist3(px, py, pz, cx, cy, cz);
            dx = px - cx; dy = py - cy; dz = pz - cz;
            pass = d2 < a.r2;
        }
        const unsigned long long mask = __ballot(pass);
        if (pass) {
            const uint32_t pos = (qh + qn + __popcll(mask & ((1ull << lane) - 1ull))) & 127u;      // 128-entry circular queue
            sm.qd[wv][pos] = make_float4(dx, dy, dz, d2); sm.qi[wv][pos] = base + i;
        }
        const uint32_t c = __popcll(mask);
        qn += c; total += c;
        if (qn >= 64) {
            // a full wave of neighbours (LDS traffic of one wave is ordered; no barrier needed)
            const uint32_t at = (qh + lane) & 127u;
            const float4 e = sm.qd[wv][at];
            shot_neighbour<COLOR, VAR>(a, hist, true, sm.qi[wv][at], e.x, e.y, e.z, e.w, fx, fy, fz, r12, r14, r34, inv_r12, r12sq_f, LRef, aRef, bRef);
            qh = (qh + 64) & 127u; qn -= 64;
        }
    });
    if (qn > 0) {
        const bool act = (uint32_t)lane < qn;
        const uint32_t at = (qh + lane) & 127u;
        const float4 e = sm.qd[wv][at];
        shot_neighbour<COLOR, VAR>(a, hist, act, act ? sm.qi[wv][at] : 0u, e.x, e.y, e.z, e.w, fx, fy, fz, r12, r14, r34, inv_r12, r12sq_f, LRef, aRef, bRef);
    }
    if (a.count && lane == 0) a.count[k] = total;
    if (total < 5) {                                    // computePointSHOT: fewer than 5 neighbours -> NaN descriptor
        for (int i = lane; i < D; i += 64) out[i] = __builtin_nanf("");
        return;
    }
    // normalizeHistogram: double accumulate of float squares, divide by float(norm)
    double acc = 0.0;
    for (int i = lane; i < D; i += 64) { const float v = (float)((double)hist[i] * SHOT_FIX_INV); acc += (double)(v * v); }
    acc = wave_sum_d(acc);
    const float fn = (float)sqrt(acc);
    for (int i = lane; i < D; i += 64) out[i] = (float)((double)hist[i] * SHOT_FIX_INV) / fn;
}

template <bool COLOR>
int launch_shot(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                const float* kpx, const float* kpy, const float* kpz, const uint32_t* kp_rgba,
                const float* lrf9, float radius, float* desc_out, uint32_t* count_out, const char* name) {
    if (!ctx || !cloud || !kp_offsets_h || !kpx || !kpy || !kpz || !lrf9 || !desc_out || !(radius > 0.f))
        return ism_set_err(ctx, ISMHIP_ERR_INVALID, std::string(name) + ": bad argument");
    if (COLOR && (!cloud->rgba || !kp_rgba)) return ism_set_err(ctx, ISMHIP_ERR_INVALID, std::string(name) + ": colour arrays missing");
    const int n_obj = cloud->n_obj;
    uint32_t maxk = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (kp_offsets_h[o + 1] < kp_offsets_h[o]) return ism_set_err(ctx, ISMHIP_ERR_INVALID, std::string(name) + ": offsets not monotone");
        maxk = std::max(maxk, kp_offsets_h[o + 1] - kp_offsets_h[o]);
    }
    if (maxk == 0) return ISMHIP_OK;
    uint32_t* ko = ism_upload_offsets(ctx, SCR_KP_OFF, kp_offsets_h, n_obj + 1);
    if (!ko) return ISMHIP_ERR_HIP;
    ShotArgs a;
    a.pt_off = cloud->pt_off; a.meta = cloud->meta; a.cell_start = cloud->cell_start;
    a.sp4 = cloud->sp4; a.sn4 = cloud->sn4; a.slab4 = cloud->slab4;
    a.kp_off = ko; a.kx = kpx; a.ky = kpy; a.kz = kpz; a.kp_rgba = kp_rgba; a.lrf = lrf9;
    a.radius = radius; a.r2 = (float)((double)radius * (double)radius);
    {   // largest float <= (radius/2)^2 taken in double: the shell test (double)d2 > r12sq of the reference, as a float compare
        const double t = 0.25 * (double)radius * (double)radius;
        float tf = (float)t;
        if ((double)tf > t) tf = nextafterf(tf, -INFINITY);
        a.r12sq_f = tf;
    }
    a.lut_srgb = ctx->lut_srgb; a.lut_sxyz = ctx->lut_sxyz;
    a.desc = desc_out; a.count = count_out;
    a.n_obj = ctx->xcd_map ? n_obj : 0; a.nbx = (int)((maxk + 3) / 4);
    TimerScope ts(ctx, name);
    const dim3 grid(ctx->xcd_map ? xcd_object_grid((unsigned)a.nbx, n_obj) : (unsigned)a.nbx * (unsigned)n_obj);
    if (ctx->shot_var & 2) hipLaunchKernelGGL((k_shot<COLOR, 2>), grid, dim3(256), 0, ctx->stream, a);      // contiguous sweep (A/B runs)
    else hipLaunchKernelGGL((k_shot<COLOR, 0>), grid, dim3(256), 0, ctx->stream, a);
    ISM_CHECK_LAUNCH(ctx, name);
    return ISMHIP_OK;
}

}  // namespace

extern "C" {

int ismhip_shot352(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                   const float* kpx, const float* kpy, const float* kpz,
                   const float* lrf9, float radius, float* desc_out, uint32_t* neighbour_count_out) {
    return launch_shot<false>(ctx, cloud, kp_offsets_h, kpx, kpy, kpz, nullptr, lrf9, radius, desc_out, neighbour_count_out, "shot352");
}

int ismhip_cshot1344(ismhip_ctx* ctx, const ismhip_cloud* cloud, const uint32_t* kp_offsets_h,
                     const float* kpx, const float* kpy, const float* kpz, const uint32_t* kp_rgba,
                     const float* lrf9, float radius, float* desc_out, uint32_t* neighbour_count_out) {
    return launch_shot<true>(ctx, cloud, kp_offsets_h, kpx, kpy, kpz, kp_rgba, lrf9, radius, desc_out, neighbour_count_out, "cshot1344");
}

}  // extern "C"
