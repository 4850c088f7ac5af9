/*
 * ism_oracle.cpp — CPU restatement (oracle) of the implicit_shape_model recognition hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ism_oracle.h). PARITY UNPINNED by the reference (no reference tests,
 * PCL/FLANN/Eigen absent => reference unbuildable here); pinned by hand-derived KATs in tests/golden/.
 *
 * Citations "ref:" are relative to /root/reference/src/implicit_shape_model/. "PCL:"/"FLANN:" mark
 * arithmetic restated from the published algorithms of PCL 1.10.0 / FLANN 1.9.1,
 * which are not vendored by the reference.
 *
 * Built with -ffp-contract=off so that float/double operation order is the written one.
 */
#include "ism_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const float kNaN = std::numeric_limits<float>::quiet_NaN();
int g_threads = 0;

inline bool finite3(float a, float b, float c) { return std::isfinite(a) && std::isfinite(b) && std::isfinite(c); }

/* FLANN: L2_Simple<float> as used by pcl::KdTreeFLANN — sequential float accumulate over x,y,z. */
inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float r = 0.f, d;
    d = ax - bx; r += d * d;
    d = ay - by; r += d * d;
    d = az - bz; r += d * d;
    return r;
}
/* PCL: KdTreeFLANN::radiusSearch passes static_cast<float>(radius*radius) with double radius. */
inline float radius_sq(float radius) { return static_cast<float>(static_cast<double>(radius) * static_cast<double>(radius)); }

/* Uniform grid over one object's points; exact fixed-radius search, results sorted by (d^2, index)
 * (PCL: pcl::search::KdTree::radiusSearch with sorted results; FLANN result sets admit dist < radius^2). */
struct ObjGrid {
    int n = 0;
    const float *x = nullptr, *y = nullptr, *z = nullptr;
    float minv[3] = {0, 0, 0};
    float cell = 1.f;
    int dim[3] = {1, 1, 1};
    std::vector<int> start;   // [ncell+1]
    std::vector<int> order;   // point indices sorted by cell

    void build(int n_, const float* x_, const float* y_, const float* z_, float cell_) {
        n = n_; x = x_; y = y_; z = z_;
        float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        minv[0] = minv[1] = minv[2] = FLT_MAX;
        for (int i = 0; i < n; ++i) {
            if (!finite3(x[i], y[i], z[i])) continue;
            minv[0] = std::min(minv[0], x[i]); mx[0] = std::max(mx[0], x[i]);
            minv[1] = std::min(minv[1], y[i]); mx[1] = std::max(mx[1], y[i]);
            minv[2] = std::min(minv[2], z[i]); mx[2] = std::max(mx[2], z[i]);
        }
        if (n == 0 || minv[0] > mx[0]) { minv[0] = minv[1] = minv[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
        cell = cell_ > 0 ? cell_ : 1.f;
        // keep the grid bounded: at most 128 cells per axis
        for (;;) {
            bool ok = true;
            for (int a = 0; a < 3; ++a) {
                double d = (static_cast<double>(mx[a]) - minv[a]) / cell;
                if (d >= 127.0) ok = false;
            }
            if (ok) break;
            cell *= 2.f;
        }
        for (int a = 0; a < 3; ++a) dim[a] = static_cast<int>(std::floor((mx[a] - minv[a]) / cell)) + 1;
        const int ncell = dim[0] * dim[1] * dim[2];
        start.assign(ncell + 1, 0);
        std::vector<int> cid(n, -1);
        for (int i = 0; i < n; ++i) {
            if (!finite3(x[i], y[i], z[i])) continue;
            int c = cell_of(x[i], y[i], z[i]);
            cid[i] = c; start[c + 1]++;
        }
        for (int c = 0; c < ncell; ++c) start[c + 1] += start[c];
        order.assign(start[ncell], 0);
        std::vector<int> cur(start.begin(), start.end() - 1);
        for (int i = 0; i < n; ++i) if (cid[i] >= 0) order[cur[cid[i]]++] = i;
    }
    inline int clampi(int v, int lo, int hi) const { return v < lo ? lo : (v > hi ? hi : v); }
    inline int cell_of(float px, float py, float pz) const {
        int cx = clampi(static_cast<int>(std::floor((px - minv[0]) / cell)), 0, dim[0] - 1);
        int cy = clampi(static_cast<int>(std::floor((py - minv[1]) / cell)), 0, dim[1] - 1);
        int cz = clampi(static_cast<int>(std::floor((pz - minv[2]) / cell)), 0, dim[2] - 1);
        return (cz * dim[1] + cy) * dim[0] + cx;
    }
    /* neighbours with d2 < r2, sorted ascending by (d2, index) */
    void radius(float qx, float qy, float qz, float r, std::vector<std::pair<float, int>>& out) const {
        out.clear();
        if (!finite3(qx, qy, qz) || n == 0) return;
        const float r2 = radius_sq(r);
        int lo[3], hi[3];
        const float q[3] = {qx, qy, qz};
        for (int a = 0; a < 3; ++a) {
            // conservative cell range (one extra cell margin absorbs float rounding of the bounds)
            double l = (static_cast<double>(q[a]) - r - minv[a]) / cell;
            double h = (static_cast<double>(q[a]) + r - minv[a]) / cell;
            lo[a] = static_cast<int>(std::floor(l)) - 1;
            hi[a] = static_cast<int>(std::floor(h)) + 1;
            if (hi[a] < 0 || lo[a] > dim[a] - 1) return;
            lo[a] = std::max(lo[a], 0); hi[a] = std::min(hi[a], dim[a] - 1);
        }
        for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy) {
                const int rb = (cz * dim[1] + cy) * dim[0];
                const int s = start[rb + lo[0]], e = start[rb + hi[0] + 1];
                for (int t = s; t < e; ++t) {
                    const int i = order[t];
                    const float d2 = sqdist3(x[i], y[i], z[i], qx, qy, qz);
                    if (d2 < r2) out.emplace_back(d2, i);
                }
            }
        std::sort(out.begin(), out.end());
    }
};

/* ---- symmetric 3x3 eigen-decomposition (double), cyclic Jacobi. Stands in for
 * Eigen::SelfAdjointEigenSolver<Matrix3d> (ref: third_party/pcl_shot_na_lrf/shot_na_lrf.hpp:95).
 * Eigenvalues ascending in w[0..2]; eigenvectors are the columns V[:,k] = (v[0][k], v[1][k], v[2][k]).
 * The identical routine runs on the device (csrc/eigen3.h), so sign/ordering conventions agree. */
void eigen_sym3(const double A[3][3], double w[3], double V[3][3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = A[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        const double diag = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0);
                const double s = t * c;
                const double apq = a[p][q];
                a[p][p] -= t * apq;
                a[q][q] += t * apq;
                a[p][q] = a[q][p] = 0.0;
                const int r = 3 - p - q;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double ev[3] = {a[0][0], a[1][1], a[2][2]};
    // stable ascending sort of three
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j)
        if (ev[idx[j]] > ev[idx[j + 1]]) std::swap(idx[j], idx[j + 1]);
    double Vs[3][3];
    for (int k = 0; k < 3; ++k) { w[k] = ev[idx[k]]; for (int i = 0; i < 3; ++i) Vs[i][k] = V[i][idx[k]]; }
    std::memcpy(V, Vs, sizeof(Vs));
}

/* ---- SHOT local reference frame -----------------------------------------------------------
 * PCL: SHOTLocalReferenceFrameEstimation::getLocalRF; in-repo twin ref: third_party/pcl_shot_na_lrf/
 * shot_na_lrf.hpp:48-178 (z-sign rule taken from upstream: vij . v3 >= 0). */
void lrf_one(const ObjGrid& g, float cx, float cy, float cz, float radius,
             std::vector<std::pair<float, int>>& nb, std::vector<double>& vij, float* out9) {
    g.radius(cx, cy, cz, radius, nb);
    vij.resize(nb.size() * 3);
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double sum = 0.0;
    int valid = 0;
    for (size_t i = 0; i < nb.size(); ++i) {
        const int p = nb[i].second;
        if (g.x[p] == cx && g.y[p] == cy && g.z[p] == cz) continue;            // :63-64
        const double vx = static_cast<double>(g.x[p] - cx);                     // float diff, then cast (:67)
        const double vy = static_cast<double>(g.y[p] - cy);
        const double vz = static_cast<double>(g.z[p] - cz);
        vij[valid * 3 + 0] = vx; vij[valid * 3 + 1] = vy; vij[valid * 3 + 2] = vz;
        const double distance = static_cast<double>(radius) - std::sqrt(static_cast<double>(nb[i].first)); // :70
        cov[0][0] += distance * (vx * vx); cov[0][1] += distance * (vx * vy); cov[0][2] += distance * (vx * vz);
        cov[1][0] += distance * (vy * vx); cov[1][1] += distance * (vy * vy); cov[1][2] += distance * (vy * vz);
        cov[2][0] += distance * (vz * vx); cov[2][1] += distance * (vz * vy); cov[2][2] += distance * (vz * vz);
        sum += distance;
        valid++;
    }
    if (valid < 5) { for (int i = 0; i < 9; ++i) out9[i] = kNaN; return; }    // :80-86
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov[i][j] /= sum;  // :88
    double w[3], V[3][3];
    eigen_sym3(cov, w, V);
    if (!std::isfinite(w[0]) || !std::isfinite(w[1]) || !std::isfinite(w[2])) {
        for (int i = 0; i < 9; ++i) out9[i] = kNaN;
        return;
    }
    double v1[3] = {V[0][2], V[1][2], V[2][2]};   // largest  -> x (:108)
    double v3[3] = {V[0][0], V[1][0], V[2][0]};   // smallest -> z (:109)
    int plusNormal = 0, plusTangent = 0;
    for (int ne = 0; ne < valid; ++ne) {
        const double* v = &vij[ne * 3];
        double dp = v[0] * v1[0] + v[1] * v1[1] + v[2] * v1[2];
        if (dp >= 0) plusTangent++;
        dp = v[0] * v3[0] + v[1] * v3[1] + v[2] * v3[2];
        if (dp >= 0) plusNormal++;
    }
    auto disambiguate = [&](int plus, double* axis) {
        plus = 2 * plus - valid;                                            // :137
        if (plus == 0) {
            const int points = 5;
            const int median = valid / 2;
            for (int i = -points / 2; i <= points / 2; ++i) {
                const double* v = &vij[(median - i) * 3];
                if (v[0] * axis[0] + v[1] * axis[1] + v[2] * axis[2] > 0) plus++;
            }
            if (plus < points / 2 + 1) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; }
        } else if (plus < 0) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; }
    };
    disambiguate(plusTangent, v1);
    disambiguate(plusNormal, v3);
    const float xf[3] = {static_cast<float>(v1[0]), static_cast<float>(v1[1]), static_cast<float>(v1[2])};
    const float zf[3] = {static_cast<float>(v3[0]), static_cast<float>(v3[1]), static_cast<float>(v3[2])};
    // rf.row(1) = rf.row(2).cross(rf.row(0)) in float (:170)
    const float yf[3] = {zf[1] * xf[2] - zf[2] * xf[1], zf[2] * xf[0] - zf[0] * xf[2], zf[0] * xf[1] - zf[1] * xf[0]};
    out9[0] = xf[0]; out9[1] = xf[1]; out9[2] = xf[2];
    out9[3] = yf[0]; out9[4] = yf[1]; out9[5] = yf[2];
    out9[6] = zf[0]; out9[7] = zf[1]; out9[8] = zf[2];
}

/* ---- RGB -> CIELab via LUTs. ref: features/features_cshot.cpp:52-70 (LUT init), features_short_cshot.cpp:651-687 */
struct LabLut {
    float srgb[256];
    float sxyz[4000];
    LabLut() {
        for (int i = 0; i < 256; i++) {
            float f = static_cast<float>(i) / 255.0f;
            if (f > 0.04045) srgb[i] = powf((f + 0.055f) / 1.055f, 2.4f);
            else srgb[i] = f / 12.92f;
        }
        for (int i = 0; i < 4000; i++) {
            float f = static_cast<float>(i) / 4000.0f;
            if (f > 0.008856) sxyz[i] = static_cast<float>(powf(f, 0.3333f));
            else sxyz[i] = static_cast<float>((7.787 * f) + (16.0 / 116.0));
        }
    }
};
const LabLut& lut() { static LabLut l; return l; }

inline void rgb2lab(unsigned char R, unsigned char G, unsigned char B, float& L, float& A, float& B2) {
    const LabLut& t = lut();
    float fr = t.srgb[R], fg = t.srgb[G], fb = t.srgb[B];
    const float x = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
    const float y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
    const float z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
    float vx = x / 0.95047f, vy = y, vz = z / 1.08883f;
    // PCL indexes sXYZ_LUT[int(v*4000)] without a bound; v can reach 1.0 (white) -> index 4000 is one past
    // the table. Clamped to 3999 here and on the device (documented deviation).
    auto li = [](float v) { int i = static_cast<int>(v * 4000); return i < 0 ? 0 : (i > 3999 ? 3999 : i); };
    vx = t.sxyz[li(vx)]; vy = t.sxyz[li(vy)]; vz = t.sxyz[li(vz)];
    L = 116.0f * vy - 16.0f; if (L > 100) L = 100.0f;
    A = 500.0f * (vx - vy); if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
    B2 = 200.0f * (vy - vz); if (B2 > 120) B2 = 120.0f; else if (B2 < -120) B2 = -120.0f;
}
inline void unpack_rgb(uint32_t rgba, unsigned char& r, unsigned char& g, unsigned char& b) {
    r = (rgba >> 16) & 0xff; g = (rgba >> 8) & 0xff; b = rgba & 0xff;   // PCL PointXYZRGB packing
}

/* ---- SHOT / CSHOT -----------------------------------------------------------------------
 * PCL: SHOTEstimation::computePointSHOT, createBinDistanceShape, interpolateSingleChannel /
 * SHOTColorEstimation::interpolateDoubleChannel, normalizeHistogram;
 * callers ref: features/features_shot.cpp:37-60, features_cshot.cpp:37-82. */
const double PST_PI = 3.1415926535897932384626433832795;
const double PST_RAD_45 = 0.78539816339744830961566084581988;
const double PST_RAD_90 = 1.5707963267948966192313216916398;
const double PST_RAD_135 = 2.3561944901923449288469825374596;
const double PST_RAD_PI_7_8 = 2.7488935718910690836548129603691;

template <bool COLOR>
void shot_one(const ObjGrid& g, const float* nx, const float* ny, const float* nz, const uint32_t* rgba,
              float cx, float cy, float cz, uint32_t kp_rgba, const float* lrf, float radius_f,
              std::vector<std::pair<float, int>>& nb, float* shot, uint32_t* count_out) {
    const int D = COLOR ? 1344 : 352;
    const int nr_shape = 10, nr_color = 30, max_sectors = 32;
    auto set_nan = [&]() { for (int i = 0; i < D; ++i) shot[i] = kNaN; };
    if (count_out) *count_out = 0;
    if (!std::isfinite(lrf[0]) || !std::isfinite(lrf[3]) || !std::isfinite(lrf[6]) || !finite3(cx, cy, cz)) { set_nan(); return; }
    g.radius(cx, cy, cz, radius_f, nb);
    if (count_out) *count_out = static_cast<uint32_t>(nb.size());
    if (nb.size() < 5) { set_nan(); return; }     // also covers "search returned 0"

    const double search_radius = static_cast<double>(radius_f);
    const double radius3_4 = (search_radius * 3) / 4, radius1_4 = search_radius / 4, radius1_2 = search_radius / 2;
    const float fx[3] = {lrf[0], lrf[1], lrf[2]}, fy[3] = {lrf[3], lrf[4], lrf[5]}, fz[3] = {lrf[6], lrf[7], lrf[8]};

    float LRef = 0, aRef = 0, bRef = 0;
    if (COLOR) {
        unsigned char r, gg, b; unpack_rgb(kp_rgba, r, gg, b);
        rgb2lab(r, gg, b, LRef, aRef, bRef);
        LRef /= 100.0f; aRef /= 120.0f; bRef /= 120.0f;
    }
    for (int i = 0; i < D; ++i) shot[i] = 0.f;

    for (size_t i = 0; i < nb.size(); ++i) {
        const int p = nb[i].second;
        // createBinDistanceShape: float dot of the Vector4f normal with the frame z axis
        if (!finite3(nx[p], ny[p], nz[p])) continue;
        // Eigen Vector4f dot (w components 0): ((a0*b0 + a1*b1) + a2*b2) + 0
        double cosineDesc = static_cast<double>(((nx[p] * fz[0] + ny[p] * fz[1]) + nz[p] * fz[2]));
        if (cosineDesc > 1.0) cosineDesc = 1.0;
        if (cosineDesc < -1.0) cosineDesc = -1.0;
        double binDistance = ((1.0 + cosineDesc) * nr_shape) / 2;
        double binDistanceColor = 0;
        if (COLOR) {
            unsigned char r, gg, b; unpack_rgb(rgba[p], r, gg, b);
            float L, a, bb; rgb2lab(r, gg, b, L, a, bb);
            L /= 100.0f; a /= 120.0f; bb /= 120.0f;
            double colorDistance = (std::fabs(LRef - L) + ((std::fabs(aRef - a) + std::fabs(bRef - bb)) / 2)) / 3;
            if (colorDistance > 1.0) colorDistance = 1.0;
            if (colorDistance < 0.0) colorDistance = 0.0;
            binDistanceColor = colorDistance * nr_color;
        }

        const float dx = g.x[p] - cx, dy = g.y[p] - cy, dz = g.z[p] - cz;
        const double distance = std::sqrt(static_cast<double>(nb[i].first));
        if (std::fabs(distance - 0.0) < 1E-15) continue;   // areEquals(distance, 0.0)

        double xInFeatRef = static_cast<double>((dx * fx[0] + dy * fx[1]) + dz * fx[2]);
        double yInFeatRef = static_cast<double>((dx * fy[0] + dy * fy[1]) + dz * fy[2]);
        double zInFeatRef = static_cast<double>((dx * fz[0] + dy * fz[1]) + dz * fz[2]);
        if (std::fabs(yInFeatRef) < 1E-30) yInFeatRef = 0;
        if (std::fabs(xInFeatRef) < 1E-30) xInFeatRef = 0;
        if (std::fabs(zInFeatRef) < 1E-30) zInFeatRef = 0;

        unsigned char bit4 = ((yInFeatRef > 0) || ((yInFeatRef == 0.0) && (xInFeatRef < 0))) ? 1 : 0;
        unsigned char bit3 = static_cast<unsigned char>(((xInFeatRef > 0) || ((xInFeatRef == 0.0) && (yInFeatRef > 0))) ? !bit4 : bit4);
        int desc_index = (bit4 << 3) + (bit3 << 2);
        desc_index = desc_index << 1;
        if ((xInFeatRef * yInFeatRef > 0) || (xInFeatRef == 0.0))
            desc_index += (std::fabs(xInFeatRef) >= std::fabs(yInFeatRef)) ? 0 : 4;
        else
            desc_index += (std::fabs(xInFeatRef) > std::fabs(yInFeatRef)) ? 4 : 0;
        desc_index += zInFeatRef > 0 ? 1 : 0;
        desc_index += (distance > radius1_2) ? 2 : 0;

        const int step_shape = static_cast<int>(std::floor(binDistance + 0.5));
        const int vol_shape = desc_index * (nr_shape + 1);
        binDistance -= step_shape;
        double wShape = 1 - std::fabs(binDistance);
        if (binDistance > 0) shot[vol_shape + ((step_shape + 1) % nr_shape)] += static_cast<float>(binDistance);
        else shot[vol_shape + ((step_shape - 1 + nr_shape) % nr_shape)] -= static_cast<float>(binDistance);

        int step_color = 0, vol_color = 0;
        double wColor = 0;
        const int stride = max_sectors * (nr_shape + 1);   // 352
        if (COLOR) {
            step_color = static_cast<int>(std::floor(binDistanceColor + 0.5));
            vol_color = stride + desc_index * (nr_color + 1);
            binDistanceColor -= step_color;
            wColor = 1 - std::fabs(binDistanceColor);
            if (binDistanceColor > 0) shot[vol_color + ((step_color + 1) % nr_color)] += static_cast<float>(binDistanceColor);
            else shot[vol_color + ((step_color - 1 + nr_color) % nr_color)] -= static_cast<float>(binDistanceColor);
        }
        auto dep = [&](int sector, double v) {   // deposit v into both channels of 'sector' at their own steps
            shot[sector * (nr_shape + 1) + step_shape] += static_cast<float>(v);
            if (COLOR) shot[stride + sector * (nr_color + 1) + step_color] += static_cast<float>(v);
        };

        // radial
        if (distance > radius1_2) {
            double rd = (distance - radius3_4) / radius1_2;
            if (distance > radius3_4) { wShape += 1 - rd; wColor += 1 - rd; }
            else { wShape += 1 + rd; wColor += 1 + rd; dep(desc_index - 2, -rd); }
        } else {
            double rd = (distance - radius1_4) / radius1_2;
            if (distance < radius1_4) { wShape += 1 + rd; wColor += 1 + rd; }
            else { wShape += 1 - rd; wColor += 1 - rd; dep(desc_index + 2, rd); }
        }
        // elevation
        double inclinationCos = zInFeatRef / distance;
        if (inclinationCos < -1.0) inclinationCos = -1.0;
        if (inclinationCos > 1.0) inclinationCos = 1.0;
        const double inclination = std::acos(inclinationCos);
        if (inclination > PST_RAD_90 || (std::fabs(inclination - PST_RAD_90) < 1e-30 && zInFeatRef <= 0)) {
            double id = (inclination - PST_RAD_135) / PST_RAD_90;
            if (inclination > PST_RAD_135) { wShape += 1 - id; wColor += 1 - id; }
            else { wShape += 1 + id; wColor += 1 + id; dep(desc_index + 1, -id); }
        } else {
            double id = (inclination - PST_RAD_45) / PST_RAD_90;
            if (inclination < PST_RAD_45) { wShape += 1 + id; wColor += 1 + id; }
            else { wShape += 1 - id; wColor += 1 - id; dep(desc_index - 1, id); }
        }
        // azimuth
        if (yInFeatRef != 0.0 || xInFeatRef != 0.0) {
            const double azimuth = std::atan2(yInFeatRef, xInFeatRef);
            const int sel = desc_index >> 2;
            double ad = (azimuth - (-PST_RAD_PI_7_8 + PST_RAD_45 * sel)) / PST_RAD_45;
            ad = std::max(-0.5, std::min(ad, 0.5));
            if (ad > 0) { wShape += 1 - ad; wColor += 1 - ad; dep((desc_index + 4) % max_sectors, ad); }
            else { wShape += 1 + ad; wColor += 1 + ad; dep((desc_index - 4 + max_sectors) % max_sectors, -ad); }
        }
        shot[vol_shape + step_shape] += static_cast<float>(wShape);
        if (COLOR) shot[vol_color + step_color] += static_cast<float>(wColor);
    }
    // normalizeHistogram: double accumulate of float squares, divide by float(norm)
    double acc = 0;
    for (int j = 0; j < D; ++j) acc += shot[j] * shot[j];
    acc = std::sqrt(acc);
    const float fn = static_cast<float>(acc);
    for (int j = 0; j < D; ++j) shot[j] /= fn;
}

/* ---- FPFH: PCL computePairFeatures / FPFHEstimation; all float ------- */
bool pair_features(const float* p1, const float* n1, const float* p2, const float* n2,
                   float& f1, float& f2, float& f3, float& f4) {
    float dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    f4 = std::sqrt((dp[0] * dp[0] + dp[1] * dp[1]) + dp[2] * dp[2]);
    if (f4 == 0.0f) { f1 = f2 = f3 = f4 = 0.0f; return false; }
    float a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
    float angle1 = ((a[0] * dp[0] + a[1] * dp[1]) + a[2] * dp[2]) / f4;
    float angle2 = ((b[0] * dp[0] + b[1] * dp[1]) + b[2] * dp[2]) / f4;
    if (std::acos(std::fabs(angle1)) > std::acos(std::fabs(angle2))) {
        std::swap(a[0], b[0]); std::swap(a[1], b[1]); std::swap(a[2], b[2]);
        dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2];
        f3 = -angle2;
    } else f3 = angle1;
    float v[3] = {dp[1] * a[2] - dp[2] * a[1], dp[2] * a[0] - dp[0] * a[2], dp[0] * a[1] - dp[1] * a[0]};
    float vn = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (vn == 0.0f) { f1 = f2 = f3 = f4 = 0.0f; return false; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    float w[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};
    f2 = (v[0] * b[0] + v[1] * b[1]) + v[2] * b[2];
    f1 = std::atan2((w[0] * b[0] + w[1] * b[1]) + w[2] * b[2], (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]);
    return true;
}

inline int clamp_bin(int h, int n) { return h < 0 ? 0 : (h >= n ? n - 1 : h); }

void spfh_one(const ObjGrid& g, const float* nx, const float* ny, const float* nz, int p, float radius,
              std::vector<std::pair<float, int>>& nb, float* h33) {
    for (int i = 0; i < 33; ++i) h33[i] = 0.f;
    g.radius(g.x[p], g.y[p], g.z[p], radius, nb);
    if (nb.empty()) return;
    const float hist_incr = 100.0f / static_cast<float>(nb.size() - 1);
    const float d_pi = 1.0f / (2.0f * static_cast<float>(M_PI));
    const float pp[3] = {g.x[p], g.y[p], g.z[p]}, pn[3] = {nx[p], ny[p], nz[p]};
    for (size_t i = 0; i < nb.size(); ++i) {
        const int q = nb[i].second;
        if (q == p) continue;
        const float qp[3] = {g.x[q], g.y[q], g.z[q]}, qn[3] = {nx[q], ny[q], nz[q]};
        float f1, f2, f3, f4;
        if (!pair_features(pp, pn, qp, qn, f1, f2, f3, f4)) continue;
        int h = static_cast<int>(std::floor(11 * ((f1 + M_PI) * d_pi)));
        h33[clamp_bin(h, 11)] += hist_incr;
        h = static_cast<int>(std::floor(11 * ((f2 + 1.0) * 0.5)));
        h33[11 + clamp_bin(h, 11)] += hist_incr;
        h = static_cast<int>(std::floor(11 * ((f3 + 1.0) * 0.5)));
        h33[22 + clamp_bin(h, 11)] += hist_incr;
    }
}

/* ---- FLANN distance functors (ref: utils/distance.h:45,65) --------------------- */
float dist_l2(const float* a, const float* b, int size) {
    float result = 0.f, diff0, diff1, diff2, diff3;
    const float* last = a + size;
    const float* lastgroup = last - 3;
    while (a < lastgroup) {
        diff0 = a[0] - b[0]; diff1 = a[1] - b[1]; diff2 = a[2] - b[2]; diff3 = a[3] - b[3];
        result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
        a += 4; b += 4;
    }
    while (a < last) { diff0 = *a++ - *b++; result += diff0 * diff0; }
    return result;
}
float dist_chi2(const float* a, const float* b, int size) {
    float result = 0.f, sum, diff;
    const float* last = a + size;
    while (a < last) {
        sum = *a + *b;
        if (sum > 0) { diff = *a - *b; result += diff * diff / sum; }
        ++a; ++b;
    }
    return result;
}
inline float dist_any(int metric, const float* a, const float* b, int n) { return metric == 1 ? dist_chi2(a, b, n) : dist_l2(a, b, n); }

/* ---- quaternion helpers (ref: utils/utils.cpp:136-178, 342-394, 560-574); quaternion = (w,x,y,z) */
struct Quat { float w, x, y, z; };
inline Quat qmul(const Quat& a, const Quat& b) {   // boost::math::quaternion operator*
    Quat r;
    r.w = +a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = +a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = +a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = +a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
inline Quat qconj(const Quat& q) { return Quat{q.w, -q.x, -q.y, -q.z}; }
/* getRotQuaternion: the Eigen matrix holds the axes as COLUMNS, its column-major memory is then read as a
 * row-major matrix by matrix2Quat => matrix[i][j] = axis_i[j] (rows = axes), Ogre's algorithm. */
Quat rot_quaternion(const float* lrf9) {
    float m[3][3] = {{lrf9[0], lrf9[1], lrf9[2]}, {lrf9[3], lrf9[4], lrf9[5]}, {lrf9[6], lrf9[7], lrf9[8]}};
    float q[4] = {0, 0, 0, 1};   // x,y,z,w
    float trace = m[0][0] + m[1][1] + m[2][2];
    float root;
    if (trace > 0.0f) {
        root = sqrtf(trace + 1.0f);
        q[3] = 0.5f * root;
        root = 0.5f / root;
        q[0] = (m[2][1] - m[1][2]) * root;
        q[1] = (m[0][2] - m[2][0]) * root;
        q[2] = (m[1][0] - m[0][1]) * root;
    } else {
        static const size_t next[3] = {1, 2, 0};
        size_t i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        size_t j = next[i], k = next[j];
        root = sqrtf(static_cast<float>(m[i][i] - m[j][j] - m[k][k] + 1.0));
        q[i] = 0.5f * root;
        root = 0.5f / root;
        q[3] = (m[k][j] - m[j][k]) * root;
        q[j] = (m[j][i] + m[i][j]) * root;
        q[k] = (m[k][i] + m[i][k]) * root;
    }
    return Quat{q[3], q[0], q[1], q[2]};
}
inline void quat_rotate(const Quat& q, float* p) {      // q * p * conj(q)
    Quat t = qmul(qmul(q, Quat{0, p[0], p[1], p[2]}), qconj(q));
    p[0] = t.x; p[1] = t.y; p[2] = t.z;
}
inline void quat_rotate_inv(const Quat& q, float* p) {  // conj(q) * p * q
    Quat t = qmul(qmul(qconj(q), Quat{0, p[0], p[1], p[2]}), q);
    p[0] = t.x; p[1] = t.y; p[2] = t.z;
}

/* gaussDist (ref: codebook/codeword_distribution.cpp:23-26); computed in double, returned as float */
inline float gauss_dist(float sigmaSqr, float dist) {
    return static_cast<float>((1 / std::sqrt(2 * M_PI * sigmaSqr)) * std::exp(-std::pow(dist, 2) / (2 * sigmaSqr)));
}

/* ---- mean shift (ref: voting/voting_mean_shift.cpp) ---------------------------------------- */
struct MSVote { float p[3]; float w; int inst; int slot; };

inline float ms_kernel(int kernel, float x) {           // :378-417
    if (kernel == 0) return static_cast<float>(std::exp(-0.5 * x));
    if (kernel == 1) return 1.f;
    return 0.f;
}
inline float ms_kernel_derivative(int kernel, float x) {
    if (kernel == 0) { float profile = static_cast<float>(std::exp(-0.5 * x)); return -0.5f * profile; }
    if (kernel == 1) return 1.f;
    return 0.f;
}
/* radius search among votes (pcl::search::KdTree, sorted, d2 < h2) */
void vote_radius(const std::vector<MSVote>& votes, const float* q, float h, std::vector<std::pair<float, int>>& out) {
    out.clear();
    if (!finite3(q[0], q[1], q[2])) return;
    const float h2 = radius_sq(h);
    for (size_t i = 0; i < votes.size(); ++i) {
        float d2 = sqdist3(votes[i].p[0], votes[i].p[1], votes[i].p[2], q[0], q[1], q[2]);
        if (d2 < h2) out.emplace_back(d2, static_cast<int>(i));
    }
    std::sort(out.begin(), out.end());
}
struct KeyLess {   // mapCompareVector :419-429 — (z, y, x) lexicographic
    bool operator()(const std::array<int, 3>& a, const std::array<int, 3>& b) const {
        if (a[2] < b[2]) return true;
        if (a[2] == b[2] && a[1] < b[1]) return true;
        if (a[2] == b[2] && a[1] == b[1] && a[0] < b[0]) return true;
        return false;
    }
};
void create_seeds(const std::vector<MSVote>& votes, float binSize, std::vector<MSVote>& seeds) {   // :431-481
    seeds.clear();
    if (binSize == 0) { seeds = votes; return; }
    std::map<std::array<int, 3>, std::pair<int, float>, KeyLess> m;
    for (const MSVote& v : votes) {
        std::array<int, 3> key = {static_cast<int>(std::floor((v.p[0] / binSize) + 0.5)),
                                  static_cast<int>(std::floor((v.p[1] / binSize) + 0.5)),
                                  static_cast<int>(std::floor((v.p[2] / binSize) + 0.5))};
        auto it = m.find(key);
        if (it != m.end()) { it->second.first += 1; it->second.second = it->second.second + v.w; }
        else m[key] = std::make_pair(1, v.w);
    }
    for (auto& kv : m) {
        if (kv.second.first >= 1) {
            MSVote s{};
            s.p[0] = kv.first[0] * binSize; s.p[1] = kv.first[1] * binSize; s.p[2] = kv.first[2] * binSize;
            s.w = kv.second.second; s.inst = 0; s.slot = -1;
            seeds.push_back(s);
        }
    }
}
bool compute_mean_shift(const std::vector<MSVote>& votes, const float* center, float* newCenter, float h, int kernel,
                        std::vector<std::pair<float, int>>& nb) {   // :331-376
    vote_radius(votes, center, h, nb);
    if (nb.empty()) return false;
    float shifted[3] = {0, 0, 0};
    double totalWeight = 0;
    for (auto& pr : nb) {
        const MSVote& v = votes[pr.second];
        float u = pr.first / (h * h);
        float gw = -ms_kernel_derivative(kernel, u) * v.w;
        shifted[0] += gw * v.p[0]; shifted[1] += gw * v.p[1]; shifted[2] += gw * v.p[2];
        totalWeight += gw;
    }
    if (totalWeight != 0) {
        // Eigen: Vector3f /= double -> scalar is cast to float first
        const float tw = static_cast<float>(totalWeight);
        shifted[0] /= tw; shifted[1] /= tw; shifted[2] /= tw;
    }
    newCenter[0] = shifted[0]; newCenter[1] = shifted[1]; newCenter[2] = shifted[2];
    return true;
}
float estimate_density(std::vector<MSVote>& votes, const float* pos, float h, int kernel, bool reweight,
                       std::vector<int>* cluster, std::vector<std::pair<float, int>>& nb) {   // :247-328
    vote_radius(votes, pos, h, nb);
    if (cluster) cluster->clear();
    if (nb.empty()) return 0;
    float density = 0;
    for (auto& pr : nb) {
        MSVote& v = votes[pr.second];
        float u = pr.first / (h * h);
        float weight = ms_kernel(kernel, u) * v.w;
        if (reweight) v.w = weight;
        if (cluster) cluster->push_back(pr.second);
        density += weight;
    }
    return density;
}
inline float norm3(const float* a, const float* b) {
    float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
typedef std::array<float, 3> V3;

void average_neighbor_maxima(const std::vector<V3>& centers, float radius, std::vector<V3>& maxima,
                             const std::vector<float>& densities) {   // ref: voting/maxima_handler.cpp:94-157
    const int n = static_cast<int>(centers.size());
    std::vector<std::vector<int>> dup(n);
    for (int i = 0; i < n; ++i) dup[i].push_back(i);
    std::vector<bool> duplicate(n, false);
    for (int k = 0; k < n; ++k) {
        if (duplicate[k]) continue;
        for (int j = k + 1; j < n; ++j) {
            if (duplicate[j]) continue;
            if (norm3(centers[k].data(), centers[j].data()) < radius) { duplicate[j] = true; dup[k].push_back(j); }
        }
    }
    maxima.clear();
    for (int i = 0; i < n; ++i) {
        if (dup[i].size() == 1) maxima.push_back(centers[dup[i][0]]);
        else {
            V3 avg = {0, 0, 0};
            float sum = 0;
            for (int j : dup[i]) {
                avg[0] += centers[j][0] * densities[j]; avg[1] += centers[j][1] * densities[j]; avg[2] += centers[j][2] * densities[j];
                sum += densities[j];
            }
            avg[0] /= sum; avg[1] /= sum; avg[2] /= sum;
            maxima.push_back(avg);
        }
    }
}
void suppress_neighbor_maxima(const std::vector<V3>& centers, const std::vector<float>& densities, float radius,
                              std::vector<V3>& maxima) {   // ref: voting/maxima_handler.cpp:51-92
    maxima.clear();
    std::vector<float> work(densities);
    for (;;) {
        auto it = std::max_element(work.begin(), work.end());
        float mx = -1;
        if (it != work.end()) mx = *it;
        if (mx != -1) {
            const size_t idx = it - work.begin();
            const V3 c = centers[idx];
            maxima.push_back(c);
            work[idx] = -1;
            for (size_t i = 0; i < centers.size(); ++i)
                if (norm3(c.data(), centers[i].data()) < radius) work[i] = -1;
        } else break;
    }
}

struct Maximum {
    V3 pos; float weight; int cls; int inst; float inst_weight; V3 bbox; int n_votes;
    float quat[4] = {1.f, 0.f, 0.f, 0.f};
};

// Utils::quatWeightedAverage (utils/utils.cpp:617-665): scatter matrix sum w q q^T (float sums, member order), then an eigenvector.
// The reference takes whichever eigenvector Eigen::EigenSolver lists FIRST (its loop over "eigenvalues.cols()" only ever looks at
// index 0 of an unordered general solver): not reproducible. This restatement and the HIP path take the eigenvector of the LARGEST
// eigenvalue of the symmetric matrix (cyclic Jacobi in double) with its first non-zero component positive. Parity unpinned.
void quat_from_scatter(const float* S, float* q) {
    double A[4][4] = {{S[0], S[1], S[2], S[3]}, {S[1], S[4], S[5], S[6]}, {S[2], S[5], S[7], S[8]}, {S[3], S[6], S[8], S[9]}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        if (!(off > 1e-30)) break;
        for (int p = 0; p < 3; ++p) for (int r = p + 1; r < 4; ++r) {
            if (std::fabs(A[p][r]) < 1e-300) continue;
            const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 4; ++k) { const double x = A[k][p], y = A[k][r]; A[k][p] = c * x - sn * y; A[k][r] = sn * x + c * y; }
            for (int k = 0; k < 4; ++k) { const double x = A[p][k], y = A[r][k]; A[p][k] = c * x - sn * y; A[r][k] = sn * x + c * y; }
            for (int k = 0; k < 4; ++k) { const double x = V[k][p], y = V[k][r]; V[k][p] = c * x - sn * y; V[k][r] = sn * x + c * y; }
        }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[best][best]) best = i;
    double v[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
    double sgn = 1.0;
    for (int i = 0; i < 4; ++i) if (v[i] != 0.0) { sgn = v[i] < 0 ? -1.0 : 1.0; break; }
    for (int i = 0; i < 4; ++i) q[i] = static_cast<float>(sgn * v[i]);
}
void quat_scatter_add(float* S, float w, const float* q) {
    S[0] += w * q[0] * q[0]; S[1] += w * q[0] * q[1]; S[2] += w * q[0] * q[2]; S[3] += w * q[0] * q[3];
    S[4] += w * q[1] * q[1]; S[5] += w * q[1] * q[2]; S[6] += w * q[1] * q[3];
    S[7] += w * q[2] * q[2]; S[8] += w * q[2] * q[3]; S[9] += w * q[3] * q[3];
}


// Voting::findMaxima per-maximum block (voting.cpp:131-236): instance id by the largest summed weight (ties and the map order
// give the smallest id), weight = sum of the cluster's vote weights, weighted mean bounding-box size.
void append_maximum(const std::vector<MSVote>& votes, const std::vector<V3>& vbbox, const std::vector<int>& cluster, const V3& pos, int c,
                    int min_votes_threshold, std::vector<Maximum>& maxima, const float* vbq = nullptr) {
    if (static_cast<int>(cluster.size()) < min_votes_threshold || cluster.empty()) return;
    std::map<unsigned, float> instance_weights;
    for (int vi : cluster) {
        const float w = votes[vi].w; const unsigned id = static_cast<unsigned>(votes[vi].inst);
        auto it = instance_weights.find(id);
        if (it != instance_weights.end()) it->second += w; else instance_weights.insert({id, w});
    }
    unsigned max_id = 0; float best = 0; bool have = false;
    for (auto& it : instance_weights) if (it.second > best) { best = it.second; max_id = it.first; have = true; }
    Maximum m;
    m.cls = c; m.inst = have ? static_cast<int>(max_id) : -1;
    m.inst_weight = have ? instance_weights[max_id] : 0.f;
    m.pos = pos; m.n_votes = static_cast<int>(cluster.size());
    float maxWeight = 0; V3 bs = {0, 0, 0};
    for (int vi : cluster) {
        const float nw = votes[vi].w;
        bs[0] += nw * vbbox[vi][0]; bs[1] += nw * vbbox[vi][1]; bs[2] += nw * vbbox[vi][2];
        maxWeight += nw;
    }
    m.weight = maxWeight;
    bs[0] /= maxWeight; bs[1] /= maxWeight; bs[2] /= maxWeight;
    m.bbox = bs;
    if (vbq) {                                     // Voting.AverageRotation (voting.cpp:186-215)
        float S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int vi : cluster) quat_scatter_add(S, votes[vi].w, vbq + static_cast<size_t>(votes[vi].slot) * 4);
        quat_from_scatter(S, m.quat);
    }
    maxima.push_back(m);
}

// MaximaHandler::filterMaxima "Merge" = mergeAndFilterMaxima(maxima, true) (maxima_handler.cpp:300-387) with mergeMaxima (:390-440).
// The list enters in class order, then iFindMaxima order (the reference's order inside a class is unordered: omp critical push).
// merged_maxima starts with maxima.size() default-constructed entries (weight 0): they only matter if no merged weight is > 0.
void filter_maxima_merge(std::vector<Maximum>& maxima, const float* class_bw, float bandwidth) {
    auto sdist = [&](int cls) { return class_bw ? class_bw[cls] : bandwidth; };
    std::vector<Maximum> out;
    std::vector<bool> dirty(maxima.size(), false);
    for (size_t i = 0; i < maxima.size(); ++i) {
        if (dirty[i]) continue;
        const float sd = sdist(maxima[i].cls);
        std::vector<size_t> close;
        for (size_t j = i + 1; j < maxima.size(); ++j) {
            if (dirty[j]) continue;
            const float dist = norm3(maxima[j].pos.data(), maxima[i].pos.data());
            if (dist < sd && sdist(maxima[j].cls) <= sd) { close.push_back(j); dirty[j] = true; }
        }
        if (close.empty()) { out.push_back(maxima[i]); continue; }
        close.push_back(i);
        std::map<unsigned, std::vector<size_t>> same;                       // class id -> members in list order
        for (size_t x : close) same[static_cast<unsigned>(maxima[x].cls)].push_back(x);
        Maximum best; best.weight = 0; best.cls = -1; best.inst = -1; best.inst_weight = 0; best.pos = {0, 0, 0}; best.bbox = {0, 0, 0}; best.n_votes = 0;
        bool have = false;
        for (auto& kv : same) {
            Maximum r; r.pos = {0, 0, 0}; r.weight = 0; r.bbox = {0, 0, 0}; r.n_votes = 0; r.cls = -1; r.inst = -1; r.inst_weight = 0;
            std::map<unsigned, float> inst;
            for (size_t x : kv.second) {
                const Maximum& m = maxima[x];
                for (int d = 0; d < 3; ++d) r.pos[d] = (r.pos[d] * r.weight + m.pos[d] * m.weight) / (r.weight + m.weight);
                for (int d = 0; d < 3; ++d) r.bbox[d] = (r.bbox[d] * r.weight + m.bbox[d] * m.weight) / (r.weight + m.weight);
                { float S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; quat_scatter_add(S, r.weight, r.quat); quat_scatter_add(S, m.weight, m.quat); quat_from_scatter(S, r.quat); }
                r.cls = m.cls; r.weight += m.weight; r.n_votes += m.n_votes;
                auto it = inst.find(static_cast<unsigned>(m.inst));
                if (it != inst.end()) it->second += m.inst_weight; else inst.insert({static_cast<unsigned>(m.inst), m.inst_weight});
                unsigned max_id = 0; float bw = 0; bool hv = false;
                for (auto& e : inst) if (e.second > bw) { bw = e.second; max_id = e.first; hv = true; }
                r.inst = hv ? static_cast<int>(max_id) : -1; r.inst_weight = hv ? inst[max_id] : 0.f;
            }
            if (r.weight > best.weight) { best = r; have = true; }
        }
        if (have) out.push_back(best);
    }
    maxima.swap(out);
}

// Voting::findMaxima tail (voting.cpp:272, 298-323, 441-462): sort, normalise, MinThreshold, BestK, outputs of one object
// MaximaHandler::filterMaxima "Simple" = suppressNeighborMaxima2 (maxima_handler.cpp:227-268): greedy NMS over all classes
void filter_maxima_simple(std::vector<Maximum>& maxima, float radius) {
    std::vector<float> work;
    for (auto& m : maxima) work.push_back(m.weight);
    std::vector<Maximum> out;
    for (;;) {
        int mi = -1;
        for (size_t i = 0; i < work.size(); ++i) if (work[i] != -1 && (mi < 0 || work[i] > work[mi])) mi = static_cast<int>(i);   // std::max_element: first largest
        if (mi < 0) break;
        out.push_back(maxima[mi]);
        work[mi] = -1;
        for (size_t i = 0; i < work.size(); ++i) {
            const float dx = maxima[mi].pos[0] - maxima[i].pos[0], dy = maxima[mi].pos[1] - maxima[i].pos[1], dz = maxima[mi].pos[2] - maxima[i].pos[2];
            if (std::sqrt(dx * dx + dy * dy + dz * dz) < radius) work[i] = -1;
        }
    }
    maxima.swap(out);
}

void finish_object(std::vector<Maximum>& maxima, int o, int C, int cap, float min_threshold, int best_k,
                   int32_t* n_max_out, float* mpos, float* mw, int32_t* mcls, int32_t* minst, float* miw, float* mbs, int32_t* mnv, float* class_score,
                   float* mbq = nullptr) {
        // sort (stable; std::sort in the reference leaves equal weights unordered), voting.cpp:272
        std::stable_sort(maxima.begin(), maxima.end(), [](const Maximum& a, const Maximum& b) { return a.weight > b.weight; });
        // normalizeWeights :441-462
        float sum = 0, sum_inst = 0;
        for (auto& m : maxima) { sum += m.weight; sum_inst += m.inst_weight; }
        for (auto& m : maxima) { m.weight = sum != 0 ? m.weight / sum : 0; m.inst_weight = sum_inst != 0 ? m.inst_weight / sum_inst : 0; }
        float thr = min_threshold;
        if (thr < 0) { float mxw = maxima.size() > 0 ? maxima.front().weight : 0.0f; thr = -thr * mxw; }   // :304-309
        std::vector<Maximum> filtered;
        for (auto& m : maxima) if (m.weight >= thr) filtered.push_back(m);
        maxima.swap(filtered);
        if (best_k > 0 && static_cast<int>(maxima.size()) >= best_k) maxima.resize(best_k);          // :322-323
        const int nm = std::min(static_cast<int>(maxima.size()), cap);
        n_max_out[o] = nm;
        for (int c = 0; c < C; ++c) class_score[static_cast<size_t>(o) * C + c] = 0.f;
        for (size_t i = 0; i < maxima.size(); ++i) {
            float& cs = class_score[static_cast<size_t>(o) * C + maxima[i].cls];
            if (maxima[i].weight > cs) cs = maxima[i].weight;
        }
        for (int i = 0; i < cap; ++i) {
            const size_t t = static_cast<size_t>(o) * cap + i;
            const bool ok = i < nm;
            mpos[t * 3] = ok ? maxima[i].pos[0] : 0.f; mpos[t * 3 + 1] = ok ? maxima[i].pos[1] : 0.f; mpos[t * 3 + 2] = ok ? maxima[i].pos[2] : 0.f;
            mw[t] = ok ? maxima[i].weight : 0.f; mcls[t] = ok ? maxima[i].cls : -1; minst[t] = ok ? maxima[i].inst : -1;
            miw[t] = ok ? maxima[i].inst_weight : 0.f; mnv[t] = ok ? maxima[i].n_votes : 0;
            if (mbs) { mbs[t * 3] = ok ? maxima[i].bbox[0] : 0.f; mbs[t * 3 + 1] = ok ? maxima[i].bbox[1] : 0.f; mbs[t * 3 + 2] = ok ? maxima[i].bbox[2] : 0.f; }
            if (mbq) for (int d = 0; d < 4; ++d) mbq[t * 4 + d] = ok ? maxima[i].quat[d] : (d == 0 ? 1.f : 0.f);
        }
}
}  // namespace

/* ============================== exported C functions ======================================= */
extern "C" {

void ismref_set_num_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}
int ismref_get_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int ismref_radius_search(int n, const float* x, const float* y, const float* z, float qx, float qy, float qz,
                         float radius, int cap, int32_t* idx_out, float* d2_out) {
    ObjGrid g; g.build(n, x, y, z, radius);
    std::vector<std::pair<float, int>> nb;
    g.radius(qx, qy, qz, radius, nb);
    for (int i = 0; i < static_cast<int>(nb.size()) && i < cap; ++i) {
        if (idx_out) idx_out[i] = nb[i].second;
        if (d2_out) d2_out[i] = nb[i].first;
    }
    return static_cast<int>(nb.size());
}

/* pcl::NormalEstimationOMPWithEigVals::computeFeature (third_party/pcl_normal_3d_omp_with_eigenvalues/normal_3d_omp_with_eigenvalues.hpp:
 * 61-144, .h:112-180) as ImplicitShapeModel::computeNormals drives it for ConsistentNormalsMethod 0 / 1 (implicit_shape_model.cpp:
 * 969-1011). EXTERNAL parts restated from PCL 1.10 (parity unpinned): computeMeanAndCovarianceMatrix = single-pass float sums of
 * x*x .. z and x, y, z over the neighbours in search order (ascending distance), cov = E[ab] - E[a]E[b]; pcl::eigen33 = analytic
 * roots of the scaled matrix, eigenvector of the smallest root from the largest cross product of rows of (A - lambda I).
 * orientation 0: viewpoint (0,0,0); 1: the cloud shifted by its float centroid, viewpoint origin, normals inverted afterwards;
 * 2: no flip at all = pcl::NormalEstimation::computePointNormal(cloud, indices, nx, ny, nz, curvature), the call
 * NormalOrientation::processSHOTLRF makes for its patch-up loop (utils/normal_orientation.cpp:96-106). */
namespace {
void pcl_roots2(float b, float c, float* r) { r[0] = 0.f; float d = b * b - 4.0f * c; if (d < 0.0f) d = 0.0f; const float sd = std::sqrt(d); r[2] = 0.5f * (b + sd); r[1] = 0.5f * (b - sd); }
void pcl_roots(const float m[3][3], float* r) {
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { pcl_roots2(c2, c1, r); return; }
    const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = std::sqrt(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = std::sqrt(-a_over_3), theta = std::atan2(std::sqrt(-q), half_b) * s_inv3, ct = std::cos(theta), st = std::sin(theta);
    r[0] = c2_over_3 + 2.f * rho * ct; r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st); r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) { std::swap(r[1], r[2]); if (r[0] >= r[1]) std::swap(r[0], r[1]); }
    if (r[0] <= 0) pcl_roots2(c2, c1, r);
}
}  // namespace
int ismref_pca_normals(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z, float radius, int orientation,
                       float* nx, float* ny, float* nz) {
    for (int o = 0; o < n_obj; ++o) {
        const int n = static_cast<int>(po[o + 1] - po[o]);
        if (n == 0) continue;
        const float* X = x + po[o]; const float* Y = y + po[o]; const float* Z = z + po[o];
        std::vector<float> sx(X, X + n), sy(Y, Y + n), sz(Z, Z + n);
        if (orientation == 1) {                                   // pcl::compute3DCentroid into a Vector4f, then the shifted copy
            float c[3] = {0, 0, 0}; int cnt = 0;
            for (int i = 0; i < n; ++i) if (std::isfinite(X[i]) && std::isfinite(Y[i]) && std::isfinite(Z[i])) { c[0] += X[i]; c[1] += Y[i]; c[2] += Z[i]; ++cnt; }
            for (int d = 0; d < 3; ++d) c[d] /= static_cast<float>(cnt ? cnt : 1);
            for (int i = 0; i < n; ++i) { sx[i] -= c[0]; sy[i] -= c[1]; sz[i] -= c[2]; }
        }
        ObjGrid g; g.build(n, sx.data(), sy.data(), sz.data(), radius);
#pragma omp parallel
        {
            std::vector<std::pair<float, int>> nb;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < n; ++i) {
                float* out[3] = {nx + po[o] + i, ny + po[o] + i, nz + po[o] + i};
                *out[0] = *out[1] = *out[2] = std::nanf("");
                if (!(std::isfinite(sx[i]) && std::isfinite(sy[i]) && std::isfinite(sz[i]))) continue;
                g.radius(sx[i], sy[i], sz[i], radius, nb);
                if (nb.size() < 3) continue;
                float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (auto& q : nb) {
                    const float px = sx[q.second], py = sy[q.second], pz = sz[q.second];
                    a[0] += px * px; a[1] += px * py; a[2] += px * pz; a[3] += py * py; a[4] += py * pz; a[5] += pz * pz; a[6] += px; a[7] += py; a[8] += pz;
                }
                const float cntf = static_cast<float>(nb.size());
                for (float& v : a) v /= cntf;
                float cov[3][3];
                cov[0][0] = a[0] - a[6] * a[6]; cov[0][1] = a[1] - a[6] * a[7]; cov[0][2] = a[2] - a[6] * a[8];
                cov[1][1] = a[3] - a[7] * a[7]; cov[1][2] = a[4] - a[7] * a[8]; cov[2][2] = a[5] - a[8] * a[8];
                cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
                float scale = 0.f;
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(cov[r][c]));
                if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
                float sm[3][3];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sm[r][c] = cov[r][c] / scale;
                float roots[3];
                pcl_roots(sm, roots);
                for (int d = 0; d < 3; ++d) sm[d][d] -= roots[0];
                auto cross = [](const float* u, const float* v, float* w) { w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0]; };
                float v1[3], v2[3], v3[3];
                cross(sm[0], sm[1], v1); cross(sm[0], sm[2], v2); cross(sm[1], sm[2], v3);
                const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2], l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
                float nrm[3];
                if (l1 >= l2 && l1 >= l3) { const float s_ = std::sqrt(l1); for (int d = 0; d < 3; ++d) nrm[d] = v1[d] / s_; }
                else if (l2 >= l1 && l2 >= l3) { const float s_ = std::sqrt(l2); for (int d = 0; d < 3; ++d) nrm[d] = v2[d] / s_; }
                else { const float s_ = std::sqrt(l3); for (int d = 0; d < 3; ++d) nrm[d] = v3[d] / s_; }
                // flipNormalTowardsViewpointMod with the viewpoint at the origin of the (possibly shifted) cloud
                const float cos_theta = (0.f - sx[i]) * nrm[0] + (0.f - sy[i]) * nrm[1] + (0.f - sz[i]) * nrm[2];
                if (cos_theta < 0 && orientation != 2) { nrm[0] *= -1; nrm[1] *= -1; nrm[2] *= -1; }
                if (orientation == 1) { nrm[0] *= -1; nrm[1] *= -1; nrm[2] *= -1; }
                *out[0] = nrm[0]; *out[1] = nrm[1]; *out[2] = nrm[2];
            }
        }
    }
    return 0;
}

int ismref_shot_lrf(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z,
                    const uint32_t* ko, const float* kpx, const float* kpy, const float* kpz,
                    float radius, float* lrf9_out) {
    for (int o = 0; o < n_obj; ++o) {
        ObjGrid g; g.build(po[o + 1] - po[o], x + po[o], y + po[o], z + po[o], radius);
#pragma omp parallel
        {
            std::vector<std::pair<float, int>> nb; std::vector<double> vij;
#pragma omp for schedule(dynamic, 16)
            for (int k = static_cast<int>(ko[o]); k < static_cast<int>(ko[o + 1]); ++k)
                lrf_one(g, kpx[k], kpy[k], kpz[k], radius, nb, vij, lrf9_out + static_cast<size_t>(k) * 9);
        }
    }
    return 0;
}

int ismref_shot352(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z,
                   const float* nx, const float* ny, const float* nz,
                   const uint32_t* ko, const float* kpx, const float* kpy, const float* kpz,
                   const float* lrf9, float radius, float* desc_out, uint32_t* cnt) {
    for (int o = 0; o < n_obj; ++o) {
        ObjGrid g; g.build(po[o + 1] - po[o], x + po[o], y + po[o], z + po[o], radius);
#pragma omp parallel
        {
            std::vector<std::pair<float, int>> nb;
#pragma omp for schedule(dynamic, 16)
            for (int k = static_cast<int>(ko[o]); k < static_cast<int>(ko[o + 1]); ++k)
                shot_one<false>(g, nx + po[o], ny + po[o], nz + po[o], nullptr, kpx[k], kpy[k], kpz[k], 0,
                                lrf9 + static_cast<size_t>(k) * 9, radius, nb, desc_out + static_cast<size_t>(k) * 352,
                                cnt ? cnt + k : nullptr);
        }
    }
    return 0;
}

int ismref_cshot1344(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z,
                     const float* nx, const float* ny, const float* nz, const uint32_t* rgba,
                     const uint32_t* ko, const float* kpx, const float* kpy, const float* kpz,
                     const uint32_t* kp_rgba, const float* lrf9, float radius, float* desc_out, uint32_t* cnt) {
    if (!rgba || !kp_rgba) return -1;
    (void)lut();
    for (int o = 0; o < n_obj; ++o) {
        ObjGrid g; g.build(po[o + 1] - po[o], x + po[o], y + po[o], z + po[o], radius);
#pragma omp parallel
        {
            std::vector<std::pair<float, int>> nb;
#pragma omp for schedule(dynamic, 16)
            for (int k = static_cast<int>(ko[o]); k < static_cast<int>(ko[o + 1]); ++k)
                shot_one<true>(g, nx + po[o], ny + po[o], nz + po[o], rgba + po[o], kpx[k], kpy[k], kpz[k], kp_rgba[k],
                               lrf9 + static_cast<size_t>(k) * 9, radius, nb, desc_out + static_cast<size_t>(k) * 1344,
                               cnt ? cnt + k : nullptr);
        }
    }
    return 0;
}

int ismref_fpfh33(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z,
                  const float* nx, const float* ny, const float* nz,
                  const uint32_t* ko, const float* kpx, const float* kpy, const float* kpz,
                  float radius, float* desc_out, uint32_t* cnt) {
    for (int o = 0; o < n_obj; ++o) {
        const int n = po[o + 1] - po[o];
        const float *ox = x + po[o], *oy = y + po[o], *oz = z + po[o];
        const float *onx = nx + po[o], *ony = ny + po[o], *onz = nz + po[o];
        ObjGrid g; g.build(n, ox, oy, oz, radius);
        const int k0 = ko[o], k1 = ko[o + 1];
        // stage 1: union of keypoint neighbourhoods (FPFHEstimation::computeSPFHSignatures)
        std::vector<char> need(n, 0);
        std::vector<std::vector<std::pair<float, int>>> kp_nb(k1 - k0);
#pragma omp parallel for schedule(dynamic, 16)
        for (int k = k0; k < k1; ++k) g.radius(kpx[k], kpy[k], kpz[k], radius, kp_nb[k - k0]);
        for (auto& v : kp_nb) for (auto& pr : v) need[pr.second] = 1;
        std::vector<int> lookup(n, -1); std::vector<int> ulist;
        for (int i = 0; i < n; ++i) if (need[i]) { lookup[i] = static_cast<int>(ulist.size()); ulist.push_back(i); }
        std::vector<float> spfh(ulist.size() * 33);
#pragma omp parallel
        {
            std::vector<std::pair<float, int>> nb;
#pragma omp for schedule(dynamic, 16)
            for (int u = 0; u < static_cast<int>(ulist.size()); ++u)
                spfh_one(g, onx, ony, onz, ulist[u], radius, nb, &spfh[static_cast<size_t>(u) * 33]);
        }
        // stage 2: weightPointSPFHSignature
#pragma omp parallel for schedule(dynamic, 16)
        for (int k = k0; k < k1; ++k) {
            float* out = desc_out + static_cast<size_t>(k) * 33;
            const auto& nb = kp_nb[k - k0];
            if (cnt) cnt[k] = static_cast<uint32_t>(nb.size());
            if (!finite3(kpx[k], kpy[k], kpz[k]) || nb.empty()) { for (int i = 0; i < 33; ++i) out[i] = kNaN; continue; }
            float sum[3] = {0, 0, 0};
            for (int i = 0; i < 33; ++i) out[i] = 0.f;
            for (auto& pr : nb) {
                if (pr.first == 0) continue;
                const float weight = 1.0f / pr.first;
                const float* h = &spfh[static_cast<size_t>(lookup[pr.second]) * 33];
                for (int b = 0; b < 3; ++b)
                    for (int i = 0; i < 11; ++i) { float v = h[b * 11 + i] * weight; sum[b] += v; out[b * 11 + i] += v; }
            }
            for (int b = 0; b < 3; ++b) {
                if (sum[b] != 0) sum[b] = 100.0f / sum[b];
                for (int i = 0; i < 11; ++i) out[b * 11 + i] *= sum[b];
            }
        }
    }
    return 0;
}

int ismref_centroids(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z, float* out) {
    // pcl::compute3DCentroid(cloud, Eigen::Vector4d) — double accumulation over finite points
    for (int o = 0; o < n_obj; ++o) {
        double s[3] = {0, 0, 0}; unsigned cp = 0;
        for (uint32_t i = po[o]; i < po[o + 1]; ++i) {
            if (!finite3(x[i], y[i], z[i])) continue;
            s[0] += x[i]; s[1] += y[i]; s[2] += z[i]; ++cp;
        }
        for (int a = 0; a < 3; ++a) out[o * 3 + a] = cp ? static_cast<float>(s[a] / cp) : 0.f;
    }
    return 0;
}
int ismref_center_dist(int n_obj, const uint32_t* po, const float* x, const float* y, const float* z,
                       const uint32_t* ko, const float* kpx, const float* kpy, const float* kpz, float* out) {
    std::vector<float> c(static_cast<size_t>(n_obj) * 3);
    ismref_centroids(n_obj, po, x, y, z, c.data());
    for (int o = 0; o < n_obj; ++o)
        for (uint32_t k = ko[o]; k < ko[o + 1]; ++k) {
            const float q[3] = {kpx[k], kpy[k], kpz[k]};
            out[k] = norm3(q, &c[o * 3]);     // ref: features/features_shot.cpp:77
        }
    return 0;
}

void ismref_rgb2lab(uint32_t rgba, float* L, float* a, float* b) {
    unsigned char r, g, bb; unpack_rgb(rgba, r, g, bb);
    rgb2lab(r, g, bb, *L, *a, *b);
}
int ismref_pair_features(const float* p1, const float* n1, const float* p2, const float* n2, float* f4_out) {
    return pair_features(p1, n1, p2, n2, f4_out[0], f4_out[1], f4_out[2], f4_out[3]) ? 1 : 0;
}

float ismref_distance(int metric, int dim, const float* a, const float* b) { return dist_any(metric, a, b, dim); }

/* exact k-NN = flann::Index::knnSearch with SearchParams(-1) (ref: activation_strategy/activation_strategy_knn.h:57-72).
 * Ascending distance, ties -> lowest row. */
int ismref_knn(int metric, int n_words, int dim, const float* words, int nq, const float* q, int k,
               int32_t* idx_out, float* dist_out) {
    if (k <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < nq; ++i) {
        std::vector<std::pair<float, int>> best;   // sorted ascending, size <= k
        const float* qi = q + static_cast<size_t>(i) * dim;
        for (int c = 0; c < n_words; ++c) {
            const float d = dist_any(metric, qi, words + static_cast<size_t>(c) * dim, dim);
            if (static_cast<int>(best.size()) < k || d < best.back().first) {
                auto pos = std::upper_bound(best.begin(), best.end(), std::make_pair(d, c));
                best.insert(pos, std::make_pair(d, c));
                if (static_cast<int>(best.size()) > k) best.pop_back();
            }
        }
        for (int j = 0; j < k; ++j) {
            idx_out[static_cast<size_t>(i) * k + j] = j < static_cast<int>(best.size()) ? best[j].second : -1;
            dist_out[static_cast<size_t>(i) * k + j] = j < static_cast<int>(best.size()) ? best[j].first : kNaN;
        }
    }
    return 0;
}
int ismref_knn_ratio(int metric, int n_words, int dim, const float* words, int nq, const float* q,
                     float thr, int32_t* idx_out, float* dist_out) {
    std::vector<int32_t> idx(static_cast<size_t>(nq) * 2); std::vector<float> d(static_cast<size_t>(nq) * 2);
    ismref_knn(metric, n_words, dim, words, nq, q, 2, idx.data(), d.data());
    for (int i = 0; i < nq; ++i) {
        idx_out[i] = idx[i * 2]; dist_out[i] = d[i * 2];
        if (idx[i * 2 + 1] >= 0 && d[i * 2] / d[i * 2 + 1] > thr) idx_out[i] = -1;   // :77-84
    }
    return 0;
}

int ismref_knn_rule(int metric, int n_words, int dim, const float* words, const uint32_t* word_class, int nq, const float* q,
                    float thr, int32_t* idx_out, float* dist_out) {
    std::vector<int32_t> idx(static_cast<size_t>(nq) * 3); std::vector<float> d(static_cast<size_t>(nq) * 3);
    ismref_knn(metric, n_words, dim, words, nq, q, 3, idx.data(), d.data());
    for (int i = 0; i < nq; ++i) {
        const int i0 = idx[i * 3], i1 = idx[i * 3 + 1], i2 = idx[i * 3 + 2];
        const float a = d[i * 3], b = d[i * 3 + 1], c = d[i * 3 + 2];
        idx_out[i] = -1; dist_out[i] = kNaN;
        if (i2 < 0) { if (i0 >= 0) { idx_out[i] = i0; dist_out[i] = a; } continue; }
        const uint32_t c0 = word_class[i0], c1 = word_class[i1], c2 = word_class[i2];
        if (c0 == c1 && c0 == c2) { idx_out[i] = i0; dist_out[i] = a; }
        else if (c0 == c1 && c0 != c2) { if (a / c < thr) { idx_out[i] = i0; dist_out[i] = a; } }
        else if (c0 != c1 && c1 == c2) { if (a / b >= thr) { idx_out[i] = i1; dist_out[i] = b; } }
        else if (c0 != c1 && c1 != c2) { if (a / b < thr) { idx_out[i] = i0; dist_out[i] = a; } }
    }
    return 0;
}

void ismref_rot_quaternion(const float* lrf9, float* out) { Quat q = rot_quaternion(lrf9); out[0] = q.w; out[1] = q.x; out[2] = q.y; out[3] = q.z; }
void ismref_rotate_into(const float* lrf9, const float* v, float* out) {
    Quat q = rot_quaternion(lrf9); out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; quat_rotate(q, out);
}
void ismref_rotate_back(const float* lrf9, const float* v, float* out) {
    Quat q = rot_quaternion(lrf9); out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; quat_rotate_inv(q, out);
}

int ismref_cast_votes(int n_words, int dim, const float* word_weight,
                      const uint32_t* vote_offsets, const float* vote_xyz, const float* vote_weight,
                      const float* vote_class_weight, const uint32_t* vote_class, const uint32_t* vote_instance,
                      const float* vote_bbox_quat, const float* vote_bbox_size,
                      int n_classes, const float* class_sigma, uint32_t flags,
                      int nq, const float* lrf9, const float* kpx, const float* kpy, const float* kpz,
                      int k, const int32_t* idx, const float* dist,
                      float* pos_out, float* w_out, int32_t* cls_out, int32_t* inst_out, int32_t* cw_out,
                      float* bq_out, float* bs_out) {
    (void)dim;
    int maxv = 0;
    for (int c = 0; c < n_words; ++c) maxv = std::max(maxv, static_cast<int>(vote_offsets[c + 1] - vote_offsets[c]));
    const size_t n_slots = static_cast<size_t>(nq) * k * maxv;
    for (size_t s = 0; s < n_slots; ++s) {
        cls_out[s] = -1; inst_out[s] = -1; cw_out[s] = -1; w_out[s] = 0.f;
        pos_out[s * 3] = pos_out[s * 3 + 1] = pos_out[s * 3 + 2] = 0.f;
        if (bq_out) { bq_out[s * 4] = 1.f; bq_out[s * 4 + 1] = bq_out[s * 4 + 2] = bq_out[s * 4 + 3] = 0.f; }
        if (bs_out) { bs_out[s * 3] = bs_out[s * 3 + 1] = bs_out[s * 3 + 2] = 0.f; }
    }
    for (int f = 0; f < nq; ++f)
        for (int j = 0; j < k; ++j) {
            const int c = idx[static_cast<size_t>(f) * k + j];
            if (c < 0 || c >= n_words) continue;
            const float d = dist[static_cast<size_t>(f) * k + j];               // codeword_distribution.cpp:87
            const Quat rq = rot_quaternion(lrf9 + static_cast<size_t>(f) * 9);
            for (uint32_t v = vote_offsets[c]; v < vote_offsets[c + 1]; ++v) {
                const size_t s = (static_cast<size_t>(f) * k + j) * maxv + (v - vote_offsets[c]);
                const uint32_t classId = vote_class[v];
                const float classWeight = vote_class_weight ? vote_class_weight[v] : 1.0f;
                const float classSigma = (static_cast<int>(classId) < n_classes) ? class_sigma[classId] : 1.0f;   // :108-117
                const float matching = gauss_dist(classSigma, d);
                const float voteWeight = vote_weight ? vote_weight[v] : 1.0f;
                float weight = 1.0f;
                weight = (flags & 1u) ? weight * classWeight : weight;
                weight = (flags & 2u) ? weight * voteWeight : weight;
                weight = (flags & 4u) ? weight * matching : weight;
                weight = (flags & 8u) ? weight * (word_weight ? word_weight[c] : 1.0f) : weight;
                if (std::fabs(d) > 2 * classSigma) continue;                        // :131
                if (weight < std::numeric_limits<float>::epsilon()) continue;       // :137
                float rb[3] = {vote_xyz[v * 3], vote_xyz[v * 3 + 1], vote_xyz[v * 3 + 2]};
                quat_rotate_inv(rq, rb);                                            // rotateBack :157
                pos_out[s * 3 + 0] = kpx[f] + rb[0]; pos_out[s * 3 + 1] = kpy[f] + rb[1]; pos_out[s * 3 + 2] = kpz[f] + rb[2];
                w_out[s] = weight; cls_out[s] = static_cast<int32_t>(classId);
                inst_out[s] = static_cast<int32_t>(vote_instance[v]); cw_out[s] = c;
                if (bq_out) {
                    Quat b = vote_bbox_quat ? Quat{vote_bbox_quat[v * 4], vote_bbox_quat[v * 4 + 1], vote_bbox_quat[v * 4 + 2], vote_bbox_quat[v * 4 + 3]}
                                            : Quat{1, 0, 0, 0};
                    Quat r = qmul(b, rq);                                           // :162
                    bq_out[s * 4] = r.w; bq_out[s * 4 + 1] = r.x; bq_out[s * 4 + 2] = r.y; bq_out[s * 4 + 3] = r.z;
                }
                if (bs_out && vote_bbox_size) { bs_out[s * 3] = vote_bbox_size[v * 3]; bs_out[s * 3 + 1] = vote_bbox_size[v * 3 + 1]; bs_out[s * 3 + 2] = vote_bbox_size[v * 3 + 2]; }
            }
        }
    return 0;
}

int ismref_create_seeds(int n, const float* pos, const float* w, float bin_size, int cap, float* sp, float* sw) {
    std::vector<MSVote> votes(n), seeds;
    for (int i = 0; i < n; ++i) { votes[i].p[0] = pos[i * 3]; votes[i].p[1] = pos[i * 3 + 1]; votes[i].p[2] = pos[i * 3 + 2]; votes[i].w = w[i]; }
    create_seeds(votes, bin_size, seeds);
    for (int i = 0; i < static_cast<int>(seeds.size()) && i < cap; ++i) {
        sp[i * 3] = seeds[i].p[0]; sp[i * 3 + 1] = seeds[i].p[1]; sp[i * 3 + 2] = seeds[i].p[2]; sw[i] = seeds[i].w;
    }
    return static_cast<int>(seeds.size());
}

/* Voting::findMaxima + VotingMeanShift::iFindMaxima (ref: voting/voting.cpp:79-328, voting_mean_shift.cpp:39-177).
 * Votes of one class are taken in slot order (the reference's order is nondeterministic, voting.cpp:73-76).
 * Out of scope here as in the product: RANSAC vote filter, global features. */
int ismref_find_maxima(int n_obj, const uint32_t* so, const float* vpos, const float* vw, const int32_t* vcls,
                       const int32_t* vinst, const float* vbs, const ismref_maxima_params* P,
                       int32_t* n_max_out, float* mpos, float* mw, int32_t* mcls, int32_t* minst, float* miw,
                       float* mbs, int32_t* mnv, float* class_score) {
    const int C = P->n_classes, cap = P->max_maxima;
#pragma omp parallel for schedule(dynamic, 1)
    for (int o = 0; o < n_obj; ++o) {
        std::vector<Maximum> maxima;
        std::vector<std::pair<float, int>> nb;
        for (int c = 0; c < C; ++c) {
            std::vector<MSVote> votes;
            std::vector<V3> vbbox;
            for (uint32_t s = so[o]; s < so[o + 1]; ++s) {
                if (vcls[s] != c) continue;
                MSVote v; v.p[0] = vpos[s * 3]; v.p[1] = vpos[s * 3 + 1]; v.p[2] = vpos[s * 3 + 2];
                v.w = vw[s]; v.inst = vinst[s]; v.slot = static_cast<int>(s);
                votes.push_back(v);
                vbbox.push_back(vbs ? V3{vbs[s * 3], vbs[s * 3 + 1], vbs[s * 3 + 2]} : V3{0, 0, 0});
            }
            if (votes.empty()) continue;            // class not present in m_votes
            float h = P->class_bandwidth ? P->class_bandwidth[c] : P->bandwidth;   // :48-49
            if (P->single_object_max_type != 0) {
                // voting_mean_shift.cpp:124-157: one maximum at the cloud centroid; bandwidth = the class's search distance |
                // SingleObjectHelper::getModelRadius | getVotingSpaceSize (single_object_mode_helper.cpp:15-40)
                const float* q = P->object_centroid + static_cast<size_t>(o) * 3;
                if (P->single_object_max_type == 2) h = P->object_radius[o];
                else if (P->single_object_max_type == 3) {
                    float md = 0;
                    for (auto& v : votes) { const float dx = v.p[0] - q[0], dy = v.p[1] - q[1], dz = v.p[2] - q[2]; const float d = dx * dx + dy * dy + dz * dz; md = md > d ? md : d; }
                    h = std::sqrt(md);
                }
                const V3 pos = {q[0], q[1], q[2]};
                std::vector<int> cluster;
                (void)estimate_density(votes, pos.data(), h, P->kernel, true, &cluster, nb);
                append_maximum(votes, vbbox, cluster, pos, c, P->min_votes_threshold, maxima, P->vote_bbox_quat);
                continue;
            }
            std::vector<MSVote> seeds;
            create_seeds(votes, (h * 2.0f) / sqrtf(2), seeds);                              // :33-37, :83
            // iDoMeanShift :201-244
            std::vector<V3> centers;
            for (const MSVote& seed : seeds) {
                float cur[3] = {seed.p[0], seed.p[1], seed.p[2]};
                int iter = 0; float diff = 0; bool skip = false;
                do {
                    float shifted[3];
                    if (!compute_mean_shift(votes, cur, shifted, h, P->kernel, nb)) { skip = true; break; }
                    diff = norm3(cur, shifted);
                    cur[0] = shifted[0]; cur[1] = shifted[1]; cur[2] = shifted[2];
                    iter++;
                } while (diff > P->threshold && iter <= P->max_iter);
                if (!skip) centers.push_back(V3{cur[0], cur[1], cur[2]});
            }
            std::vector<float> densities;
            for (auto& cc : centers) densities.push_back(estimate_density(votes, cc.data(), h, P->kernel, false, nullptr, nb));
            if (P->suppression == 0) {
                std::vector<V3> avg;
                average_neighbor_maxima(centers, h, avg, densities);
                densities.clear();
                for (auto& cc : avg) densities.push_back(estimate_density(votes, cc.data(), h, P->kernel, false, nullptr, nb));
                centers = avg;
            }
            std::vector<V3> positions;
            if (P->suppression == 0 || P->suppression == 1) suppress_neighbor_maxima(centers, densities, h, positions);
            // :158-176 (votes are re-weighted IN PLACE across the maxima of this class)
            for (auto& pos : positions) {
                std::vector<int> cluster;
                float density = estimate_density(votes, pos.data(), h, P->kernel, true, &cluster, nb);
                (void)density;
                append_maximum(votes, vbbox, cluster, pos, c, P->min_votes_threshold, maxima, P->vote_bbox_quat);
            }
        }
        if (P->max_filter == 1) filter_maxima_simple(maxima, P->bandwidth);
        if (P->max_filter == 2) filter_maxima_merge(maxima, P->class_bandwidth, P->bandwidth);
        finish_object(maxima, o, C, cap, P->min_threshold, P->best_k, n_max_out, mpos, mw, mcls, minst, miw, mbs, mnv, class_score, P->max_bbox_quat_out);
    }
    return 0;
}

/* Voting::findMaxima + VotingHough3D::iFindMaxima (ref: voting/voting_hough_3d.cpp:33-95) over pcl::recognition::HoughSpace3D
 * (EXTERNAL, PCL 1.10 recognition/cg/hough_3d; restated from SURVEY Appendix A.7 -- parity unpinned):
 *   bins: ceil((max - min) / bin) per axis, index = x + nx (y + ny z); accumulator in double
 *   vote   : the bin of floor((p - min) / bin); a vote outside the space is dropped
 *   voteInt: central bin c; per axis the central weight wc = 1 - |p - min - centre| / bin (centre = (2 c bin + bin) / 2, float)
 *            and the neighbour on the vote's side with 1 - wc; the (up to) 8 bins get weight * product, bins with product 0 or
 *            outside the space get nothing; every bin remembers its voters in vote order
 *   findMaxima(-rel): threshold = rel * max(H) (rel > 1: max(H)); a bin >= threshold with no STRICTLY greater 26-neighbour is a
 *            maximum; maxima in ascending bin index
 * then per maximum: position = sum(pos * w) / sum(w) over its voters (float, vote order) and the Voting::findMaxima block.
 * The reference makes the bins cubic with edge 2 * MaximaHandler::getSearchDistForClass(class) (= BinSize[0] for "Config"). */
int ismref_hough3d_maxima(int n_obj, const uint32_t* so, const float* vpos, const float* vw, const int32_t* vcls,
                          const int32_t* vinst, const float* vbs, const ismref_hough_params* P,
                          int32_t* n_max_out, float* mpos, float* mw, int32_t* mcls, int32_t* minst, float* miw,
                          float* mbs, int32_t* mnv, float* class_score) {
    const int C = P->n_classes, cap = P->max_maxima;
#pragma omp parallel for schedule(dynamic, 1)
    for (int o = 0; o < n_obj; ++o) {
        std::vector<Maximum> maxima;
        for (int c = 0; c < C; ++c) {
            std::vector<MSVote> votes;
            std::vector<V3> vbbox;
            for (uint32_t s = so[o]; s < so[o + 1]; ++s) {
                if (vcls[s] != c) continue;
                MSVote v; v.p[0] = vpos[s * 3]; v.p[1] = vpos[s * 3 + 1]; v.p[2] = vpos[s * 3 + 2];
                v.w = vw[s]; v.inst = vinst[s]; v.slot = static_cast<int>(s);
                votes.push_back(v);
                vbbox.push_back(vbs ? V3{vbs[s * 3], vbs[s * 3 + 1], vbs[s * 3 + 2]} : V3{0, 0, 0});
            }
            if (votes.empty()) continue;            // class not present in m_votes
            const double bin = static_cast<double>(P->class_bin ? P->class_bin[c] : P->bin_size);
            long long cnt[3], total = 1;
            for (int d = 0; d < 3; ++d) {
                cnt[d] = static_cast<long long>(std::ceil((static_cast<double>(P->max_coord[d]) - static_cast<double>(P->min_coord[d])) / bin));
                if (cnt[d] < 0) cnt[d] = 0;
                total *= cnt[d];
            }
            if (total <= 0) continue;
            // the space is sparse: (bin index -> value, voters); neighbours that were never voted for hold 0
            std::map<long long, std::pair<double, std::vector<int>>> H;
            for (size_t i = 0; i < votes.size(); ++i) {
                long long cc[3]; double diff[3]; float wc[3]; bool in = true;
                for (int d = 0; d < 3; ++d) {
                    const double rel = static_cast<double>(votes[i].p[d]) - static_cast<double>(P->min_coord[d]);
                    cc[d] = static_cast<long long>(std::floor(rel / bin));
                    if (cc[d] >= cnt[d] || cc[d] < 0) { in = false; break; }
                    const float centre = static_cast<float>((2.0 * static_cast<double>(cc[d]) * bin + bin) / 2.0);
                    diff[d] = rel - static_cast<double>(centre);
                    wc[d] = static_cast<float>(1.0 - std::fabs(diff[d]) / bin);
                }
                if (!in) continue;
                if (!P->use_interpolation) {
                    auto& e = H[cc[0] + cnt[0] * (cc[1] + cnt[1] * cc[2])];
                    e.first += static_cast<double>(votes[i].w); e.second.push_back(static_cast<int>(i));
                    continue;
                }
                for (int n = 0; n < 27; ++n) {
                    const int nb[3] = {n % 3 - 1, (n / 3) % 3 - 1, n / 9 - 1};
                    float iw = 1.0f; bool ok = true; long long idx = 0, stride = 1;
                    for (int d = 0; d < 3 && ok; ++d) {
                        const long long b = cc[d] + nb[d];
                        if (b < 0 || b >= cnt[d]) { ok = false; break; }
                        if (nb[d] == 0) iw *= wc[d];
                        else if ((diff[d] < 0 ? -1 : 1) == nb[d]) iw *= 1.0f - wc[d];
                        else ok = false;
                        idx += b * stride; stride *= cnt[d];
                    }
                    if (!ok || !(iw > 0.0f)) continue;
                    auto& e = H[idx];
                    e.first += static_cast<double>(votes[i].w) * static_cast<double>(iw); e.second.push_back(static_cast<int>(i));
                }
            }
            double hmax = std::numeric_limits<double>::min();
            for (auto& kv : H) if (kv.second.first > hmax) hmax = kv.second.first;
            const double rel = static_cast<double>(P->rel_threshold);
            const double thr = rel <= 1.0 ? rel * hmax : hmax;
            for (auto& kv : H) {                                   // std::map: ascending bin index
                const double v = kv.second.first;
                if (v < thr) continue;
                const long long i0 = kv.first % cnt[0], i1 = (kv.first / cnt[0]) % cnt[1], i2 = kv.first / (cnt[0] * cnt[1]);
                bool is_max = true;
                for (int n = 0; n < 27 && is_max; ++n) {
                    if (n == 13) continue;
                    const long long b0 = i0 + n % 3 - 1, b1 = i1 + (n / 3) % 3 - 1, b2 = i2 + n / 9 - 1;
                    if (b0 < 0 || b0 >= cnt[0] || b1 < 0 || b1 >= cnt[1] || b2 < 0 || b2 >= cnt[2]) continue;
                    auto it = H.find(b0 + cnt[0] * (b1 + cnt[1] * b2));
                    if (it != H.end() && it->second.first > v) is_max = false;
                }
                if (!is_max) continue;
                // voting_hough_3d.cpp:70-93: weighted cluster centre of the bin's voters
                float cx = 0, cy = 0, cz = 0, wsum = 0;
                for (int vi : kv.second.second) {
                    cx += votes[vi].p[0] * votes[vi].w; cy += votes[vi].p[1] * votes[vi].w; cz += votes[vi].p[2] * votes[vi].w;
                    wsum += votes[vi].w;
                }
                append_maximum(votes, vbbox, kv.second.second, V3{cx / wsum, cy / wsum, cz / wsum}, c, P->min_votes_threshold, maxima, P->vote_bbox_quat);
            }
        }
        if (P->max_filter == 1) filter_maxima_simple(maxima, P->bin_size / 2);
        if (P->max_filter == 2) filter_maxima_merge(maxima, nullptr, P->bin_size / 2);
        finish_object(maxima, o, C, cap, P->min_threshold, P->best_k, n_max_out, mpos, mw, mcls, minst, miw, mbs, mnv, class_score, P->max_bbox_quat_out);
    }
    return 0;
}

/* pcl::VoxelGrid (ref: keypoints/keypoints_voxel_grid.cpp:38-44) */
int ismref_voxel_grid(int n, const float* x, const float* y, const float* z, const uint32_t* rgba,
                      float leaf, int cap, float* kx, float* ky, float* kz, uint32_t* krgba) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) {
        if (!finite3(x[i], y[i], z[i])) continue;
        mn[0] = std::min(mn[0], x[i]); mx[0] = std::max(mx[0], x[i]);
        mn[1] = std::min(mn[1], y[i]); mx[1] = std::max(mx[1], y[i]);
        mn[2] = std::min(mn[2], z[i]); mx[2] = std::max(mx[2], z[i]);
    }
    if (mn[0] > mx[0]) return 0;
    const float inv = 1.0f / leaf;
    int minb[3], maxb[3], divb[3];
    for (int a = 0; a < 3; ++a) {
        minb[a] = static_cast<int>(std::floor(mn[a] * inv));
        maxb[a] = static_cast<int>(std::floor(mx[a] * inv));
        divb[a] = maxb[a] - minb[a] + 1;
    }
    const int64_t mul[3] = {1, divb[0], static_cast<int64_t>(divb[0]) * divb[1]};
    std::vector<std::pair<int64_t, int>> iv;
    iv.reserve(n);
    for (int i = 0; i < n; ++i) {
        if (!finite3(x[i], y[i], z[i])) continue;
        int64_t i0 = static_cast<int64_t>(std::floor(x[i] * inv) - static_cast<float>(minb[0]));
        int64_t i1 = static_cast<int64_t>(std::floor(y[i] * inv) - static_cast<float>(minb[1]));
        int64_t i2 = static_cast<int64_t>(std::floor(z[i] * inv) - static_cast<float>(minb[2]));
        iv.emplace_back(i0 * mul[0] + i1 * mul[1] + i2 * mul[2], i);
    }
    std::stable_sort(iv.begin(), iv.end(), [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) { return a.first < b.first; });
    int out = 0;
    size_t i = 0;
    while (i < iv.size()) {
        size_t j = i;
        float s[3] = {0, 0, 0}, c[3] = {0, 0, 0};
        while (j < iv.size() && iv[j].first == iv[i].first) {
            const int p = iv[j].second;
            s[0] += x[p]; s[1] += y[p]; s[2] += z[p];
            if (rgba) { c[0] += static_cast<float>((rgba[p] >> 16) & 0xff); c[1] += static_cast<float>((rgba[p] >> 8) & 0xff); c[2] += static_cast<float>(rgba[p] & 0xff); }
            ++j;
        }
        const float cnt = static_cast<float>(j - i);
        if (out < cap) {
            kx[out] = s[0] / cnt; ky[out] = s[1] / cnt; kz[out] = s[2] / cnt;
            if (krgba) krgba[out] = rgba ? ((static_cast<uint32_t>(static_cast<uint8_t>(c[0] / cnt)) << 16) |
                                            (static_cast<uint32_t>(static_cast<uint8_t>(c[1] / cnt)) << 8) |
                                            static_cast<uint32_t>(static_cast<uint8_t>(c[2] / cnt))) : 0u;
        }
        ++out;
        i = j;
    }
    return out;
}

/* per-class sigma of Codebook::activate step 1 (ref: codebook/codebook.cpp:94-193). Features are visited
 * class by class (std::map order), model by model, in input order within (class, model). */
int ismref_class_sigmas(int metric, int dim, int n_feat, const float* feats, const uint32_t* feat_class,
                        const uint32_t* feat_model, const int32_t* activated_word, int n_words, const float* words,
                        int n_classes, float* sigma_out) {
    for (int c = 0; c < n_classes; ++c) {
        std::vector<int> ids;
        for (int i = 0; i < n_feat; ++i) if (static_cast<int>(feat_class[i]) == c) ids.push_back(i);
        if (ids.empty()) { sigma_out[c] = kNaN; continue; }
        std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return feat_model[a] < feat_model[b]; });
        const int max_elements = static_cast<int>(std::sqrt(static_cast<double>(ids.size())));   // :107
        std::vector<int> allFeat, allWords;
        size_t i = 0;
        while (i < ids.size()) {
            size_t j = i;
            while (j < ids.size() && feat_model[ids[j]] == feat_model[ids[i]]) {
                const int w = activated_word[ids[j]];
                if (static_cast<int>(allWords.size()) < max_elements && w >= 0 && w < n_words) allWords.push_back(w);   // :151-152
                ++j;
            }
            if (static_cast<int>(allFeat.size()) < max_elements) for (size_t t = i; t < j; ++t) allFeat.push_back(ids[t]);   // :154-155
            i = j;
        }
        float sum = 0; std::vector<float> distances;
        for (int f : allFeat) for (int w : allWords) {
            float d = dist_any(metric, feats + static_cast<size_t>(f) * dim, words + static_cast<size_t>(w) * dim, dim);
            sum += d; distances.push_back(d);
        }
        const int num = static_cast<int>(allFeat.size() * allWords.size());
        const float mean = sum / num;
        float variance = 0;
        for (float d : distances) { float diff = d - mean; variance += diff * diff; }
        variance /= num - 1;
        sigma_out[c] = variance;
    }
    return 0;
}


/* Codebook::activate (codebook/codebook.cpp:64-368) with one codeword per training feature (clustering_none.cpp:25-35):
 *   step 1  every feature (classes ascending, models in order, features in order) activates its k nearest codewords (exact,
 *           ties -> lowest row; KNNRule trains with k = 1); every activation appends a vote = rotateInto(centre - keypoint, LRF)
 *           to the codeword's distribution (codeword_distribution.cpp:37-71); class sigma^2 = sample variance of the distances
 *           between the first >= sqrt(n_c) features (whole models) and the first >= sqrt(n_c) activated codewords (:94-193)
 *   clean-up (KNN with K = 1 only, :201-224): keep distributions with exactly one vote
 *   step 2  CodewordDistribution::computeWeights (codeword_distribution.cpp:169-243): per vote the MEDIAN over the activating
 *           features j of exp(-|keyPos_j + rotateBack(vote_i, LRF_j) - modelCentre_i|^2 / 0.25)
 *   steps 3-9 statistical class weights term1[c] * term2[word] * term3[c]; m_term3 is keyed by class only, so the value of the
 *           LAST distribution (largest codeword id) holding class c is the one every word uses (:325-339) -- reproduced.
 * Outputs: kept codewords in ascending id (word_src = its training feature), votes as CSR (vote_feature = activating feature). */
int ismref_activate(int metric, int dim, int n, const float* feats, const float* lrf9, const float* kx, const float* ky, const float* kz,
                    const uint32_t* feat_class, const uint32_t* feat_model, const float* feat_center,
                    int n_codewords, const float* codewords,
                    int k, int clean_up, int n_classes,
                    int32_t* n_words_out, uint32_t* word_src, uint32_t* vote_off, uint32_t* vote_feature, float* vote_xyz,
                    float* vote_weight, float* vote_class_weight, float* class_sigma) {
    if (n <= 0 || k <= 0) return -1;
    if (!codewords) { codewords = feats; n_codewords = n; }     // clustering_none.cpp:25-35; else the cluster centres (implicit_shape_model.cpp:445-475)
    if (n_codewords < k) k = n_codewords;                       // FLANN returns as many neighbours as the index has rows
    // exact kNN of every feature in the codebook
    std::vector<int32_t> act(static_cast<size_t>(n) * k, -1);
    std::vector<float> actd(static_cast<size_t>(n) * k, 0.f);
    ismref_knn(metric, n_codewords, dim, codewords, n, feats, k, act.data(), actd.data());
    // iteration order of the reference: classes ascending, models as they come, features as they come
    std::vector<int> order;
    for (int c = 0; c < n_classes; ++c) for (int i = 0; i < n; ++i) if (static_cast<int>(feat_class[i]) == c) order.push_back(i);
    std::vector<std::vector<int>> dist(n_codewords);          // per codeword: activating features in activation order
    for (int c = 0; c < n_classes; ++c) {
        class_sigma[c] = std::nanf("");
        std::vector<int> ids;
        for (int i = 0; i < n; ++i) if (static_cast<int>(feat_class[i]) == c) ids.push_back(i);
        if (ids.empty()) continue;
        const int max_elements = static_cast<int>(std::sqrt(static_cast<double>(ids.size())));
        std::vector<int> allFeat, allWords;
        size_t i = 0;
        while (i < ids.size()) {
            size_t j = i;
            while (j < ids.size() && feat_model[ids[j]] == feat_model[ids[i]]) {
                std::vector<int> activated;
                for (int t = 0; t < k; ++t) { const int w = act[static_cast<size_t>(ids[j]) * k + t]; if (w >= 0) activated.push_back(w); }
                for (int w : activated) dist[w].push_back(ids[j]);
                if (static_cast<int>(allWords.size()) < max_elements) allWords.insert(allWords.end(), activated.begin(), activated.end());
                ++j;
            }
            if (static_cast<int>(allFeat.size()) < max_elements) for (size_t t = i; t < j; ++t) allFeat.push_back(ids[t]);
            i = j;
        }
        float sum = 0; std::vector<float> ds;
        for (int fi : allFeat) for (int w : allWords) {
            const float d = dist_any(metric, feats + static_cast<size_t>(fi) * dim, codewords + static_cast<size_t>(w) * dim, dim);
            sum += d; ds.push_back(d);
        }
        const int num = static_cast<int>(allFeat.size() * allWords.size());
        const float mean = sum / num;
        float variance = 0;
        for (float d : ds) { const float diff = d - mean; variance += diff * diff; }
        variance /= num - 1;
        class_sigma[c] = variance;
    }
    // clean-up + CSR in ascending codeword id
    std::vector<int> kept;
    for (int w = 0; w < n_codewords; ++w) { if (dist[w].empty()) continue; if (clean_up && dist[w].size() != 1) continue; kept.push_back(w); }
    *n_words_out = static_cast<int32_t>(kept.size());
    vote_off[0] = 0;
    std::vector<int> vote_word;
    size_t nv = 0;
    for (size_t e = 0; e < kept.size(); ++e) {
        const int w = kept[e];
        word_src[e] = static_cast<uint32_t>(w);
        for (int fi : dist[w]) {
            float v[3] = {feat_center[fi * 3] - kx[fi], feat_center[fi * 3 + 1] - ky[fi], feat_center[fi * 3 + 2] - kz[fi]};
            quat_rotate(rot_quaternion(lrf9 + static_cast<size_t>(fi) * 9), v);          // rotateInto
            vote_xyz[nv * 3] = v[0]; vote_xyz[nv * 3 + 1] = v[1]; vote_xyz[nv * 3 + 2] = v[2];
            vote_feature[nv] = static_cast<uint32_t>(fi);
            vote_word.push_back(static_cast<int>(e));
            ++nv;
        }
        vote_off[e + 1] = static_cast<uint32_t>(nv);
    }
    // step 2: computeWeights
    const float sigma = 0.5f;
    for (size_t e = 0; e < kept.size(); ++e) {
        const uint32_t v0 = vote_off[e], v1 = vote_off[e + 1];
        for (uint32_t vi = v0; vi < v1; ++vi) {
            std::vector<float> list;
            const int fi_i = static_cast<int>(vote_feature[vi]);
            for (uint32_t vj = v0; vj < v1; ++vj) {
                const int fj = static_cast<int>(vote_feature[vj]);
                float r[3] = {vote_xyz[vi * 3], vote_xyz[vi * 3 + 1], vote_xyz[vi * 3 + 2]};
                quat_rotate_inv(rot_quaternion(lrf9 + static_cast<size_t>(fj) * 9), r);  // rotateBack
                const float cx = kx[fj] + r[0], cy = ky[fj] + r[1], cz = kz[fj] + r[2];
                const float dx = cx - feat_center[fi_i * 3], dy = cy - feat_center[fi_i * 3 + 1], dz = cz - feat_center[fi_i * 3 + 2];
                const float d = std::sqrt(dx * dx + dy * dy + dz * dz);
                list.push_back(static_cast<float>(std::exp(static_cast<double>((-1 * (d * d)) / (sigma * sigma)))));
            }
            std::sort(list.begin(), list.end());
            vote_weight[vi] = list.size() % 2 == 0 ? (list[list.size() / 2 - 1] + list[list.size() / 2]) / 2 : list[list.size() / 2];
        }
    }
    // steps 3-9
    std::vector<std::map<int, int>> votesForClass(n_classes);    // class -> (word -> votes)
    std::vector<int> numFeatures(n_classes, 0);
    for (size_t e = 0; e < kept.size(); ++e)
        for (uint32_t vi = vote_off[e]; vi < vote_off[e + 1]; ++vi) { const int c = static_cast<int>(feat_class[vote_feature[vi]]); votesForClass[c][static_cast<int>(e)]++; numFeatures[c]++; }
    std::map<int, float> sum;
    for (int c = 0; c < n_classes; ++c)
        for (auto& kv : votesForClass[c]) {
            if (sum.find(kv.first) == sum.end()) sum[kv.first] = kv.second / static_cast<float>(numFeatures[c]);
            else sum[kv.first] += kv.second / static_cast<float>(numFeatures[c]);
        }
    std::vector<float> term1(n_classes, 0.f), term3(n_classes, 0.f);
    for (int c = 0; c < n_classes; ++c) if (!votesForClass[c].empty()) term1[c] = 1.0f / static_cast<float>(votesForClass[c].size());
    for (size_t e = 0; e < kept.size(); ++e) {                   // ascending codeword id: later words overwrite term3[c]
        std::vector<int> classes;
        for (uint32_t vi = vote_off[e]; vi < vote_off[e + 1]; ++vi) { const int c = static_cast<int>(feat_class[vote_feature[vi]]); if (std::find(classes.begin(), classes.end(), c) == classes.end()) classes.push_back(c); }
        for (int c : classes) term3[c] = (votesForClass[c][static_cast<int>(e)] / static_cast<float>(numFeatures[c])) / sum[static_cast<int>(e)];
    }
    for (size_t e = 0; e < kept.size(); ++e) {
        const float term2 = 1.0f / static_cast<float>(vote_off[e + 1] - vote_off[e]);
        for (uint32_t vi = vote_off[e]; vi < vote_off[e + 1]; ++vi) { const int c = static_cast<int>(feat_class[vote_feature[vi]]); vote_class_weight[vi] = term1[c] * term2 * term3[c]; }
    }
    (void)order;
    return 0;
}


/* ClusteringKMeans::cluster (clustering/clustering_kmeans.h:53-131) = one level of FLANN 1.9.1's k-means (EXTERNAL: kmeans_index.h
 * computeClustering, center_chooser.h), followed by the nearest centre of every feature. Restated with the SAME build-defined
 * choices as csrc/kmeans.hip (randomness = splitmix64(seed + draw), integer k-means++ potentials, integer means, float centres,
 * exact functor distances with ties to the lowest row): written independently from that kernel file, sequential and plain.
 * centers_out[n_clusters * dim], assign_out[n], dist_out[n] or NULL. Returns 0; *n_clusters_out <= n_clusters. */
namespace {
uint64_t km_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t km_mulhi(uint64_t a, uint64_t b) { return static_cast<uint64_t>((static_cast<unsigned __int128>(a) * b) >> 64); }
}  // namespace
int ismref_kmeans(int metric, int n, int dim, const float* feats, int n_clusters, int max_iterations, int centers_init, uint64_t seed,
                  float* centers_out, int32_t* assign_out, float* dist_out, int32_t* n_clusters_out, int32_t* iterations_out) {
    if (n <= 0 || dim <= 0 || n_clusters <= 0) return -1;
    const int k = std::min(n_clusters, n);
    auto row = [&](int i) { return feats + static_cast<size_t>(i) * dim; };
    // ---- centre choosers
    std::vector<int> chosen;
    std::vector<float> closest(n, 3.3895314e38f);             // bit pattern 0x7f7f7f7f, as the device initialises it
    auto update = [&](int c) { for (int i = 0; i < n; ++i) { const float d = dist_any(metric, row(i), row(c), dim); if (d < closest[i]) closest[i] = d; } };
    if (centers_init == 0) {                                  // RANDOM: seeded permutation, skip points that coincide with a centre
        std::vector<uint32_t> p(n); std::iota(p.begin(), p.end(), 0u);
        for (int i = n - 1; i > 0; --i) std::swap(p[i], p[km_mix(seed + static_cast<uint64_t>(n - i)) % static_cast<uint64_t>(i + 1)]);
        chosen.push_back(static_cast<int>(p[0]));
        size_t cur = 1;
        while (static_cast<int>(chosen.size()) < k) {
            update(chosen.back());
            while (cur < p.size() && closest[p[cur]] < 1e-16f) ++cur;
            if (cur >= p.size()) break;
            chosen.push_back(static_cast<int>(p[cur++]));
        }
    } else {
        chosen.push_back(static_cast<int>(km_mix(seed) % static_cast<uint64_t>(n)));
        float dmax0 = 0.f;
        for (int step = 1; step < k; ++step) {
            update(chosen.back());
            if (step == 1) for (int i = 0; i < n; ++i) if (closest[i] == closest[i] && closest[i] > dmax0) dmax0 = closest[i];
            if (centers_init == 1) {                          // GONZALES: the farthest point, lowest index on ties, distance > 0
                int best = -1; float bv = 0.f;
                for (int i = 0; i < n; ++i) if (closest[i] > bv) { bv = closest[i]; best = i; }
                if (best < 0) break;
                chosen.push_back(best);
            } else {                                          // KMEANSPP, one local try, integer potentials
                auto pot = [&](float d) -> uint64_t {
                    if (!(d > 0.f) || !(dmax0 > 0.f)) return 0;
                    return static_cast<uint64_t>(static_cast<double>(d) / static_cast<double>(dmax0) * static_cast<double>(1ull << 40));
                };
                uint64_t total = 0;
                for (int i = 0; i < n; ++i) total += pot(closest[i]);
                if (total == 0) break;
                const uint64_t r = km_mulhi(km_mix(seed + static_cast<uint64_t>(step)), total);
                uint64_t run = 0; int pick = -1;
                for (int i = 0; i < n; ++i) { const uint64_t q = pot(closest[i]); if (q && run + q > r) { pick = i; break; } run += q; }
                if (pick < 0) break;
                chosen.push_back(pick);
            }
        }
    }
    const int kc = static_cast<int>(chosen.size());
    for (int c = 0; c < kc; ++c) std::copy(row(chosen[c]), row(chosen[c]) + dim, centers_out + static_cast<size_t>(c) * dim);
    // ---- Lloyd
    float amax = 0.f;
    for (size_t t = 0; t < static_cast<size_t>(n) * dim; ++t) { const float a = std::fabs(feats[t]); if (a <= std::numeric_limits<float>::infinity() && a > amax) amax = a; }
    int e = 0;
    if (amax > 0.f && std::isfinite(amax)) std::frexp(amax, &e);
    const double scale = std::ldexp(1.0, 40 - e);
    std::vector<int32_t> belongs(n), nearest(n);
    std::vector<float> nd(n);
    ismref_knn(metric, kc, dim, centers_out, n, feats, 1, belongs.data(), nd.data());
    std::vector<long long> sums(static_cast<size_t>(kc) * dim);
    std::vector<uint32_t> counts(kc);
    int it = 0; bool converged = false;
    while (!converged && it < max_iterations) {
        ++it; converged = true;
        std::fill(sums.begin(), sums.end(), 0ll); std::fill(counts.begin(), counts.end(), 0u);
        for (int i = 0; i < n; ++i) {
            const int c = belongs[i];
            if (c < 0) continue;
            for (int j = 0; j < dim; ++j) { const double v = static_cast<double>(row(i)[j]) * scale; if (v == v) sums[static_cast<size_t>(c) * dim + j] += std::llrint(v); }
            counts[c]++;
        }
        for (int c = 0; c < kc; ++c) if (counts[c]) for (int j = 0; j < dim; ++j)
            centers_out[static_cast<size_t>(c) * dim + j] = static_cast<float>((static_cast<double>(sums[static_cast<size_t>(c) * dim + j]) / scale) / static_cast<double>(counts[c]));
        ismref_knn(metric, kc, dim, centers_out, n, feats, 1, nearest.data(), nd.data());
        for (int i = 0; i < n; ++i) if (nearest[i] >= 0 && nearest[i] != belongs[i]) {
            if (belongs[i] >= 0) counts[belongs[i]]--;
            counts[nearest[i]]++; belongs[i] = nearest[i]; converged = false;
        }
        if (n > kc) for (int i = 0; i < kc; ++i) if (counts[i] == 0) {
            int j = (i + 1) % kc;
            for (int t = 0; t < kc && counts[j] <= 1; ++t) j = (j + 1) % kc;
            if (counts[j] <= 1) continue;
            for (int p = 0; p < n; ++p) if (belongs[p] == j) { belongs[p] = i; counts[j]--; counts[i]++; break; }
            converged = false;
        }
    }
    std::vector<float> dtmp(n);
    ismref_knn(metric, kc, dim, centers_out, n, feats, 1, assign_out, dist_out ? dist_out : dtmp.data());
    *n_clusters_out = kc;
    if (iterations_out) *iterations_out = it;
    return 0;
}

}  // extern "C"
