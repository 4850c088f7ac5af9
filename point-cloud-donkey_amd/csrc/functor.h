// functor.h — the FLANN distance functors (utils/distance.cpp:33-52) evaluated by a whole wave, shared by the kNN re-rank /
// exact scan (knn.hip) and the training-side sigma computation (train.hip). Included inside an anonymous namespace.
#pragma once
#ifndef ISM_F32X4_DEFINED
#define ISM_F32X4_DEFINED
typedef float ism_f32x4 __attribute__((ext_vector_type(4)));
#endif
// The FLANN functors evaluated by a whole wave, bit-identical to the scalar loops above. L2: the functor adds one 4-element
// group sum ((d0^2 + d1^2) + d2^2) + d3^2 per step to the running result; the group sums are independent, so the lanes compute
// them from coalesced 16-byte loads and only the chain of additions (dim/4 of them, from LDS) stays sequential. chi2: the
// per-element terms are independent, the chain adds them one by one (a skipped term adds +0, which leaves the result unchanged).
// Every lane returns the distance. sT: per-wave scratch of dim floats.
__device__ __forceinline__ float wave_functor(int metric, const float* __restrict__ a, const float* __restrict__ b, int dim, int lane, float* sT) {
    const int n4 = dim >> 2;
    int n_terms;
    if (metric == ISMHIP_METRIC_CHI2) {
        for (int i = lane; i < dim; i += 64) {
            const float x = a[i], y = b[i], sum = x + y, diff = x - y;
            sT[i] = sum > 0 ? diff * diff / sum : 0.f;
        }
        n_terms = dim;
    } else {
        for (int g = lane; g < n4; g += 64) {
            const ism_f32x4 x = *(const ism_f32x4*)(a + 4 * g), y = *(const ism_f32x4*)(b + 4 * g);
            const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
            sT[g] = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        for (int i = 4 * n4 + lane; i < dim; i += 64) { const float d0 = a[i] - b[i]; sT[n4 + (i - 4 * n4)] = d0 * d0; }
        n_terms = n4 + (dim - 4 * n4);
    }
    float result = 0.f;                      // LDS traffic of one wave is ordered: the stores above are visible to the loads below
    int i = 0;
    for (; i + 3 < n_terms; i += 4) {
        const ism_f32x4 v = *(const ism_f32x4*)(sT + i);
        result += v[0]; result += v[1]; result += v[2]; result += v[3];
    }
    for (; i < n_terms; ++i) result += sT[i];
    return result;
}

