// eigen3.h — symmetric 3x3 eigen-decomposition in double (cyclic Jacobi), device side.
// Stands in for Eigen::SelfAdjointEigenSolver<Matrix3d> used by PCL's SHOT LRF
// (reference statement: third_party/pcl_shot_na_lrf/shot_na_lrf.hpp:95). Same sweep order, rotation
// formulas and stopping rule as the oracle's eigen_sym3 so both pick the same vectors.
#pragma once
#include <hip/hip_runtime.h>

// a: symmetric input (row-major 3x3, destroyed); w: ascending eigenvalues; V: eigenvectors as columns
__device__ inline void eigen_sym3(double a[3][3], double w[3], double V[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                const double apq = a[p][q];
                a[p][p] -= t * apq;
                a[q][q] += t * apq;
                a[p][q] = a[q][p] = 0.0;
                const int r = 3 - p - q;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double ev[3] = {a[0][0], a[1][1], a[2][2]};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 - i; ++j)
            if (ev[idx[j]] > ev[idx[j + 1]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
    double Vs[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = ev[idx[k]];
#pragma unroll
        for (int i = 0; i < 3; ++i) Vs[i][k] = V[i][idx[k]];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
}
