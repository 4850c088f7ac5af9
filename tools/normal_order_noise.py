"""How much of a PCA normal is decided by the ORDER in which PCL sums the neighbours?  ("PCA normals")
pcl::computeMeanAndCovarianceMatrix accumulates x*x .. z in float in one pass and forms cov = E[ab] - E[a]E[b]: with coordinates ~1 and a
5 cm neighbourhood the covariance (~1e-3) is the difference of sums rounded at ~6e-8 * sqrt(n). This script evaluates that arithmetic
for the same neighbour set in two different orders and prints the angle between the two smallest eigenvectors. CPU only, numpy."""
import numpy as np

rng = np.random.default_rng(0)


def normal(P):
    acc = np.zeros(9, np.float32)
    for p in P:
        acc += np.array([p[0] * p[0], p[0] * p[1], p[0] * p[2], p[1] * p[1], p[1] * p[2], p[2] * p[2], p[0], p[1], p[2]], np.float32)
    acc /= np.float32(len(P))
    C = np.array([[acc[0] - acc[6] * acc[6], acc[1] - acc[6] * acc[7], acc[2] - acc[6] * acc[8]],
                  [0, acc[3] - acc[7] * acc[7], acc[4] - acc[7] * acc[8]], [0, 0, acc[5] - acc[8] * acc[8]]], np.float32)
    C = C + np.triu(C, 1).T
    return np.linalg.eigh(C.astype(np.float64))[1][:, 0]      # the eigen-solve itself in double: only the accumulation differs


angles = []
for _ in range(2000):
    c = rng.uniform(-1, 1, 3)
    n = int(rng.integers(20, 200))
    u = rng.normal(size=3); u /= np.linalg.norm(u)
    a = np.cross(u, [1, 0, 0]); a /= np.linalg.norm(a); b = np.cross(u, a)
    r = 0.05 * np.sqrt(rng.random(n)); th = rng.random(n) * 2 * np.pi
    P = (c + np.outer(r * np.cos(th), a) + np.outer(r * np.sin(th), b) + np.outer(rng.normal(size=n) * 0.002, u)).astype(np.float32)
    angles.append(np.arccos(min(1.0, abs(normal(P) @ normal(P[rng.permutation(n)])))))
angles = np.array(angles)
print("angle between the two normals [rad]: median %.2e  p90 %.2e  p99 %.2e  p99.5 %.2e  max %.2e"
      % (np.median(angles), *np.percentile(angles, [90, 99, 99.5]), angles.max()))
