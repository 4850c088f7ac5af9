"""Generator of the hand-derived known-answer vectors (tests/golden/kat.json).

The reference ships no tests or fixtures and cannot be built or imported here, so these vectors are NOT
reference outputs: every expected value below is derived in closed form from the published algorithm
with plain numpy arithmetic in THIS script — it does not call the oracle or the HIP library. They pin the oracle; the HIP
path is then pinned against the oracle and against these same vectors.

    python tests/golden/make_golden.py        # rewrites tests/golden/kat.json
"""
import json
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
f32 = np.float32


def shot_sector_centres():
    """KAT 1: identity LRF, r = 1, five neighbours at sector centres with normals = +z.
    cos = 1 -> bin distance 10 -> step 10, all interpolation distances are 0 at a sector centre, so each neighbour puts
    weight 1 (cosine) + 1 (radial) + 1 (elevation) + 1 (azimuth) = 4 into slot 10 of its own sector; after the L2
    normalisation each of the five slots holds 1/sqrt(5)."""
    picks = [(0, 1, 1), (2, 0, 0), (3, 1, 0), (5, 0, 1), (7, 1, 1)]      # (azimuth sector, upper, outer)
    pts, expect = [], np.zeros(352)
    for sel, upper, outer in picks:
        az = -7 * np.pi / 8 + sel * np.pi / 4
        th = np.pi / 4 if upper else 3 * np.pi / 4
        d = 0.75 if outer else 0.25
        pts.append([d * np.sin(th) * np.cos(az), d * np.sin(th) * np.sin(az), d * np.cos(th)])
        sector = sel * 4 + (2 if outer else 0) + (1 if upper else 0)
        expect[sector * 11 + 10] = 1 / np.sqrt(5)
    return dict(points=pts, normals=[[0, 0, 1]] * 5, keypoint=[0, 0, 0], lrf=[1, 0, 0, 0, 1, 0, 0, 0, 1], radius=1.0,
                expected=expect.tolist(), tol=2e-6)


def lrf_paraboloid():
    """KAT 2: a cap z = -a (x^2 + y^2) sampled symmetrically in x and y on an ellipse (longer in x), plus extra points ON the
    plane z = 0 at x > 0 (y-mirrored). The weighted covariance is diagonal with xx > yy > zz, so the eigenvectors are the
    coordinate axes. z: every neighbour has v.z <= 0 for +z -> flipped to (0,0,-1). x: more neighbours at x > 0 -> (1,0,0).
    y = z x x = (0,-1,0)."""
    a = 0.3
    pts = []
    for ix in range(-6, 7):
        for iy in range(-3, 4):
            if ix == 0 or iy == 0:
                continue
            x, y = ix * 0.05, iy * 0.05
            pts.append([x, y, -a * (x * x + y * y)])
    for x in (0.11, 0.17, 0.23):
        for y in (-0.07, 0.07):
            pts.append([x, y, 0.0])
    return dict(points=pts, keypoint=[0, 0, 0], radius=0.5, expected=[1, 0, 0, 0, -1, 0, 0, 0, -1], tol=1e-6)


def rgb2lab_cases():
    """KAT 3: RGB -> CIELab through the LUT formulas (features/features_cshot.cpp:52-70, features_short_cshot.cpp:651-687),
    evaluated here directly in float32; the sXYZ index is clamped to 3999 (white reaches 4000, one past PCL's table)."""
    def lut_srgb(i):
        f = f32(i) / f32(255.0)
        return f32(np.power(f32((f + f32(0.055)) / f32(1.055)), f32(2.4))) if f > 0.04045 else f32(f / f32(12.92))

    def lut_xyz(v):
        i = min(3999, max(0, int(f32(v) * f32(4000))))
        f = f32(i) / f32(4000.0)
        return f32(np.power(f, f32(0.3333))) if f > 0.008856 else f32(7.787 * float(f) + 16.0 / 116.0)

    out = []
    for (r, g, b) in [(0, 0, 0), (255, 255, 255), (255, 0, 0), (12, 200, 99)]:
        fr, fg, fb = lut_srgb(r), lut_srgb(g), lut_srgb(b)
        x = f32(f32(fr * f32(0.412453)) + f32(fg * f32(0.357580))) + f32(fb * f32(0.180423))
        y = f32(f32(fr * f32(0.212671)) + f32(fg * f32(0.715160))) + f32(fb * f32(0.072169))
        z = f32(f32(fr * f32(0.019334)) + f32(fg * f32(0.119193))) + f32(fb * f32(0.950227))
        vx, vy, vz = lut_xyz(f32(x) / f32(0.95047)), lut_xyz(y), lut_xyz(f32(z) / f32(1.08883))
        L = min(100.0, float(f32(116.0) * vy - f32(16.0)))
        A = max(-120.0, min(120.0, float(f32(500.0) * f32(vx - vy))))
        B = max(-120.0, min(120.0, float(f32(200.0) * f32(vy - vz))))
        out.append(dict(rgba=(r << 16) | (g << 8) | b, L=L, a=A, b=B))
    return dict(cases=out, tol=2e-4)


def fpfh_two_points():
    """KAT 4: two surface points with orthogonal normals, p1 = 0 (n = +z), p2 = (d,0,0) (n = +y). Both pair features give
    f1 = atan2(0,0) = 0 -> bin 5, f2 = -1 -> bin 0, f3 = 0 -> bin 5, each SPFH holds 100 in those bins (hist_incr = 100/(2-1));
    any keypoint seeing both therefore gets FPFH bins (5, 11+0, 22+5) = 100."""
    exp = np.zeros(33); exp[5] = exp[11] = exp[27] = 100.0
    return dict(points=[[0, 0, 0], [0.2, 0, 0]], normals=[[0, 0, 1], [0, 1, 0]], keypoint=[0.1, 0.05, 0.0], radius=0.5,
                expected=exp.tolist(), tol=1e-4, pair=dict(f1=0.0, f2=-1.0, f3=0.0, f4=0.2))


def distances():
    """KAT 5: FLANN functors. L2 is SQUARED (no sqrt); chi-square skips terms with a+b = 0."""
    a, b = [1, 2, 3, 4, 0], [2, 2, 1, 0, 0]
    return dict(a=a, b=b, l2=21.0, chi2=1.0 / 3.0 + 0.0 + 4.0 / 4.0 + 16.0 / 4.0, tol=1e-6)


def rotations():
    """KAT 6: rows of the rotation are the LRF axes: rotateInto(v) = (x.v, y.v, z.v),
    rotateBack(u) = u0 x + u1 y + u2 z."""
    c, s = np.cos(0.7), np.sin(0.7)
    R1 = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
    c2, s2 = np.cos(-1.1), np.sin(-1.1)
    R2 = np.array([[1, 0, 0], [0, c2, s2], [0, -s2, c2]])
    R = (R2 @ R1)
    flip = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]]) @ R      # trace may be negative: exercises the other quaternion branch
    cases = []
    for M in (np.eye(3), R, flip, np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], float), np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]], float)):
        v = np.array([0.3, -1.2, 0.8])
        cases.append(dict(lrf=M.reshape(-1).tolist(), v=v.tolist(), into=(M @ v).tolist(), back=(M.T @ v).tolist()))
    return dict(cases=cases, tol=2e-6)


def seeds_order():
    """KAT 7a: createSeeds (voting_mean_shift.cpp:431-481): key = floor(p/bin + 0.5), one seed per cell at key*bin, cells
    iterated in (z, y, x) order, weight = sum of the cell's vote weights."""
    binsz = 1.0
    pos = [[0.1, 0.2, 2.1], [1.2, 0.1, 0.0], [0.0, 1.4, 0.1], [-0.2, 0.1, 0.1], [0.3, -0.4, 0.2], [1.1, -0.2, 0.1], [0.2, 0.1, 1.9]]
    w = [1, 2, 3, 4, 5, 6, 7]
    # cells: (0,0,2):{0,6} (1,0,0):{1,5} (0,1,0):{2} (0,0,0):{3,4}
    exp_pos = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 2]]       # z-major, then y, then x
    exp_w = [9, 8, 3, 8]
    return dict(pos=pos, w=w, bin=binsz, expected_pos=exp_pos, expected_w=exp_w)


def voxel_grid():
    """KAT 8: pcl::VoxelGrid centroids ordered by voxel linear index, x fastest."""
    pts = [[0.05, 0.05, 0.05], [0.07, 0.01, 0.03], [0.15, 0.05, 0.05], [0.05, 0.15, 0.05], [0.05, 0.05, 0.15], [0.16, 0.06, 0.04]]
    exp = [[0.06, 0.03, 0.04], [0.155, 0.055, 0.045], [0.05, 0.15, 0.05], [0.05, 0.05, 0.15]]
    return dict(points=pts, leaf=0.1, expected=exp, tol=1e-6)


def knn_ties():
    """KAT 9: exact kNN, ascending distance, ties go to the lowest row."""
    words = [[0, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0.5, 0.5, 0, 0]]
    q = [[1, 0, 0, 0], [0, 0, 1, 0]]
    # q0: rows 1 and 3 at distance 0 -> (1, 3); then row 4 at 0.5
    # q1: rows 0,1,2,3 all at distance 2, row 4 at 1.5 -> (4, 0, 1)
    return dict(words=words, q=q, k=3, expected_idx=[[1, 3, 4], [4, 0, 1]], expected_l2=[[0, 0, 0.5], [1.5, 2, 2]])


def main():
    kat = dict(shot_sector_centres=shot_sector_centres(), lrf_paraboloid=lrf_paraboloid(), rgb2lab=rgb2lab_cases(),
               fpfh_two_points=fpfh_two_points(), distances=distances(), rotations=rotations(), seeds_order=seeds_order(),
               voxel_grid=voxel_grid(), knn_ties=knn_ties())
    with open(OUT, "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
