"""Generator of the hand-derived known-answer vectors (tests/golden/kat.json).

The reference ships no tests or fixtures and cannot be built or imported here, so these vectors are NOT
reference outputs: every expected value below is derived in closed form from the published algorithm
with plain numpy arithmetic in THIS script — it does not call the oracle or the HIP library. They pin the oracle; the HIP
path is then pinned against the oracle and against these same vectors.

    python tests/golden/make_golden.py        # rewrites tests/golden/kat.json
"""
import json
import sys
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


# ---------------------------------------------------------------------------------------------------------------------------
# Round-2 vectors: the parts of the path a common-mode misreading could hide behind the sector-centre vector above.
# ---------------------------------------------------------------------------------------------------------------------------
def _lut_srgb(i):
    f = f32(i) / f32(255.0)
    return f32(np.power(f32((f + f32(0.055)) / f32(1.055)), f32(2.4))) if f > 0.04045 else f32(f / f32(12.92))


def _lut_xyz(v):
    i = min(3999, max(0, int(f32(v) * f32(4000))))
    f = f32(i) / f32(4000.0)
    return f32(np.power(f, f32(0.3333))) if f > 0.008856 else f32(7.787 * float(f) + 16.0 / 116.0)


def _lab_norm(rgb):
    """normalised CIELab (L/100, a/120, b/120) of an 8-bit RGB triple, float32 as the reference's LUT code (Appendix A.3)"""
    r, g, b = rgb
    fr, fg, fb = _lut_srgb(r), _lut_srgb(g), _lut_srgb(b)
    x = f32(f32(fr * f32(0.412453)) + f32(fg * f32(0.357580))) + f32(fb * f32(0.180423))
    y = f32(f32(fr * f32(0.212671)) + f32(fg * f32(0.715160))) + f32(fb * f32(0.072169))
    z = f32(f32(fr * f32(0.019334)) + f32(fg * f32(0.119193))) + f32(fb * f32(0.950227))
    vx, vy, vz = _lut_xyz(f32(x) / f32(0.95047)), _lut_xyz(y), _lut_xyz(f32(z) / f32(1.08883))
    L = min(f32(100.0), f32(f32(116.0) * vy - f32(16.0)))
    A = max(f32(-120.0), min(f32(120.0), f32(f32(500.0) * f32(vx - vy))))
    B = max(f32(-120.0), min(f32(120.0), f32(f32(200.0) * f32(vy - vz))))
    return f32(L / f32(100.0)), f32(A / f32(120.0)), f32(B / f32(120.0))


def _shot_one_neighbour(p, n, radius, colour=None):
    """Deposits of ONE neighbour p (float32 xyz relative to a keypoint at the origin, IDENTITY frame) with normal n, following
    SURVEY Appendix A.2 / A.3 term by term in float64. Returns ({bin: weight} for the 352 shape bins, {bin: weight} for the 992
    colour bins or None, parts) where parts names the four interpolation deposits so that the caller can assert they are all
    non-zero and land in distinct bins. colour = (rgb of the keypoint, rgb of the neighbour)."""
    p = np.asarray(p, np.float32).astype(np.float64); n = np.asarray(n, np.float32).astype(np.float64)
    r12, r14, r34 = radius / 2, radius / 4, radius * 3 / 4
    c = min(1.0, max(-1.0, float(n[2])))                       # n . z with z = (0,0,1)
    b = (1.0 + c) * 10 / 2
    d = float(np.sqrt(f32(p @ p)))
    xl, yl, zl = p
    bit4 = 1 if (yl > 0 or (yl == 0 and xl < 0)) else 0
    bit3 = (1 - bit4) if (xl > 0 or (xl == 0 and yl > 0)) else bit4
    s = ((bit4 << 3) + (bit3 << 2)) << 1
    if xl * yl > 0 or xl == 0:
        s += 0 if abs(xl) >= abs(yl) else 4
    else:
        s += 4 if abs(xl) > abs(yl) else 0
    s += 1 if zl > 0 else 0
    s += 2 if d > r12 else 0
    step = int(np.floor(b + 0.5)); b -= step
    shape, parts = {}, {}
    add = lambda h, k, v: h.__setitem__(k, h.get(k, 0.0) + v)
    wgt = 1 - abs(b)
    cos_bin = s * 11 + ((step + 1) % 10 if b > 0 else (step - 1 + 10) % 10)
    add(shape, cos_bin, abs(b)); parts["cosine"] = (cos_bin, abs(b))
    inc_w, nb = 0.0, []                                          # spatial increments shared by the shape and colour channels
    if d > r12:
        rd = (d - r34) / r12
        if d > r34: inc_w += 1 - rd
        else: inc_w += 1 + rd; nb.append(("radial", s - 2, -rd))
    else:
        rd = (d - r14) / r12
        if d < r14: inc_w += 1 + rd
        else: inc_w += 1 - rd; nb.append(("radial", s + 2, rd))
    inc = float(np.arccos(min(1.0, max(-1.0, zl / d))))
    if inc > np.pi / 2 or (abs(inc - np.pi / 2) < 1e-30 and zl <= 0):
        idd = (inc - 3 * np.pi / 4) / (np.pi / 2)
        if inc > 3 * np.pi / 4: inc_w += 1 - idd
        else: inc_w += 1 + idd; nb.append(("elevation", s + 1, -idd))
    else:
        idd = (inc - np.pi / 4) / (np.pi / 2)
        if inc < np.pi / 4: inc_w += 1 + idd
        else: inc_w += 1 - idd; nb.append(("elevation", s - 1, idd))
    if yl != 0 or xl != 0:
        az = float(np.arctan2(yl, xl)); sel = s >> 2
        ad = min(0.5, max(-0.5, (az - (-7 * np.pi / 8 + sel * np.pi / 4)) / (np.pi / 4)))
        if ad > 0: inc_w += 1 - ad; nb.append(("azimuth", (s + 4) % 32, ad))
        else: inc_w += 1 + ad; nb.append(("azimuth", (s - 4 + 32) % 32, -ad))
    for name, sec, w in nb:
        add(shape, sec * 11 + step, w); parts[name] = (sec * 11 + step, w)
    add(shape, s * 11 + step, wgt + inc_w); parts["main"] = (s * 11 + step, wgt + inc_w)
    col = None
    if colour is not None:
        (Lr, ar, br), (L, a, bb) = _lab_norm(colour[0]), _lab_norm(colour[1])
        cd = f32((f32(abs(f32(Lr - L))) + f32(f32(abs(f32(ar - a)) + abs(f32(br - bb))) / f32(2))) / f32(3))
        cd = min(1.0, max(0.0, float(cd)))
        bc = cd * 30
        step_c = int(np.floor(bc + 0.5)); bc -= step_c
        col = {}
        add(col, s * 31 + ((step_c + 1) % 30 if bc > 0 else (step_c - 1 + 30) % 30), abs(bc))
        for name, sec, w in nb:
            add(col, sec * 31 + step_c, w)
        add(col, s * 31 + step_c, (1 - abs(bc)) + inc_w)
        parts["colour"] = (step_c, bc)
    return shape, col, parts


def _sph(d, polar_deg, az_deg):
    t, a = np.radians(polar_deg), np.radians(az_deg)
    return [d * np.sin(t) * np.cos(a), d * np.sin(t) * np.sin(a), d * np.cos(t)]


def shot_off_centre():
    """KAT 10: ONE off-centre neighbour (x 5 copies so the >= 5-neighbour rule passes; the L2 normalisation cancels the factor):
    identity frame, r = 1, d = 0.6 (outer shell, below 3r/4 -> radial neighbour s-2), polar angle 60 deg (upper hemisphere,
    above 45 deg -> elevation neighbour s-1), azimuth 76.5 deg (sector 5, centre 67.5 deg -> ad = +0.2 -> azimuth neighbour
    s+4), n.z = 0.36 -> b = 6.8 -> step 7, offset -0.2 -> cosine neighbour step 6. All four interpolation deposits are non-zero
    and land in four distinct bins besides the main one. A second neighbour exercises the opposite branch of every ladder:
    inner shell above r/4, lower hemisphere below 135 deg, negative azimuth offset, positive cosine offset."""
    cases = []
    for p, c in ((_sph(0.6, 60.0, 76.5), 0.36), (_sph(0.4, 120.0, -140.0), -0.47)):
        n = [np.sqrt(1 - c * c), 0.0, c]
        shape, _, parts = _shot_one_neighbour(np.asarray(p, np.float32), np.asarray(n, np.float32), 1.0)
        bins = [parts[k][0] for k in ("main", "cosine", "radial", "elevation", "azimuth")]
        assert len(set(bins)) == 5 and all(parts[k][1] > 0.05 for k in ("cosine", "radial", "elevation", "azimuth")), parts
        v = np.zeros(352)
        for k, w in shape.items():
            v[k] = w
        cases.append(dict(point=np.asarray(p, np.float32).astype(float).tolist(), normal=np.asarray(n, np.float32).astype(float).tolist(),
                          copies=5, radius=1.0, parts={k: [int(b), float(w)] for k, (b, w) in parts.items()},
                          expected=(v / np.linalg.norm(v)).tolist()))
    assert cases[0]["parts"]["main"][0] == 23 * 11 + 7 and cases[0]["parts"]["azimuth"][0] == 27 * 11 + 7
    return dict(cases=cases, tol=3e-6)


def cshot_colour_pairs():
    """KAT 11: the colour channel of CSHOT-1344 (Appendix A.3) for two (keypoint rgb, neighbour rgb) pairs on the geometry of
    KAT 10's first neighbour: colour distance -> bin 30 cd, its own hard step + neighbour step (modulo 30), the SAME
    radial / elevation / azimuth increments and neighbour sectors as the shape channel, joint L2 norm over all 1344 values."""
    p = np.asarray(_sph(0.6, 60.0, 76.5), np.float32)
    n = np.asarray([np.sqrt(1 - 0.36 ** 2), 0.0, 0.36], np.float32)
    cases = []
    for ref, nb in (((200, 30, 30), (150, 90, 40)), ((12, 200, 99), (200, 240, 10))):      # steps 2 (+0.164) and 5 (-0.106)
        shape, col, parts = _shot_one_neighbour(p, n, 1.0, colour=(ref, nb))
        assert abs(parts["colour"][1]) > 0.05                     # a real colour-neighbour deposit
        v = np.zeros(1344)
        for k, w in shape.items():
            v[k] = w
        for k, w in col.items():
            v[352 + k] = w
        cases.append(dict(point=p.astype(float).tolist(), normal=n.astype(float).tolist(), copies=5, radius=1.0,
                          kp_rgba=(ref[0] << 16) | (ref[1] << 8) | ref[2], rgba=(nb[0] << 16) | (nb[1] << 8) | nb[2],
                          colour_step=int(parts["colour"][0]), colour_offset=float(parts["colour"][1]),
                          expected=(v / np.linalg.norm(v)).tolist()))
    assert cases[0]["colour_step"] != cases[1]["colour_step"]
    return dict(cases=cases, tol=3e-6)


def _ogre_quat(M):
    """Utils::matrix2Quat (utils.cpp:342-380, Ogre's algorithm) on the 3x3 whose ROWS are the frame axes; returns (w, x, y, z)"""
    m = np.asarray(M, np.float64).reshape(3, 3)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    q = [0.0] * 4                                                # x, y, z, w
    if tr > 0:
        root = np.sqrt(tr + 1); q[3] = 0.5 * root; root = 0.5 / root
        q[0] = (m[2, 1] - m[1, 2]) * root; q[1] = (m[0, 2] - m[2, 0]) * root; q[2] = (m[1, 0] - m[0, 1]) * root
    else:
        i = 0
        if m[1, 1] > m[0, 0]: i = 1
        if m[2, 2] > m[i, i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        root = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1); q[i] = 0.5 * root; root = 0.5 / root
        q[3] = (m[k, j] - m[j, k]) * root; q[j] = (m[j, i] + m[i, j]) * root; q[k] = (m[k, i] + m[i, k]) * root
    return np.array([q[3], q[0], q[1], q[2]])


def _qmul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def cast_votes_vector():
    """KAT 12: CodewordDistribution::castVotes / castVote (codeword_distribution.cpp:73-167). Two codewords (two and one stored
    votes), class variances sigma^2 = {0: 0.04, 1: 0.5}; six (feature, codeword, functor distance) activations that walk the
    gate |dist| > 2 sigma^2 on both sides and AT the boundary, the weight < FLT_EPSILON cut, the Gaussian matching weight
    N(dist; 0, sigma^2) = exp(-dist^2 / (2 sigma^2)) / sqrt(2 pi sigma^2), the vote geometry centre = keypoint + u0 x + u1 y + u2 z
    (rows of the frame are the axes) and the bounding-box quaternion stored (x) q(frame) (boost product order, Ogre conversion)."""
    c, s_ = np.cos(0.7), np.sin(0.7)
    R1 = np.array([[c, s_, 0], [-s_, c, 0], [0, 0, 1]])
    c2, s2 = np.cos(-1.1), np.sin(-1.1)
    R = np.array([[1, 0, 0], [0, c2, s2], [0, -s2, c2]]) @ R1
    flip = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]]) @ R      # negative trace: the other branch of matrix2Quat
    frames = [np.eye(3), R, flip]
    cb = dict(words=[[0.0, 0.0], [1.0, 0.0]], vote_offsets=[0, 2, 3],
              vote_xyz=[[0.5, -0.25, 1.0], [-1.0, 0.5, 0.25], [0.1, 0.2, -0.3]], vote_class=[0, 1, 0], vote_instance=[3, 4, 5],
              vote_weight=[0.5, 1e-8, 2e-7], vote_class_weight=[0.25, 0.75, 0.25], word_weight=[2.0, 3.0],
              vote_bbox_quat=[[1, 0, 0, 0], [np.cos(0.4), np.sin(0.4), 0, 0], [np.cos(0.9), 0, 0, np.sin(0.9)]],
              vote_bbox_size=[[1, 2, 3], [4, 5, 6], [7, 8, 9]], class_sigma=[0.04, 0.5])
    acts = [  # (frame, keypoint, codeword, dist)
        (0, [0.0, 0.0, 0.0], 0, 0.07), (1, [1.0, -2.0, 0.5], 0, 0.08), (2, [0.3, 0.1, -0.2], 0, 0.09),
        (1, [0.0, 1.0, 0.0], 1, 0.01), (2, [2.0, 0.0, 1.0], 1, 0.0), (0, [0.0, 0.0, 1.0], -1, 0.0)]
    flagsets = [0, 1, 2, 4, 8, 15]
    eps, out = float(np.finfo(np.float32).eps), {}
    for flags in flagsets:
        rows = []
        for f, kp, w, dist in acts:
            for v in range(2):                                   # max votes per word = 2 -> two slots per activation
                if w < 0 or cb["vote_offsets"][w] + v >= cb["vote_offsets"][w + 1]:
                    rows.append(dict(cls=-1)); continue
                vi = cb["vote_offsets"][w] + v
                cls = cb["vote_class"][vi]; sig = float(f32(cb["class_sigma"][cls])); d = float(f32(dist))
                wt = 1.0
                if flags & 1: wt *= cb["vote_class_weight"][vi]
                if flags & 2: wt *= float(f32(cb["vote_weight"][vi]))
                if flags & 4: wt *= float(f32(np.exp(-d * d / (2 * sig)) / np.sqrt(2 * np.pi * sig)))
                if flags & 8: wt *= cb["word_weight"][w]
                if abs(d) > float(f32(2) * f32(sig)) or float(f32(wt)) < eps:
                    rows.append(dict(cls=-1)); continue
                F = frames[f]
                centre = np.asarray(kp) + F.T @ np.asarray(cb["vote_xyz"][vi])
                rows.append(dict(cls=int(cls), inst=int(cb["vote_instance"][vi]), codeword=int(w), weight=float(wt), pos=centre.tolist(),
                                 bbox_quat=_qmul(np.asarray(cb["vote_bbox_quat"][vi], float), _ogre_quat(F)).tolist(),
                                 bbox_size=cb["vote_bbox_size"][vi]))
        out[str(flags)] = rows
    n_kept = {k: sum(r["cls"] >= 0 for r in v) for k, v in out.items()}
    # 8 stored votes are reached (2+2+2+1+1); the gate drops the class-0 vote at dist 0.09 (not the one AT 2 sigma^2 = 0.08, not the
    # class-1 vote of the same codeword); with UseVoteWeight the 1e-8 vote of codeword 0 drops three more, the 2e-7 one stays
    assert n_kept["0"] == 7 and n_kept["2"] == 4 and n_kept["4"] == 7 and n_kept["15"] == 4, n_kept
    cb = {k: np.asarray(v, float).tolist() if k not in ("vote_offsets", "vote_class", "vote_instance") else v for k, v in cb.items()}
    return dict(codebook=cb, frames=[F.reshape(-1).tolist() for F in frames], activations=[dict(frame=f, kp=kp, word=w, dist=d) for f, kp, w, d in acts],
                expected=out, tol=2e-6)


def knn_rule_table():
    """KAT 13: ActivationStrategyKnnRule at detection time (activation_strategy_knn_rule.h:79-118): the four class patterns of
    the 3 nearest codewords x both outcomes of the ratio test. Query j sits at x = 100 j; its three codewords at x + 1,
    x + a, x + b, so the squared-L2 distances are (1, a^2, b^2) (float32 arithmetic reproduced here)."""
    thr = 0.9
    far, near = (2.0, 3.0), (1.02, 1.04)                           # d1/d2, d1/d3 = .25, .11 | .961, .925
    table = [  # (classes of k1 k2 k3, offsets of k2 k3, accepted neighbour: 0 = k1, 1 = k2, -1 = none)
        ((0, 0, 0), near, 0), ((0, 0, 1), far, 0), ((0, 0, 1), near, -1), ((0, 1, 1), near, 1), ((0, 1, 1), far, -1),
        ((0, 1, 2), far, 0), ((0, 1, 2), near, -1), ((0, 1, 0), far, 0), ((0, 1, 0), near, -1)]
    words, wcls, q, exp_idx, exp_d = [], [], [], [], []
    for j, (cls, (a, b), acc) in enumerate(table):
        x0 = f32(100.0 * j)
        row0 = len(words)
        for off, c in zip((1.0, a, b), cls):
            words.append([float(f32(x0 + f32(off))), 0.0]); wcls.append(c)
        q.append([float(x0), 0.0])
        d = [float(f32(f32(f32(words[row0 + t][0]) - x0) ** 2)) for t in range(3)]
        assert d[0] < d[1] < d[2]
        r12, r13 = d[0] / d[1], d[0] / d[2]
        assert (r12 < thr) == (a == far[0]) and (r13 < thr) == (b == far[1])
        exp_idx.append(row0 + acc if acc >= 0 else -1); exp_d.append(d[acc] if acc >= 0 else None)
    return dict(words=words, word_class=wcls, q=q, threshold=thr, expected_idx=exp_idx, expected_dist=exp_d)


def maxima_thresholds():
    """KAT 14: Voting::findMaxima tail (voting.cpp:296-323, 441-462): three classes, every vote of a class at ONE point (so the
    Gaussian mean-shift mode is that point and the reweighting factor e^{-u/2} is 1): class weights 6, 3, 1 -> normalised
    0.6, 0.3, 0.1; MinThreshold > 0 is absolute, < 0 relative to the best; BestK > 0 keeps the first BestK."""
    pos = [[0, 0, 0]] * 4 + [[10, 0, 0]] * 3 + [[0, 10, 0]] * 2
    w = [1.5, 1.5, 2.0, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5]
    cls = [0] * 4 + [1] * 3 + [2] * 2
    inst = [7, 7, 8, 8, 1, 1, 2, 5, 5]
    cases = [dict(min_threshold=0.0, best_k=-1, n=3), dict(min_threshold=0.2, best_k=-1, n=2), dict(min_threshold=0.31, best_k=-1, n=1),
             dict(min_threshold=-0.4, best_k=-1, n=2), dict(min_threshold=-0.6, best_k=-1, n=1), dict(min_threshold=0.0, best_k=1, n=1),
             dict(min_threshold=0.0, best_k=2, n=2), dict(min_threshold=0.0, best_k=5, n=3)]
    return dict(pos=pos, w=w, cls=cls, inst=inst, bandwidth=1.0, n_classes=3, cases=cases, weights=[0.6, 0.3, 0.1], classes=[0, 1, 2],
                instances=[7, 1, 5], instance_weights=[3.0 / 5.0, 2.0 / 5.0, 1.0 / 5.0], n_votes=[4, 3, 2], tol=1e-6)


def hough3d_three_bins():
    """KAT 15: VotingHough3D over HoughSpace3D, space [-5,5]^3, bin 0.2 (50^3 bins, bin c centred at
    -5 + (c + 0.5) 0.2). Vote A (w 1) sits on the centre of bin (25,25,25); vote B (w 2) a quarter bin off the centre of bin
    (30,25,25) in +x and +y: trilinear split 0.75*0.75*2 = 1.125 central, 0.375 to (31,25,25) and (30,26,25), 0.125 to (31,26,25);
    vote C (w 1) on the centre of bin (25,30,25). RelThreshold 0.8 -> threshold 0.9: the three central bins are the maxima
    (ascending bin index: A, B, C), each with ITS vote's full weight -> normalised 0.25, 0.5, 0.25, sorted B, A, C."""
    ctr = lambda c: -5.0 + (c + 0.5) * 0.2
    A = [ctr(25), ctr(25), ctr(25)]; B = [ctr(30) + 0.05, ctr(25) + 0.05, ctr(25)]; Cc = [ctr(25), ctr(30), ctr(25)]
    return dict(pos=[A, B, Cc], w=[1.0, 2.0, 1.0], cls=[0, 0, 0], inst=[4, 5, 6], bin=0.2, min_coord=[-5, -5, -5], max_coord=[5, 5, 5],
                rel_threshold=0.8, n_classes=2, expected_n=3, expected_pos=[B, A, Cc], expected_weight=[0.5, 0.25, 0.25], expected_inst=[5, 4, 6],
                expected_n_votes=[1, 1, 1],
                # with RelThreshold 0.3 (threshold 0.3375) B's x and y neighbours (0.375) pass the threshold but have a strictly greater
                # neighbour (B's central bin), so the maxima stay the same three; without interpolation likewise
                tol=2e-6)


def activate_weights():
    """Round 3, KAT A: Codebook::activate on six 2-d features of two classes (codebook.cpp:64-368, codeword_distribution.cpp:169-243),
    Clustering "None", K = 1, no clean-up, squared L2, identity reference frames. Class-major rows:
        0: A0 (0,0)   1: A1 (0,0)   2: A2 (10,0)   |   3: B0 (0,0)   4: B1 (10,0)   5: B2 (20,0)
    Exact duplicates activate the LOWEST row, so the kept codewords are rows 0 (votes of features 0,1,3), 2 (features 2,4), 5 (5).
    computeWeights: vote i of a codeword with member features j gets the MEDIAN over j of exp(-|kp_j - kp_i|^2 / 0.25) (identity
    frames: centre_ij - modelCentre_i = kp_j - kp_i). Word 0 has three members (odd list), word 2 two (even list: mean of both).
    Statistical weights term1 * term2 * term3 with m_term3 keyed by CLASS only, the last codeword (ascending id) that holds the
    class wins (codebook.cpp:325-339):
        votes per class and word: c0 {w0: 2, w2: 1}, c1 {w0: 1, w2: 1, w5: 1}; features per class 3 and 3
        sum[w] = 2/3 + 1/3 = 1, 1/3 + 1/3 = 2/3, 1/3;  term1 = 1/2 (c0: two words), 1/3 (c1: three);  term2 = 1/3, 1/2, 1
        term3[c0]: w0 -> (2/3)/1, then w2 -> (1/3)/(2/3) = 1/2 (kept);  term3[c1]: 1/3, 1/2, then w5 -> (1/3)/(1/3) = 1 (kept)
        class weights: w0: c0 1/2*1/3*1/2 = 1/12, c1 1/3*1/3*1 = 1/9;  w2: c0 1/2*1/2*1/2 = 1/8, c1 1/3*1/2*1 = 1/6;  w5: c1 1/3
      (a per-codeword term3 would give w0/c0 = 1/2*1/3*2/3 = 1/9: the vector tells the two apart)."""
    feats = [[0, 0], [0, 0], [10, 0], [0, 0], [10, 0], [20, 0]]
    kp = [[0, 0, 0], [0.5, 0, 0], [0, 0, 0], [0, 1, 0], [0.3, 0.4, 0], [0, 0, 0]]
    cls = [0, 0, 0, 1, 1, 1]
    model = [0, 0, 1, 2, 3, 4]
    centre = [[1, 2, 3]] * 6
    e1, e4 = float(np.exp(-1.0)), float(np.exp(-4.0))
    return dict(feats=feats, kp=kp, cls=cls, model=model, centre=centre, k=1, n_classes=2,
                word_src=[0, 2, 5], vote_offsets=[0, 3, 5, 6], vote_feature=[0, 1, 3, 2, 4, 5],
                vote_weight=[e1, e1, e4, (1.0 + e1) / 2, (1.0 + e1) / 2, 1.0],
                vote_class_weight=[1 / 12, 1 / 12, 1 / 9, 1 / 8, 1 / 6, 1 / 3],
                vote_xyz=[[1 - k_[0], 2 - k_[1], 3 - k_[2]] for k_ in (kp[0], kp[1], kp[3], kp[2], kp[4], kp[5])], tol=2e-6)


def meanshift_step_and_double_reweight():
    """Round 3, KAT B: VotingMeanShift with Bandwidth 1, MaxIter 0 (the do-while loop runs exactly ONE mean-shift step per seed,
    voting_mean_shift.cpp:221-242), Gaussian kernel, MaximaSuppression "Suppress". Everything below is evaluated here in float64
    straight from the formulas:
      seeds: cells of edge 2h/sqrt(2), key floor(x/edge + 0.5), seed = key * edge
      step : c' = sum g_i x_i / sum g_i over the votes with d_i^2 < h^2, g_i = 0.5 exp(-d_i^2 / (2 h^2)) w_i
      density at a centre = sum exp(-d^2 / (2 h^2)) w over the votes within h; greedy suppression by descending density
      final pass IN suppression order: w_i <- exp(-d_i^2 / (2 h^2)) w_i IN PLACE for the votes within h, maximum weight = their sum
    class 0: three votes in one cell -> one seed at the origin -> one step -> one maximum (new centre, reweighted sum)
    class 1: one vote exactly on its seed -> weight 1 (pins the normalisation)
    class 2: votes A (w 4), B (w 4) 1.6 apart and M (w 1) between them: two maxima 1.34 apart; M lies within h of BOTH, so it is
             reweighted twice -- the second maximum (A's, lower density) sees M's already reduced weight (4.609 instead of 4.765)."""
    h = 1.0
    edge = 2 * h / np.sqrt(2.0)
    votes = {0: [([0.2, 0, 0], 1.0), ([-0.1, 0, 0], 2.0), ([0.5, 0, 0], 1.0)],
             1: [([3 * float(f32(edge)), 0, 0], 1.0)],
             2: [([10, 0, 0], 4.0), ([10, 1.6, 0], 4.0), ([10, 0.8, 0], 1.0)]}
    maxima = []
    for c, vs in votes.items():
        P = np.asarray([v[0] for v in vs], float); W = np.asarray([v[1] for v in vs], float)
        keys = sorted({tuple(np.floor(p / edge + 0.5).astype(int)[::-1]) for p in P})          # (z, y, x) order
        centres = []
        for kz, ky, kx in keys:
            s0 = np.asarray([kx, ky, kz], float) * edge
            d2 = ((P - s0) ** 2).sum(1); m = d2 < h * h
            if not m.any():
                continue
            g = 0.5 * np.exp(-d2[m] / (2 * h * h)) * W[m]
            centres.append((g[:, None] * P[m]).sum(0) / g.sum())
        dens = []
        for cc in centres:
            d2 = ((P - cc) ** 2).sum(1); m = d2 < h * h
            dens.append((np.exp(-d2[m] / (2 * h * h)) * W[m]).sum())
        order, work = [], list(dens)
        while True:                                                   # suppressNeighborMaxima: first largest, drop everything closer than h
            mi = int(np.argmax(work)) if max(work) > -1 else -1
            if mi < 0 or work[mi] == -1:
                break
            order.append(mi); c0 = centres[mi]; work[mi] = -1
            for i, cc in enumerate(centres):
                if np.linalg.norm(cc - c0) < h:
                    work[i] = -1
        Wc = W.copy()
        for mi in order:
            d2 = ((P - centres[mi]) ** 2).sum(1); m = d2 < h * h
            Wc[m] = np.exp(-d2[m] / (2 * h * h)) * Wc[m]
            maxima.append(dict(cls=c, pos=centres[mi].tolist(), raw=float(Wc[m].sum()), n_votes=int(m.sum())))
    assert len(maxima) == 4 and abs(maxima[2]["raw"] - 4.7653) < 1e-3 and abs(maxima[3]["raw"] - 4.6091) < 1e-3, maxima
    maxima.sort(key=lambda m_: -m_["raw"])
    tot = sum(m_["raw"] for m_ in maxima)
    pos, w, cls = [], [], []
    for c, vs in votes.items():
        for p_, w_ in vs:
            pos.append(p_); w.append(w_); cls.append(c)
    return dict(pos=pos, w=w, cls=cls, inst=[0] * len(w), n_classes=3, bandwidth=h, max_iter=0, suppression=1,
                expected_cls=[m_["cls"] for m_ in maxima], expected_pos=[m_["pos"] for m_ in maxima],
                expected_weight=[m_["raw"] / tot for m_ in maxima], expected_n_votes=[m_["n_votes"] for m_ in maxima],
                single_reweight_would_be=4.76505 / (tot - 4.60908 + 4.76505), tol=2e-6)


def _pair_features(p1, n1, p2, n2):
    """pcl::computePairFeatures restated in float64"""
    p1, n1, p2, n2 = (np.asarray(a, float) for a in (p1, n1, p2, n2))
    dp = p2 - p1; f4 = np.linalg.norm(dp)
    a1 = n1 @ dp / f4; a2 = n2 @ dp / f4
    if np.arccos(abs(a1)) > np.arccos(abs(a2)):
        n1, n2 = n2, n1; dp = -dp; f3 = -a2
    else:
        f3 = a1
    v = np.cross(dp, n1); v /= np.linalg.norm(v)
    w = np.cross(n1, v)
    return float(np.arctan2(w @ n2, n1 @ n2)), float(v @ n2), float(f3), float(f4)


def fpfh_three_points():
    """Round 3, KAT C: three surface points that all see each other, a keypoint at DIFFERENT distances from them. SPFH(p): every
    pair (p, q != p) adds 100 / (n - 1) = 50 to one bin of each 11-bin block (bins floor(11 (f1 + pi) / 2pi), floor(11 (f2 + 1) / 2),
    floor(11 (f3 + 1) / 2), clamped to 0..10). FPFH(keypoint) = sum_q SPFH(q) / d_q^2, every block rescaled to 100, i.e. the
    SPFHs are mixed with weights (1/d_q^2) / sum(1/d^2): with d = 0.1, 0.2, 0.4 that is 16/21, 4/21, 1/21."""
    pts = [[0.1, 0, 0], [-0.2, 0, 0], [0, 0.4, 0]]
    s = 1 / np.sqrt(2.0)
    nrm = [[0, 0, 1], [0, s, s], [s, 0, s]]
    kp = [0, 0, 0]
    spfh = np.zeros((3, 33))
    for i in range(3):
        for j in range(3):
            if i == j:
                continue
            f1, f2, f3, _ = _pair_features(pts[i], nrm[i], pts[j], nrm[j])
            b = [int(np.floor(11 * (f1 + np.pi) / (2 * np.pi))), int(np.floor(11 * (f2 + 1) / 2)), int(np.floor(11 * (f3 + 1) / 2))]
            for blk, bb in enumerate(b):
                spfh[i, 11 * blk + min(10, max(0, bb))] += 50.0
    d2 = np.asarray([((np.asarray(p) - kp) ** 2).sum() for p in pts])
    wq = (1 / d2) / (1 / d2).sum()
    exp = (wq[:, None] * spfh).sum(0)
    assert abs(wq[0] - 16 / 21) < 1e-12 and np.allclose(exp.reshape(3, 11).sum(1), 100)
    assert len(set(np.round(exp[exp > 0], 6))) >= 3                      # the three weights really show up as different bin values
    return dict(points=pts, normals=nrm, keypoint=kp, radius=1.0, expected=exp.tolist(), spfh=spfh.tolist(), mix=wq.tolist(), tol=1e-3)


def lrf_majority_sign():
    """Round 3, KAT D: SHOT reference frame of five neighbours (the minimum) with an ASYMMETRIC sign vote (shot_na_lrf.hpp:48-178 /
    SURVEY Appendix A.1): covariance sum (r - d) v v^T / sum (r - d) in float64, x = eigenvector of the largest eigenvalue, z = of the
    smallest; x is flipped so that at least as many neighbours have v.x >= 0 as not: here 3 neighbours on one side, 2 on the other
    (no tie rule involved); z likewise with 4 against 1; y = z cross x."""
    pts = np.asarray([[0.30, 0.05, 0.012], [0.25, -0.05, 0.008], [0.20, 0.02, 0.011], [-0.28, 0.04, -0.009], [-0.22, -0.06, 0.010]])
    r = 0.5
    d = np.linalg.norm(pts, axis=1); wgt = r - d
    cov = (wgt[:, None, None] * pts[:, :, None] * pts[:, None, :]).sum(0) / wgt.sum()
    ev, evec = np.linalg.eigh(cov)
    x, z = evec[:, 2].copy(), evec[:, 0].copy()
    px = int((pts @ x >= 0).sum()); pz = int((pts @ z >= 0).sum())
    assert {px, 5 - px} == {3, 2} and {pz, 5 - pz} == {4, 1}
    if 2 * px - 5 < 0:
        x = -x
    if 2 * pz - 5 < 0:
        z = -z
    y = np.cross(z, x)
    assert x[0] > 0.9 and z[2] > 0.9                                       # towards the three neighbours at +x, the four at +z
    return dict(points=pts.tolist(), keypoint=[0, 0, 0], radius=r, expected=np.concatenate([x, y, z]).tolist(), counts=[3, 2, 4, 1], tol=2e-6)


def pca_normals_slab():
    """Round 3, KAT E: PCA normals (pcl::NormalEstimationOMPWithEigVals as ImplicitShapeModel::computeNormals drives it,
    implicit_shape_model.cpp:969-1011) on two parallel 9 x 9 grids z = +0.5 and z = -0.5 (spacing 0.1) plus two isolated points.
    NormalRadius 0.25 never reaches the other plane, so every neighbourhood is exactly planar: the smallest eigenvector is +-e_z whatever the
    summation order. Orientation 0 (ConsistentNormalsMethod 0) flips it TOWARDS the viewpoint (0, 0, 0): (0, 0, -1) on the upper plane,
    (0, 0, +1) on the lower one; orientation 1 (method 1) points it AWAY from the cloud's centroid -- the origin, because the two isolated
    points are a mirror pair -- i.e. the opposite signs. A point with fewer than three neighbours inside the radius (the isolated ones:
    only themselves) has no normal: NaN in all components."""
    g = np.arange(-4, 5) * 0.1
    X, Y = np.meshgrid(g, g, indexing="ij")
    top = np.stack([X.ravel(), Y.ravel(), np.full(81, 0.5)], 1)
    bot = np.stack([X.ravel(), Y.ravel(), np.full(81, -0.5)], 1)
    lone = np.asarray([[3.0, 0.0, 0.0], [-3.0, 0.0, 0.0]])              # a mirror pair keeps the centroid at the origin; each is alone in its ball
    pts = np.concatenate([top, bot, lone])
    assert np.allclose(pts.mean(0), 0)
    towards = np.concatenate([np.tile([0, 0, -1.0], (81, 1)), np.tile([0, 0, 1.0], (81, 1)), np.full((2, 3), np.nan)])
    return dict(points=pts.tolist(), radius=0.25, towards_origin=[[None if np.isnan(v) else v for v in r] for r in towards.tolist()], tol=2e-6)


def kmeans_two_blobs():
    """Round 3, KAT F: ClusteringKMeans (clustering/clustering_kmeans.cpp -> flann::hierarchicalClustering, branching = number of clusters,
    one level) on two groups of four 2-d points, 14 apart with a spread of 2. Whatever the initial centres (they are drawn, so the vector
    cannot name them), the result must be a FIXED POINT of Lloyd's iteration: every centre the mean of its members, every point with its
    nearest centre, the reported distance the squared distance to it. From initial centres in different groups (15 of the 28 pairs of
    points, and what k-means++ / Gonzales seeding pick here: the second centre is the point farthest from, or drawn by squared distance
    from, the first) that fixed point is the global optimum: centres (1, 1) and (11, 11), every point at squared distance 2. The pairs
    listed in `stuck` are the same-group starts from which Lloyd stops in a worse fixed point (worked below) -- k-means is not required
    to escape them, and the check accepts them only as fixed points."""
    a = [[0, 0], [0, 2], [2, 0], [2, 2]]
    b = [[10, 10], [10, 12], [12, 10], [12, 12]]
    pts = np.asarray(a + b, float)
    good, stuck = 0, []
    for i in range(8):                                                    # Lloyd from every pair of points as initial centres
        for j in range(i + 1, 8):
            c = pts[[i, j]].copy()
            for _ in range(20):
                asg = ((pts[:, None, :] - c[None]) ** 2).sum(2).argmin(1)
                c = np.stack([pts[asg == t].mean(0) for t in range(2)])
            if sorted(map(tuple, c.tolist())) == [(1.0, 1.0), (11.0, 11.0)]:
                good += 1
            else:
                stuck.append([i, j])
            if (i < 4) != (j < 4):
                assert sorted(map(tuple, c.tolist())) == [(1.0, 1.0), (11.0, 11.0)], (i, j, c)
    assert good >= 16
    return dict(points=pts.tolist(), centres=[[1, 1], [11, 11]], groups=[0, 0, 0, 0, 1, 1, 1, 1], sqdist=2.0, stuck=stuck)


def main():
    kat = dict(pca_normals_slab=pca_normals_slab(), kmeans_two_blobs=kmeans_two_blobs(),
               shot_sector_centres=shot_sector_centres(), lrf_paraboloid=lrf_paraboloid(), rgb2lab=rgb2lab_cases(),
               fpfh_two_points=fpfh_two_points(), distances=distances(), rotations=rotations(), seeds_order=seeds_order(),
               voxel_grid=voxel_grid(), knn_ties=knn_ties(), shot_off_centre=shot_off_centre(), cshot_colour_pairs=cshot_colour_pairs(),
               cast_votes_vector=cast_votes_vector(), knn_rule_table=knn_rule_table(), maxima_thresholds=maxima_thresholds(),
               hough3d_three_bins=hough3d_three_bins(), activate_weights=activate_weights(),
               meanshift_step_and_double_reweight=meanshift_step_and_double_reweight(), fpfh_three_points=fpfh_three_points(),
               lrf_majority_sign=lrf_majority_sign())
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
