"""CPU tests: the oracle against the hand-derived known-answer vectors (tests/golden/kat.json) and against
size-independent properties. The reference ships no tests or fixtures of its own (parity unpinned, oracle/ism_oracle.h)."""
import json
import os

import numpy as np
import pytest

from conftest import make_cloud
import kat_checks

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def _soa(p):
    p = np.asarray(p, np.float32)
    return p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()


def test_shot_sector_centres(ora):
    k = KAT["shot_sector_centres"]
    x, y, z = _soa(k["points"]); nx, ny, nz = _soa(k["normals"])
    kp = np.asarray([k["keypoint"]], np.float32)
    d, cnt = ora.shot352([0, 5], x, y, z, nx, ny, nz, [0, 1], kp[:, 0], kp[:, 1], kp[:, 2], np.asarray([k["lrf"]], np.float32), k["radius"])
    assert cnt[0] == 5
    np.testing.assert_allclose(d[0], np.asarray(k["expected"], np.float32), atol=k["tol"])
    assert abs(np.linalg.norm(d[0]) - 1) < 1e-6


def test_shot_every_azimuth_sector_consistent(ora):
    # hard sector assignment and azimuth interpolation must agree for all 8 azimuth sectors, both hemispheres, both shells
    for sel in range(8):
        for upper in (0, 1):
            for outer in (0, 1):
                az = -7 * np.pi / 8 + sel * np.pi / 4
                th = np.pi / 4 if upper else 3 * np.pi / 4
                dd = 0.75 if outer else 0.25
                base = np.array([dd * np.sin(th) * np.cos(az), dd * np.sin(th) * np.sin(az), dd * np.cos(th)])
                pts = np.tile(base, (5, 1)).astype(np.float32)
                x, y, z = _soa(pts); nx, ny, nz = _soa([[0, 0, 1]] * 5)
                d, _ = ora.shot352([0, 5], x, y, z, nx, ny, nz, [0, 1], [0], [0], [0], np.eye(3, dtype=np.float32).reshape(1, 9), 1.0)
                sector = sel * 4 + (2 if outer else 0) + (1 if upper else 0)
                assert abs(d[0, sector * 11 + 10] - 1.0) < 1e-5, (sel, upper, outer)


def test_shot_fewer_than_five_neighbours_is_nan(ora):
    rng = np.random.default_rng(1)
    p, n = make_cloud(rng, 4)
    x, y, z = _soa(p); nx, ny, nz = _soa(n)
    d, cnt = ora.shot352([0, 4], x, y, z, nx, ny, nz, [0, 1], [0], [0], [0], np.eye(3, dtype=np.float32).reshape(1, 9), 10.0)
    assert cnt[0] == 4 and np.isnan(d).all()
    lrf = ora.shot_lrf([0, 4], x, y, z, [0, 1], [0], [0], [0], 10.0)
    assert np.isnan(lrf).all()
    # NaN frame -> NaN descriptor even with enough neighbours
    p, n = make_cloud(rng, 50)
    x, y, z = _soa(p); nx, ny, nz = _soa(n)
    d, _ = ora.shot352([0, 50], x, y, z, nx, ny, nz, [0, 1], [0], [0], [0], np.full((1, 9), np.nan, np.float32), 10.0)
    assert np.isnan(d).all()


def test_lrf_paraboloid(ora):
    k = KAT["lrf_paraboloid"]
    x, y, z = _soa(k["points"])
    lrf = ora.shot_lrf([0, len(x)], x, y, z, [0, 1], [k["keypoint"][0]], [k["keypoint"][1]], [k["keypoint"][2]], k["radius"])
    np.testing.assert_allclose(lrf[0], np.asarray(k["expected"], np.float32), atol=k["tol"])


def test_lrf_is_right_handed_orthonormal(ora):
    rng = np.random.default_rng(2)
    p, _ = make_cloud(rng, 3000, "ellipsoid")
    x, y, z = _soa(p)
    kp = p[rng.choice(len(p), 64, replace=False)] * 0.97
    lrf = ora.shot_lrf([0, len(p)], x, y, z, [0, 64], kp[:, 0], kp[:, 1], kp[:, 2], 0.3).reshape(-1, 3, 3)
    ok = ~np.isnan(lrf[:, 0, 0])
    assert ok.sum() > 50
    R = lrf[ok]
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (len(R), 1, 1)), atol=2e-6)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)


def test_lrf_sign_tie_uses_median_neighbours(ora):
    # 6 neighbours, 3 at +x and 3 at -x (tie on x): the 5 around the median rank by distance decide.
    # distances: +x points are the nearer ones -> ranks 0,1,2 = +x, 3,4,5 = -x; median = 3 -> ranks 1..5 -> 2 positive < 3 -> flip.
    pts = np.array([[0.10, 0.01, 0.0], [0.12, -0.01, 0.001], [0.14, 0.0, -0.001],
                    [-0.20, 0.01, 0.0], [-0.22, -0.01, 0.001], [-0.24, 0.0, -0.001]], np.float32)
    x, y, z = _soa(pts)
    lrf = ora.shot_lrf([0, 6], x, y, z, [0, 1], [0], [0], [0], 1.0)[0].reshape(3, 3)
    assert abs(abs(lrf[0, 0]) - 1) < 1e-3
    assert lrf[0, 0] < 0, "x axis must point to the side holding the majority of the 5 median neighbours"


def test_rgb2lab(ora):
    for c in KAT["rgb2lab"]["cases"]:
        L, a, b = ora.rgb2lab(c["rgba"])
        assert abs(L - c["L"]) < KAT["rgb2lab"]["tol"] and abs(a - c["a"]) < KAT["rgb2lab"]["tol"] and abs(b - c["b"]) < KAT["rgb2lab"]["tol"]
    assert abs(ora.rgb2lab(0)[0]) < 1e-4 and abs(ora.rgb2lab(0xffffff)[1]) < 1e-6


def test_cshot_shape_part_matches_shot_direction(ora):
    # the first 352 entries of CSHOT are the SHOT histogram before the joint normalisation: same direction
    rng = np.random.default_rng(3)
    p, n = make_cloud(rng, 2000)
    rgba = rng.integers(0, 1 << 24, size=len(p)).astype(np.uint32)
    x, y, z = _soa(p); nx, ny, nz = _soa(n)
    kp = p[:8] * 0.98
    lrf = ora.shot_lrf([0, len(p)], x, y, z, [0, 8], kp[:, 0], kp[:, 1], kp[:, 2], 0.3)
    s, _ = ora.shot352([0, len(p)], x, y, z, nx, ny, nz, [0, 8], kp[:, 0], kp[:, 1], kp[:, 2], lrf, 0.4)
    c, _ = ora.cshot1344([0, len(p)], x, y, z, nx, ny, nz, rgba, [0, 8], kp[:, 0], kp[:, 1], kp[:, 2], rgba[:8], lrf, 0.4)
    np.testing.assert_allclose(np.linalg.norm(c, axis=1), 1.0, atol=1e-5)
    shape = c[:, :352] / np.linalg.norm(c[:, :352], axis=1, keepdims=True)
    np.testing.assert_allclose(shape, s, atol=1e-5)
    # both channels receive the same total mass (4 per neighbour before normalisation)
    np.testing.assert_allclose(c[:, :352].sum(1), c[:, 352:].sum(1), rtol=1e-4)


def test_fpfh_two_points(ora):
    k = KAT["fpfh_two_points"]
    x, y, z = _soa(k["points"]); nx, ny, nz = _soa(k["normals"])
    ok, f = ora.pair_features(k["points"][0], k["normals"][0], k["points"][1], k["normals"][1])
    assert ok
    np.testing.assert_allclose(f, [k["pair"]["f1"], k["pair"]["f2"], k["pair"]["f3"], k["pair"]["f4"]], atol=1e-6)
    d, cnt = ora.fpfh33([0, 2], x, y, z, nx, ny, nz, [0, 1], [k["keypoint"][0]], [k["keypoint"][1]], [k["keypoint"][2]], k["radius"])
    assert cnt[0] == 2
    np.testing.assert_allclose(d[0], np.asarray(k["expected"], np.float32), atol=k["tol"])


def test_fpfh_blocks_sum_to_100(ora):
    rng = np.random.default_rng(4)
    p, n = make_cloud(rng, 1500, "ellipsoid")
    x, y, z = _soa(p); nx, ny, nz = _soa(n)
    kp = p[:16] * 0.99
    d, _ = ora.fpfh33([0, len(p)], x, y, z, nx, ny, nz, [0, 16], kp[:, 0], kp[:, 1], kp[:, 2], 0.25)
    np.testing.assert_allclose(d.reshape(16, 3, 11).sum(-1), 100.0, rtol=1e-4)
    # a keypoint with nothing in range -> NaN row
    d, cnt = ora.fpfh33([0, len(p)], x, y, z, nx, ny, nz, [0, 1], [50.0], [0], [0], 0.25)
    assert cnt[0] == 0 and np.isnan(d).all()


def test_distances(ora):
    k = KAT["distances"]
    assert abs(ora.distance(0, k["a"], k["b"]) - k["l2"]) < k["tol"]
    assert abs(ora.distance(1, k["a"], k["b"]) - k["chi2"]) < k["tol"]
    assert ora.distance(0, k["a"], k["b"]) == ora.distance(0, k["b"], k["a"])
    assert ora.distance(1, k["a"], k["b"]) == ora.distance(1, k["b"], k["a"])


def test_knn_ties_lowest_row(ora):
    k = KAT["knn_ties"]
    idx, dist = ora.knn(0, k["words"], k["q"], k["k"])
    assert idx.tolist() == k["expected_idx"]
    np.testing.assert_allclose(dist, k["expected_l2"], atol=1e-6)
    idx, _ = ora.knn(0, k["words"][:2], k["q"], 3)         # fewer words than k -> -1 padding
    assert (idx[:, 2] == -1).all()


def test_knn_ratio(ora):
    words = np.eye(4, dtype=np.float32)
    q = np.array([[1, 0.05, 0, 0], [0.5, 0.5, 0, 0.02]], np.float32)
    idx, _ = ora.knn_ratio(0, words, q, 0.8)
    assert idx[0, 0] == 0 and idx[1, 0] == -1     # second query is ambiguous: d1/d2 > 0.8 -> discarded


def test_rotations(ora):
    for c in KAT["rotations"]["cases"]:
        np.testing.assert_allclose(ora.rotate_into(c["lrf"], c["v"]), c["into"], atol=KAT["rotations"]["tol"])
        np.testing.assert_allclose(ora.rotate_back(c["lrf"], c["v"]), c["back"], atol=KAT["rotations"]["tol"])
        np.testing.assert_allclose(ora.rotate_back(c["lrf"], ora.rotate_into(c["lrf"], c["v"])), c["v"], atol=KAT["rotations"]["tol"])
        q = ora.rot_quaternion(c["lrf"])
        assert abs(np.linalg.norm(q) - 1) < 1e-6


def test_seeds_order(ora):
    k = KAT["seeds_order"]
    sp, sw = ora.create_seeds(k["pos"], k["w"], k["bin"])
    np.testing.assert_allclose(sp, k["expected_pos"], atol=1e-6)
    np.testing.assert_allclose(sw, k["expected_w"], atol=1e-6)


def test_voxel_grid(ora):
    k = KAT["voxel_grid"]
    x, y, z = _soa(k["points"])
    kx, ky, kz, _ = ora.voxel_grid(x, y, z, k["leaf"])
    np.testing.assert_allclose(np.stack([kx, ky, kz], 1), k["expected"], atol=k["tol"])


def test_radius_search_is_strict_and_sorted(ora):
    x = np.array([0.0, 0.5, 1.0, 0.25, 2.0], np.float32)
    z0 = np.zeros(5, np.float32)
    idx, d2 = ora.radius_search(x, z0, z0, (0, 0, 0), 1.0)
    assert idx.tolist() == [0, 3, 1]            # the point at distance exactly r is excluded (FLANN: dist < r^2)
    np.testing.assert_allclose(d2, [0, 0.0625, 0.25])


def _blobs(rng, centres, n, sigma, cls):
    pos = np.concatenate([c + sigma * rng.normal(size=(n, 3)) for c in centres]).astype(np.float32)
    return dict(pos=pos, weight=np.ones(len(pos), np.float32), cls=np.full(len(pos), cls, np.int32),
                inst=np.arange(len(pos), dtype=np.int32) % 3)


def test_meanshift_two_blobs(ora):
    rng = np.random.default_rng(5)
    centres = [np.array([0.0, 0, 0]), np.array([3.0, 0, 0])]
    v = _blobs(rng, centres, 60, 0.05, cls=1)
    out = ora.find_maxima([0, len(v["pos"])], v, n_classes=3, bandwidth=0.5, max_maxima=8)
    assert out["n"][0] == 2
    got = out["pos"][0, :2]
    got = got[np.argsort(got[:, 0])]
    np.testing.assert_allclose(got, np.stack(centres), atol=0.03)
    assert (out["cls"][0, :2] == 1).all()
    np.testing.assert_allclose(out["weight"][0, :2].sum(), 1.0, atol=1e-6)       # normalizeWeights
    assert out["weight"][0, 0] >= out["weight"][0, 1]                               # sorted
    np.testing.assert_allclose(out["class_score"][0], [0, out["weight"][0, 0], 0], atol=1e-7)
    assert out["n_votes"][0, 0] == 60


def test_meanshift_thresholds_and_bestk(ora):
    rng = np.random.default_rng(6)
    v1 = _blobs(rng, [np.array([0.0, 0, 0])], 80, 0.05, cls=0)
    v2 = _blobs(rng, [np.array([5.0, 0, 0])], 20, 0.05, cls=2)
    v = {k: np.concatenate([v1[k], v2[k]]) for k in v1}
    full = ora.find_maxima([0, 100], v, 3, 0.5, max_maxima=8)
    assert full["n"][0] == 2 and full["cls"][0, 0] == 0 and full["cls"][0, 1] == 2
    rel = ora.find_maxima([0, 100], v, 3, 0.5, max_maxima=8, min_threshold=-0.5)       # relative to the best
    assert rel["n"][0] == 1
    bk = ora.find_maxima([0, 100], v, 3, 0.5, max_maxima=8, best_k=1)
    assert bk["n"][0] == 1 and bk["cls"][0, 0] == 0
    mv = ora.find_maxima([0, 100], v, 3, 0.5, max_maxima=8, min_votes_threshold=50)
    assert mv["n"][0] == 1
    # empty object and object whose slots cast no vote
    v["cls"][:] = -1
    e = ora.find_maxima([0, 0, 100], v, 3, 0.5, max_maxima=4)
    assert e["n"].tolist() == [0, 0] and (e["class_score"] == 0).all()


def test_cast_votes_gate_and_geometry(ora):
    rng = np.random.default_rng(7)
    lrf = np.array([[0, 1, 0, -1, 0, 0, 0, 0, 1]], np.float32)           # rows = axes
    cb = dict(words=rng.random((3, 8)).astype(np.float32), vote_offsets=[0, 1, 1, 3],
              vote_xyz=np.array([[1, 2, 3], [0.5, 0, 0], [0, 0.5, 0]], np.float32), vote_class=[0, 1, 1], vote_instance=[4, 5, 6],
              class_sigma=np.array([1.0, 0.1], np.float32), vote_weight=np.array([0.5, 1, 1], np.float32),
              vote_class_weight=np.array([0.25, 1, 1], np.float32), word_weight=np.array([2.0, 1, 1], np.float32))
    idx = np.array([[0], [2], [1], [-1]], np.int32)
    dist = np.array([[0.5], [0.5], [0.1], [0.0]], np.float32)
    lr = np.tile(lrf, (4, 1))
    kp = np.array([[10, 0, 0]] * 4, np.float32)
    v = ora.cast_votes(cb, 0, lr, kp[:, 0], kp[:, 1], kp[:, 2], idx, dist)
    assert len(v["cls"]) == 4 * 1 * 2
    # feature 0 -> word 0, one vote, centre = kp + (1*x + 2*y + 3*z) = (10,0,0) + (0,1,0) + (-2,0,0) + (0,0,3)
    np.testing.assert_allclose(v["pos"][0], [8, 1, 3], atol=1e-5)
    assert v["cls"][0] == 0 and v["inst"][0] == 4 and v["weight"][0] == 1.0 and v["cls"][1] == -1
    # feature 1 -> word 2 (class 1, sigma 0.1): |d| = 0.5 > 2*0.1 -> both votes discarded
    assert (v["cls"][2:4] == -1).all()
    # feature 2 -> word 1 has no votes; feature 3 -> no match
    assert (v["cls"][4:] == -1).all()
    w = ora.cast_votes(cb, 1 | 2 | 8, lr, kp[:, 0], kp[:, 1], kp[:, 2], idx, dist)
    assert abs(w["weight"][0] - 0.25 * 0.5 * 2.0) < 1e-7
    m = ora.cast_votes(cb, 4, lr, kp[:, 0], kp[:, 1], kp[:, 2], idx, dist)
    assert abs(m["weight"][0] - (1 / np.sqrt(2 * np.pi * 1.0)) * np.exp(-0.25 / 2.0)) < 1e-6


def test_activate_matches_class_sigmas_and_single_vote_weights(ora):
    """ismref_activate against the older sigma helper, and the closed forms of the K = 1 case (codebook.cpp:201-368): every kept
    word has one vote, vote weight exp(0) = 1 (the vote reproduces the model centre), class weight 1 / words of the class."""
    rng = np.random.default_rng(8)
    n = 600
    feats = rng.random((n, 16)).astype(np.float32)
    cls = np.repeat(np.arange(3), 200).astype(np.uint32)
    model = (np.arange(n) // 50).astype(np.uint32)
    A = rng.normal(size=(n, 3, 3)); Q, _ = np.linalg.qr(A); Q[np.linalg.det(Q) < 0, 2] *= -1
    lrf = Q.reshape(n, 9).astype(np.float32); kp = rng.normal(size=(n, 3)).astype(np.float32)
    centre = np.repeat(rng.normal(size=(12, 3)), 50, axis=0).astype(np.float32)
    for metric in (0, 1):
        act, _ = ora.knn(metric, feats, feats, 1)
        r = ora.activate(metric, feats, lrf, kp, cls, model, centre, k=1, clean_up=True, n_classes=3)
        np.testing.assert_array_equal(r["class_sigma"], ora.class_sigmas(metric, feats, cls, model, act[:, 0], feats, 3))
        assert (np.diff(r["vote_offsets"].astype(np.int64)) == 1).all()
        np.testing.assert_allclose(r["vote_weight"], 1.0, atol=1e-6)
        per_class = np.bincount(cls[r["vote_feature"]], minlength=3)
        np.testing.assert_allclose(r["vote_class_weight"], 1.0 / per_class[cls[r["vote_feature"]]], rtol=1e-6)
        # vote = rows-of-the-frame rotation of (centre - keypoint)
        v = np.einsum("nij,nj->ni", lrf.reshape(-1, 3, 3)[r["vote_feature"]], (centre - kp)[r["vote_feature"]])
        np.testing.assert_allclose(r["vote_xyz"], v, atol=2e-6)
    # K = 2, no clean-up: every feature votes twice; term3 is taken from the LAST word holding the class
    r = ora.activate(0, feats, lrf, kp, cls, model, centre, k=2, clean_up=False, n_classes=3)
    assert r["vote_offsets"][-1] == 2 * n and ((r["vote_weight"] >= 0) & (r["vote_weight"] <= 1)).all()
    e_of_vote = np.repeat(np.arange(len(r["word_src"])), np.diff(r["vote_offsets"].astype(np.int64)))
    vc = cls[r["vote_feature"]]
    nF = np.bincount(vc, minlength=3).astype(np.float32)
    for c in range(3):
        e_last = e_of_vote[vc == c].max()
        in_e = e_of_vote == e_last
        s_e = sum(np.float32((vc[in_e] == cc).sum()) / nF[cc] for cc in sorted(set(vc[in_e])))
        t3 = (np.float32((vc[in_e] == c).sum()) / nF[c]) / np.float32(s_e)
        t1 = 1.0 / len(set(e_of_vote[vc == c]))
        sel = np.nonzero(vc == c)[0][:5]
        for v_ in sel:
            m = (e_of_vote == e_of_vote[v_]).sum()
            np.testing.assert_allclose(r["vote_class_weight"][v_], t1 * (1.0 / m) * t3, rtol=1e-5)


# ---- round-2 vectors (kat_checks.py runs the same checks on the HIP path in test_gpu_parity.py) ---------------------------
def _ora_shot(ora):
    def f(pts, nrm, radius):
        x, y, z = _soa(pts); nx, ny, nz = _soa(nrm)
        d, cnt = ora.shot352([0, len(pts)], x, y, z, nx, ny, nz, [0, 1], [0], [0], [0], np.eye(3, dtype=np.float32).reshape(1, 9), radius)
        assert cnt[0] == len(pts)
        return d[0]
    return f


def test_shot_off_centre_interpolation(ora):
    kat_checks.shot_off_centre(_ora_shot(ora))


def test_cshot_colour_channel(ora):
    def f(pts, nrm, rgba, kp_rgba, radius):
        x, y, z = _soa(pts); nx, ny, nz = _soa(nrm)
        d, _ = ora.cshot1344([0, len(pts)], x, y, z, nx, ny, nz, rgba, [0, 1], [0], [0], [0], np.asarray([kp_rgba], np.uint32),
                             np.eye(3, dtype=np.float32).reshape(1, 9), radius)
        return d[0]
    kat_checks.cshot_colour_pairs(f)


def test_cast_votes_vector(ora):
    kat_checks.cast_votes_vector(lambda cb, flags, lrf, kp, idx, dist: ora.cast_votes(cb, flags, lrf, kp[:, 0], kp[:, 1], kp[:, 2], idx, dist))


def test_knn_rule_truth_table(ora):
    kat_checks.knn_rule_table(ora.knn_rule)


def test_maxima_thresholds_and_bestk_vector(ora):
    kat_checks.maxima_thresholds(ora.find_maxima)


def test_hough3d_three_bins_vector(ora):
    kat_checks.hough3d_three_bins(ora.hough3d_maxima)


def test_hough3d_properties(ora):
    rng = np.random.default_rng(9)
    # two tight blobs of one class far apart -> two maxima near the blob means; a vote outside the space is ignored
    a = np.array([0.31, -1.07, 0.52]) + 0.02 * rng.normal(size=(40, 3)); b = np.array([-2.2, 1.4, 0.9]) + 0.02 * rng.normal(size=(25, 3))
    pos = np.concatenate([a, b, [[7.0, 0, 0]]]).astype(np.float32)
    v = dict(pos=pos, weight=np.ones(len(pos), np.float32), cls=np.ones(len(pos), np.int32), inst=(np.arange(len(pos)) % 3).astype(np.int32))
    out = ora.hough3d_maxima([0, len(pos)], v, 3, 0.4, rel_threshold=0.3, max_maxima=8)
    assert out["n"][0] == 2 and (out["cls"][0, :2] == 1).all()
    np.testing.assert_allclose(out["pos"][0, 0], a.mean(0), atol=0.05)
    np.testing.assert_allclose(out["pos"][0, 1], b.mean(0), atol=0.05)
    assert out["weight"][0, 0] > out["weight"][0, 1] and abs(out["weight"][0, :2].sum() - 1) < 1e-6
    e = ora.hough3d_maxima([0, 0, len(pos)], dict(v, cls=np.full(len(pos), -1, np.int32)), 3, 0.4, max_maxima=4)
    assert e["n"].tolist() == [0, 0]


def test_max_filter_simple_keeps_the_heavier_of_two_colliding_classes(ora):
    """MaximaHandler::filterMaxima "Simple": two classes peaking 0.1 apart (bandwidth 0.5) -> only the heavier maximum survives;
    a third class far away is untouched; without the filter all three are reported."""
    pos = [[0, 0, 0]] * 4 + [[0.1, 0, 0]] * 3 + [[5, 0, 0]] * 2
    v = dict(pos=np.asarray(pos, np.float32), weight=np.ones(9, np.float32), cls=np.asarray([0] * 4 + [1] * 3 + [2] * 2, np.int32), inst=np.zeros(9, np.int32))
    a = ora.find_maxima([0, 9], v, n_classes=3, bandwidth=0.5, max_maxima=8)
    b = ora.find_maxima([0, 9], v, n_classes=3, bandwidth=0.5, max_maxima=8, max_filter=1)
    assert a["n"][0] == 3 and b["n"][0] == 2 and b["cls"][0, :2].tolist() == [0, 2]
    np.testing.assert_allclose(b["weight"][0, :2], [4 / 6, 2 / 6], atol=1e-6)


# ---- round-3 vectors -----------------------------------------------------------------------------------------------------------
def test_activate_weights_vector(ora):
    kat_checks.activate_weights(lambda metric, feats, lrf, kp, cls, model, centre, k, clean_up, n_classes:
                                ora.activate(metric, feats, lrf, kp, cls, model, centre, k=k, clean_up=clean_up, n_classes=n_classes))


def test_meanshift_step_and_double_reweight_vector(ora):
    kat_checks.meanshift_step_and_double_reweight(ora.find_maxima)


def test_fpfh_three_points_vector(ora):
    def f(pts, nrm, kp, radius):
        x, y, z = _soa(pts); nx, ny, nz = _soa(nrm)
        d, cnt = ora.fpfh33([0, len(pts)], x, y, z, nx, ny, nz, [0, 1], [kp[0]], [kp[1]], [kp[2]], radius)
        assert cnt[0] == len(pts)
        return d[0]
    kat_checks.fpfh_three_points(f)


def test_lrf_majority_sign_vector(ora):
    def f(pts, kp, radius):
        x, y, z = _soa(pts)
        return ora.shot_lrf([0, len(pts)], x, y, z, [0, 1], [kp[0]], [kp[1]], [kp[2]], radius)[0]
    kat_checks.lrf_majority_sign(f)


def test_pca_normals_slab_vector(ora):
    def f(pts, radius, orientation):
        x, y, z = _soa(pts)
        return ora.pca_normals([0, len(pts)], x, y, z, radius, orientation)
    kat_checks.pca_normals_slab(f)


def test_kmeans_two_blobs_vector(ora):
    def f(pts, n_clusters, init, seed):
        c, asg, d, _ = ora.kmeans(0, pts, n_clusters, centers_init=init, seed=seed)
        return c, asg, d
    kat_checks.kmeans_two_blobs(f)


def test_kat_json_is_what_the_committed_script_writes(tmp_path):
    """tests/golden/kat.json must be exactly the output of tests/golden/make_golden.py (closed-form numpy, no oracle, no HIP): a vector
    nobody can regenerate is not a fixture."""
    import json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "kat.json")
    r = subprocess.run([sys.executable, os.path.join(here, "golden", "make_golden.py"), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.load(open(out)) == json.load(open(os.path.join(here, "golden", "kat.json")))
