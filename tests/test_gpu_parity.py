"""GPU parity tests (-m gpu): the HIP path behind the C ABI against the CPU oracle on the same seeded inputs, against the
hand-derived KATs, and through size-independent properties at BASELINE.json's full sizes.
Tolerance (north_star): 1e-4 on descriptor vectors and vote weights; indices / labels bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import make_cloud
import kat_checks

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
TOL = 1e-4


def T(a, dev, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


class Scene:
    """several objects concatenated: numpy SoA + device tensors + a Cloud"""

    def __init__(self, pkg, gpu, objs, kps, cell, rgba=None, kp_rgba=None):
        import torch
        self.ctx, self.dev = gpu
        self.pt_off = np.concatenate([[0], np.cumsum([len(o[0]) for o in objs])]).astype(np.uint32)
        self.kp_off = np.concatenate([[0], np.cumsum([len(k) for k in kps])]).astype(np.uint32)
        cat = lambda parts, w: np.concatenate(parts).astype(np.float32).reshape(-1, w) if len(parts) else np.zeros((0, w), np.float32)
        self.p = cat([o[0] for o in objs], 3); self.n = cat([o[1] for o in objs], 3); self.kp = cat(kps, 3)
        self.rgba = None if rgba is None else np.concatenate(rgba).astype(np.uint32)
        self.kp_rgba = None if kp_rgba is None else np.concatenate(kp_rgba).astype(np.uint32)
        d = self.dev
        self.t = [T(self.p[:, i], d) for i in range(3)] + [T(self.n[:, i], d) for i in range(3)]
        self.tk = [T(self.kp[:, i], d) for i in range(3)]
        self.t_rgba = None if self.rgba is None else T(self.rgba.astype(np.int64), d, torch.int32)
        self.t_kp_rgba = None if self.kp_rgba is None else T(self.kp_rgba.astype(np.int64), d, torch.int32)
        self.cloud = pkg.capi.Cloud(self.ctx, self.pt_off, *self.t, cell, rgba=self.t_rgba)

    def soa(self):
        return [self.p[:, i].copy() for i in range(3)], [self.n[:, i].copy() for i in range(3)], [self.kp[:, i].copy() for i in range(3)]


def edge_scene(pkg, gpu, seed=0, with_color=False, cell=0.15):
    """ragged batch: sphere, ellipsoid with NaN points, EMPTY object, tiny object (<5 points), noisy plane"""
    rng = np.random.default_rng(seed)
    o0 = make_cloud(rng, 3000, "sphere", noise=0.01)
    o1 = list(make_cloud(rng, 2500, "ellipsoid"))
    o1[0][::97] = np.nan                                    # non-finite points are dropped by the search surface
    o2 = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    o3 = make_cloud(rng, 4, "sphere")
    o4 = make_cloud(rng, 2000, "plane", noise=0.02)
    objs = [o0, tuple(o1), o2, o3, o4]
    kps = []
    for i, (p, _) in enumerate(objs):
        if len(p) == 0:
            kps.append(np.array([[0.1, 0.2, 0.3]], np.float32))        # a keypoint in an empty object
            continue
        ok = np.isfinite(p).all(1)
        sel = p[ok][rng.choice(ok.sum(), min(40, ok.sum()), replace=False)]
        k = sel * 0.98
        k[0] = sel[0]                                                     # coincides with a surface point
        k = np.concatenate([k, [[30.0, 0, 0]], [[np.nan, 0, 0]]]).astype(np.float32)   # far away / non-finite keypoint
        kps.append(k)
    rgba = kp_rgba = None
    if with_color:
        rgba = [rng.integers(0, 1 << 24, size=len(o[0])).astype(np.uint32) for o in objs]
        kp_rgba = [rng.integers(0, 1 << 24, size=len(k)).astype(np.uint32) for k in kps]
    return Scene(pkg, gpu, objs, kps, cell, rgba, kp_rgba)


def assert_close_nan(a, b, atol):
    assert a.shape == b.shape
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"NaN pattern differs: {na.sum()} vs {nb.sum()}"
    if (~na).any():
        assert np.abs(a[~na] - b[~nb]).max() <= atol, np.abs(a[~na] - b[~nb]).max()


# ------------------------------------------------------------------------------------------------ LRF
@pytest.mark.parametrize("radius", [0.3, 0.12])
def test_lrf_matches_oracle(pkg, gpu, ora, radius):
    s = edge_scene(pkg, gpu, 1)
    ctx, dev = gpu
    got = pkg.capi.shot_lrf(ctx, s.cloud, s.kp_off, *s.tk, radius).cpu().numpy()
    (x, y, z), _, (kx, ky, kz) = s.soa()
    want = ora.shot_lrf(s.pt_off, x, y, z, s.kp_off, kx, ky, kz, radius)
    assert_close_nan(got, want, 1e-5)
    assert np.isnan(want).any() and (~np.isnan(want)).any()


def test_lrf_kat_and_forced_ties(pkg, gpu, ora):
    ctx, dev = gpu
    k = KAT["lrf_paraboloid"]
    tie = np.array([[0.10, 0.01, 0.0], [0.12, -0.01, 0.001], [0.14, 0.0, -0.001],
                    [-0.20, 0.01, 0.0], [-0.22, -0.01, 0.001], [-0.24, 0.0, -0.001]], np.float32)
    rng = np.random.default_rng(3)
    # symmetric clouds: many exact sign ties (mirror pairs) -> exercises k_lrf_tie
    base = rng.normal(size=(400, 3)).astype(np.float32) * 0.2
    sym = np.concatenate([base, -base]).astype(np.float32)
    objs = [(np.asarray(k["points"], np.float32), np.zeros((len(k["points"]), 3), np.float32)), (tie, np.zeros((6, 3), np.float32)),
            (sym, np.zeros((800, 3), np.float32))]
    kps = [np.asarray([k["keypoint"]], np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32)]
    s = Scene(pkg, gpu, objs, kps, 0.25)
    got = pkg.capi.shot_lrf(ctx, s.cloud, s.kp_off, *s.tk, 0.5).cpu().numpy()
    np.testing.assert_allclose(got[0], np.asarray(k["expected"], np.float32), atol=k["tol"])
    (x, y, z), _, (kx, ky, kz) = s.soa()
    want = ora.shot_lrf(s.pt_off, x, y, z, s.kp_off, kx, ky, kz, 0.5)
    assert_close_nan(got, want, 1e-5)
    assert got[1, 0] < 0


# ------------------------------------------------------------------------------------------------ descriptors
def test_shot352_matches_oracle(pkg, gpu, ora):
    s = edge_scene(pkg, gpu, 2)
    ctx, dev = gpu
    lrf = pkg.capi.shot_lrf(ctx, s.cloud, s.kp_off, *s.tk, 0.3)
    got, cnt = pkg.capi.shot352(ctx, s.cloud, s.kp_off, *s.tk, lrf, 0.4, want_counts=True)
    (x, y, z), (nx, ny, nz), (kx, ky, kz) = s.soa()
    want, wcnt = ora.shot352(s.pt_off, x, y, z, nx, ny, nz, s.kp_off, kx, ky, kz, lrf.cpu().numpy(), 0.4)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), wcnt)
    assert_close_nan(got.cpu().numpy(), want, TOL)
    fin = ~np.isnan(want[:, 0])
    np.testing.assert_allclose(np.linalg.norm(got.cpu().numpy()[fin], axis=1), 1.0, atol=1e-5)


def test_shot352_kat(pkg, gpu):
    ctx, dev = gpu
    k = KAT["shot_sector_centres"]
    s = Scene(pkg, gpu, [(np.asarray(k["points"], np.float32), np.asarray(k["normals"], np.float32))], [np.asarray([k["keypoint"]], np.float32)], 0.5)
    got = pkg.capi.shot352(ctx, s.cloud, s.kp_off, *s.tk, T(np.asarray([k["lrf"]], np.float32), dev), k["radius"]).cpu().numpy()
    np.testing.assert_allclose(got[0], np.asarray(k["expected"], np.float32), atol=5e-6)


def test_cshot1344_matches_oracle(pkg, gpu, ora):
    s = edge_scene(pkg, gpu, 4, with_color=True)
    ctx, dev = gpu
    lrf = pkg.capi.shot_lrf(ctx, s.cloud, s.kp_off, *s.tk, 0.3)
    got, cnt = pkg.capi.cshot1344(ctx, s.cloud, s.kp_off, *s.tk, s.t_kp_rgba, lrf, 0.35, want_counts=True)
    (x, y, z), (nx, ny, nz), (kx, ky, kz) = s.soa()
    want, wcnt = ora.cshot1344(s.pt_off, x, y, z, nx, ny, nz, s.rgba, s.kp_off, kx, ky, kz, s.kp_rgba, lrf.cpu().numpy(), 0.35)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), wcnt)
    assert_close_nan(got.cpu().numpy(), want, TOL)


def test_fpfh33_matches_oracle(pkg, gpu, ora):
    s = edge_scene(pkg, gpu, 5, cell=0.1)
    ctx, dev = gpu
    got, cnt = pkg.capi.fpfh33(ctx, s.cloud, s.kp_off, *s.tk, 0.2, want_counts=True)
    (x, y, z), (nx, ny, nz), (kx, ky, kz) = s.soa()
    want, wcnt = ora.fpfh33(s.pt_off, x, y, z, nx, ny, nz, s.kp_off, kx, ky, kz, 0.2)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), wcnt)
    # FPFH lives on a 0..100 scale: 1e-4 relative to the scale
    assert_close_nan(got.cpu().numpy(), want, 100 * TOL)


def test_fpfh33_kat(pkg, gpu):
    ctx, dev = gpu
    k = KAT["fpfh_two_points"]
    s = Scene(pkg, gpu, [(np.asarray(k["points"], np.float32), np.asarray(k["normals"], np.float32))], [np.asarray([k["keypoint"]], np.float32)], 0.25)
    got = pkg.capi.fpfh33(ctx, s.cloud, s.kp_off, *s.tk, k["radius"]).cpu().numpy()
    np.testing.assert_allclose(got[0], np.asarray(k["expected"], np.float32), atol=1e-3)


def test_centroids_and_center_dist(pkg, gpu, ora):
    s = edge_scene(pkg, gpu, 6)
    ctx, dev = gpu
    (x, y, z), _, (kx, ky, kz) = s.soa()
    np.testing.assert_allclose(pkg.capi.cloud_centroids(ctx, s.cloud, dev).cpu().numpy(), ora.centroids(s.pt_off, x, y, z), atol=1e-6)
    got = pkg.capi.center_dist(ctx, s.cloud, s.kp_off, *s.tk).cpu().numpy()
    want = ora.center_dist(s.pt_off, x, y, z, s.kp_off, kx, ky, kz)
    assert_close_nan(got, want, 1e-5)


def test_compact_features(pkg, gpu):
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(7)
    kp_off = np.array([0, 5, 5, 12, 20], np.uint32)
    desc = rng.random((20, 33)).astype(np.float32)
    lrf = rng.random((20, 9)).astype(np.float32)
    desc[[1, 7, 19], 3] = np.nan
    lrf[[2, 7], 0] = np.nan
    kp = rng.random((20, 3)).astype(np.float32)
    keep, d, l, x, y, z, src = pkg.capi.compact_features(ctx, kp_off, T(desc, dev), T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev))
    good = np.array([i for i in range(20) if i not in (1, 2, 7, 19)])
    assert keep.tolist() == [0, 3, 3, 9, 16]
    assert np.array_equal(src.cpu().numpy(), good)
    np.testing.assert_array_equal(d.cpu().numpy(), desc[good]); np.testing.assert_array_equal(l.cpu().numpy(), lrf[good])
    np.testing.assert_array_equal(y.cpu().numpy(), kp[good, 1])


def test_compact_descriptor_rows(pkg, gpu):
    """ismhip_compact_descriptor_rows: rows that are NaN as a whole (what the descriptor kernels write) are dropped like
    ismhip_compact_features drops them; when nothing goes, the input tensors come back untouched and uncopied."""
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(8)
    kp_off = np.array([0, 300, 300, 1000, 1100], np.uint32)
    desc = rng.random((1100, 352)).astype(np.float32)
    lrf = rng.random((1100, 9)).astype(np.float32)
    kp = rng.random((1100, 3)).astype(np.float32)
    args = lambda d, l: (ctx, kp_off, T(d, dev), T(l, dev), T(kp[:, 0].copy(), dev), T(kp[:, 1].copy(), dev), T(kp[:, 2].copy(), dev))
    a = args(desc, lrf)
    keep, d, l, x, y, z, src = pkg.capi.compact_descriptor_rows(*a)
    assert keep.tolist() == kp_off.tolist() and d.data_ptr() == a[2].data_ptr() and l.data_ptr() == a[3].data_ptr() and x.data_ptr() == a[4].data_ptr()
    assert np.array_equal(src.cpu().numpy(), np.arange(1100))
    bad_rows = rng.choice(1100, 90, replace=False)
    desc2, lrf2 = desc.copy(), lrf.copy()
    desc2[bad_rows[:50]] = np.nan
    lrf2[bad_rows[50:70], 3] = np.nan; lrf2[bad_rows[70:], 6] = np.inf
    got = pkg.capi.compact_descriptor_rows(*args(desc2, lrf2))
    want = pkg.capi.compact_features(*args(desc2, lrf2))
    assert got[0].tolist() == want[0].tolist() and int(got[0][-1]) == 1100 - 90
    for g, w in zip(got[1:], want[1:]):
        np.testing.assert_array_equal(g.cpu().numpy(), w.cpu().numpy())


def test_filter_normals_on_the_device(pkg, gpu):
    """ImplicitShapeModel::filterNormals (implicit_shape_model.cpp:1034-1075): points whose normal holds a NaN in ANY component leave
    their cloud, order kept; empty objects, an object that loses everything and objects longer than one scan block are covered."""
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(11)
    off = np.array([0, 700, 700, 705, 1500], np.uint32)
    n = int(off[-1])
    a = rng.random((n, 6)).astype(np.float32)
    rgba = rng.integers(0, 1 << 24, n).astype(np.int32)
    bad = rng.random(n) < 0.2
    bad[700:705] = True                                  # the third object loses every point
    comp = rng.integers(3, 6, n)
    a[bad, comp[bad]] = np.nan
    for with_rgba in (True, False):
        cols = [T(np.ascontiguousarray(a[:, j]), dev) for j in range(6)]
        got = pkg.capi.filter_normals(ctx, off, *cols, rgba=T(rgba, dev) if with_rgba else None)
        keep = ~bad
        want_off = [0] + [int(keep[:e].sum()) for e in off[1:]]
        assert got[0].tolist() == want_off
        for j in range(6):
            np.testing.assert_array_equal(got[1 + j].cpu().numpy(), a[keep, j])
        if with_rgba:
            np.testing.assert_array_equal(got[7].cpu().numpy(), rgba[keep])
        else:
            assert got[7] is None
    with pytest.raises(pkg.capi.IsmHipError, match="must not alias"):
        x = T(np.ascontiguousarray(a[:, 0]), dev)
        arr = pkg.capi._PointArrays(*([x.data_ptr()] * 6), None)
        new = np.zeros(len(off), np.uint32)
        ctx.check(pkg.capi.lib().ismhip_filter_normals(ctx._h, 4, pkg.capi._p(off), __import__("ctypes").byref(arr), __import__("ctypes").byref(arr), pkg.capi._p(new)), "ismhip_filter_normals")


# ------------------------------------------------------------------------------------------------ kNN
def _cb(pkg, gpu, words, n_classes=4, votes_per_word=1, seed=0):
    rng = np.random.default_rng(seed)
    n = len(words)
    off = np.arange(n + 1, dtype=np.uint32) * votes_per_word
    nv = int(off[-1])
    host = dict(words=np.asarray(words, np.float32), vote_offsets=off, vote_xyz=rng.normal(size=(nv, 3)).astype(np.float32),
                vote_class=rng.integers(0, n_classes, nv).astype(np.uint32), vote_instance=rng.integers(0, 9, nv).astype(np.uint32),
                class_sigma=rng.uniform(0.2, 2.0, n_classes).astype(np.float32), word_weight=rng.uniform(0.5, 1.5, n).astype(np.float32),
                vote_weight=rng.uniform(0.5, 1.5, nv).astype(np.float32), vote_class_weight=rng.uniform(0.1, 1, nv).astype(np.float32),
                vote_bbox_quat=rng.normal(size=(nv, 4)).astype(np.float32), vote_bbox_size=rng.uniform(0.1, 1, (nv, 3)).astype(np.float32))
    ctx, dev = gpu
    dev_cb = pkg.capi.Codebook(ctx, host["words"], off, host["vote_xyz"], host["vote_class"], host["vote_instance"], n_classes,
                               host["class_sigma"], word_weight=host["word_weight"], vote_weight=host["vote_weight"],
                               vote_class_weight=host["vote_class_weight"], vote_bbox_quat=host["vote_bbox_quat"],
                               vote_bbox_size=host["vote_bbox_size"])
    return host, dev_cb


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("shape", [(1000, 352, 777), (130, 33, 65), (5000, 1344, 300), (3, 16, 10)])
@pytest.mark.parametrize("k", [1, 2, 3])
def test_knn_matches_oracle(pkg, gpu, ora, metric, shape, k):
    ctx, dev = gpu
    n_words, dim, nq = shape
    rng = np.random.default_rng(n_words + dim + k)
    words = rng.random((n_words, dim)).astype(np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    q = rng.random((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[: min(5, n_words, nq)] = words[: min(5, n_words, nq)]               # exact hits: distance 0
    host, cb = _cb(pkg, gpu, words)
    idx, dist = pkg.capi.knn(ctx, cb, metric, T(q, dev), k)
    widx, wdist = ora.knn(metric, words, q, k)
    assert np.array_equal(idx.cpu().numpy(), widx)                           # bit-exact indices
    got = dist.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(wdist))
    assert np.array_equal(got[~np.isnan(got)], wdist[~np.isnan(wdist)])    # same functor, same summation order: bit-exact


@pytest.mark.parametrize("metric", [0, 1])
def test_knn_adversarial_needs_the_exact_fallback(pkg, gpu, ora, metric):
    """Candidate generation keeps 2 entries per lane slot; these codebooks put hundreds of near-duplicates (and exact duplicates)
    inside the fp32 contraction error of the best match, and use un-normalised magnitudes (|c|^2 ~ 1e4) where the error bound is
    large. The proof step must notice and the exact scan must still return the oracle's answer, ties to the lowest row."""
    ctx, dev = gpu
    rng = np.random.default_rng(5 + metric)
    base = rng.random((40, 96)).astype(np.float32) * 30.0
    words = np.repeat(base, 60, axis=0)                                   # 60 copies of each of 40 prototypes ...
    words += (rng.random(words.shape) < 0.02) * 1e-4 * rng.random(words.shape)   # ... some of them perturbed in the 6th digit
    words = words.astype(np.float32)
    perm = rng.permutation(len(words)); words = words[perm]
    q = (base[rng.integers(0, 40, 300)] + 1e-3 * rng.random((300, 96))).astype(np.float32)
    host, cb = _cb(pkg, gpu, words)
    for k in (1, 3):
        idx, dist = pkg.capi.knn(ctx, cb, metric, T(q, dev), k)
        widx, wdist = ora.knn(metric, words, q, k)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(dist.cpu().numpy(), wdist)
    ms, n = ctx.timer("knn_fallback")


def test_knn_matches_oracle_on_clustered_unit_vectors(pkg, gpu, ora):
    """descriptor-like data: unit vectors in tight clusters (many near neighbours within 1e-3 of the best) — the regime where
    the bf16x3 candidate scores are least able to separate the winners and the proof/fallback has to work."""
    ctx, dev = gpu
    rng = np.random.default_rng(99)
    centres = rng.random((50, 352)).astype(np.float32)
    words = (np.repeat(centres, 80, axis=0) + 0.01 * rng.random((4000, 352))).astype(np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    q = (centres[rng.integers(0, 50, 1000)] + 0.01 * rng.random((1000, 352))).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    host, cb = _cb(pkg, gpu, words)
    for k in (1, 2, 4):
        idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
        widx, wdist = ora.knn(0, words, q, k)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(dist.cpu().numpy(), wdist)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("shape,k", [((6000, 352, 5000), 5), ((6000, 352, 5000), 16), ((300, 33, 200), 9), ((12, 16, 40), 16), ((3000, 96, 600), 8)])
def test_knn_more_than_four_neighbours(pkg, gpu, ora, metric, shape, k):
    """KNN activation with K > 4 (activation_strategy_knn.h:57-72 takes any K): the candidate slots keep at most 4 rows each, so a
    slot that held more of the true K best fails its proof and the exact scan of that slot finishes the query. Bit-exact against the
    oracle, descriptor-like clustered rows included (several of the K best in one slot); n_words < K pads with -1 / NaN."""
    ctx, dev = gpu
    n_words, dim, nq = shape
    rng = np.random.default_rng(n_words + dim + k + metric)
    centres = rng.random((max(2, n_words // 40), dim)).astype(np.float32)
    words = (centres[rng.integers(0, len(centres), n_words)] + 0.05 * rng.random((n_words, dim))).astype(np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    q = (centres[rng.integers(0, len(centres), nq)] + 0.05 * rng.random((nq, dim))).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    words[7 % n_words] = words[3 % n_words]                               # a tie: the lower row first
    host, cb = _cb(pkg, gpu, words)
    ctx.timers_enable(True); ctx.timers_reset()
    idx, dist = pkg.capi.knn(ctx, cb, metric, T(q, dev), k)
    ctx.sync()
    print(f"k={k} metric={metric} {shape}: unproven queries / slots {_knn_flagged(ctx)}")
    ctx.timers_enable(False)
    widx, wdist = ora.knn(metric, words, q, k)
    assert np.array_equal(idx.cpu().numpy(), widx)
    got = dist.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(wdist))
    assert np.array_equal(got[~np.isnan(got)], wdist[~np.isnan(wdist)])
    if n_words < k:
        assert (widx[:, n_words:] == -1).all()


@pytest.mark.parametrize("case", ["histograms", "negative_words", "negative_queries", "nan_query"])
def test_knn_chi2_sum_test_paths(pkg, gpu, ora, case):
    """The chi-square functor skips terms whose sum is not positive (utils/distance.h:65, FLANN ChiSquareDistance). The candidate
    kernel drops that test -- and uses the packed FP32 instructions -- only when neither the codebook nor the query batch holds a
    negative or NaN element; sparse histograms (many 0 + 0 terms) take that path, any negative element the exact one. Both must
    return the oracle's neighbours bit for bit, cancelling terms (c = -q) included."""
    ctx, dev = gpu
    rng = np.random.default_rng(17)
    words = rng.random((700, 96)).astype(np.float32)
    words[rng.random(words.shape) < 0.6] = 0.0                       # sparse histograms: most sums are 0 + 0
    q = (words[rng.integers(0, 700, 300)] + 0.05 * rng.random((300, 96)) * (rng.random((300, 96)) < 0.3)).astype(np.float32)
    if case == "negative_words":
        words[5, :10] = -words[5, :10] - 0.25
        q[0, :10] = -words[5, :10]                                   # sums of exactly 0 with a non-zero difference: skipped by the functor
    elif case == "negative_queries":
        q[3, 7] = -0.5; q[4, :] = -q[4, :]
    elif case == "nan_query":
        q[9, 2] = np.nan
    host, cb = _cb(pkg, gpu, words)
    for k in (1, 3):
        idx, dist = pkg.capi.knn(ctx, cb, 1, T(q, dev), k)
        widx, wdist = ora.knn(1, words, q, k)
        ok = ~np.isnan(wdist).any(1)                                 # a NaN query has no defined order
        assert np.array_equal(idx.cpu().numpy()[ok], widx[ok])
        assert np.array_equal(dist.cpu().numpy()[ok], wdist[ok])
    assert ok.sum() >= 299


def _knn_flagged(ctx):
    return int(ctx.timer("knn_flagged_queries")[0]), int(ctx.timer("knn_flagged_items")[0])


@pytest.mark.parametrize("dim", [352, 1344])
def test_knn_chi2_hellinger_candidates_match_oracle(pkg, gpu, ora, dim):
    """chi-square on histogram data: candidates from the squared-L2 MFMA kernels on sqrt images (chi2 >= |sqrt q - sqrt c|^2), exact
    chi-square re-rank + proof (k_knn_rerank_hell), unproven queries through the VALU kernel. Sparse non-negative rows with exact zeros
    (sum == 0 terms), duplicates (ties to the lowest row), exact hits; a batch with ONE negative element must keep the VALU path and
    still match. Bit-exact against the oracle's functor."""
    ctx, dev = gpu
    rng = np.random.default_rng(dim)
    n_words, nq = 6000, 3000
    protos = rng.random((50, dim)).astype(np.float32) ** 3
    def draw(n):
        x = protos[rng.integers(0, 50, n)] * (0.5 + rng.random((n, dim)).astype(np.float32)) * (rng.random((n, dim)) > 0.4)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    words, q = draw(n_words), draw(nq)
    words[3000:3004] = words[11]
    q[:6] = words[:6]; q[6] = words[11]
    host, cb = _cb(pkg, gpu, words)
    ctx.timers_enable(True)
    try:
        for k in (1, 2, 3):
            idx, dist = pkg.capi.knn(ctx, cb, 1, T(q, dev), k)
            n2 = int(ctx.timer("knn_stage2_queries")[0])
            widx, wdist = ora.knn(1, words, q, k)
            assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(dist.cpu().numpy(), wdist)
            assert n2 < nq // 2, n2                                       # the Hellinger stage proved the bulk
        qn = q.copy(); qn[17, 5] = -1e-3
        idx, dist = pkg.capi.knn(ctx, cb, 1, T(qn, dev), 1)
        widx, wdist = ora.knn(1, words, qn, 1)
        assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(dist.cpu().numpy(), wdist)
    finally:
        ctx.timers_enable(False)


def test_knn_few_unproven_slots_take_the_split_scan(pkg, gpu, ora):
    """Six identical copies of one codeword inside ONE candidate slot (rows 0-3, 8, 9 of the first tile: same wave-row block, same
    accumulator half) overflow the slot's top-4 with equal scores, so the proof fails for exactly that (query, slot) pair. With so
    few unproven items the exact scan cuts every item into row ranges; results must still be the oracle's (lowest row on ties)."""
    ctx, dev = gpu
    rng = np.random.default_rng(123)
    words = rng.random((3000, 352)).astype(np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    for r in (1, 2, 3, 8, 9):
        words[r] = words[0]
    words[2500] = words[0]                                              # and one more copy far away (another slot)
    q = rng.random((500, 352)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[7] = words[0] + 1e-4 * rng.random(352).astype(np.float32)
    q[8] = words[0]
    host, cb = _cb(pkg, gpu, words)
    ctx.timers_enable(True)
    try:
        for k in (1, 2, 3, 4):
            idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
            gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
            nfq, nfi = _knn_flagged(ctx)
            widx, wdist = ora.knn(0, words, q, k)
            assert np.array_equal(gi, widx) and np.array_equal(gd, wdist)
            assert 1 <= nfq <= 8 and nfi >= 1, (nfq, nfi)              # the construction really exercised the few-items path
    finally:
        ctx.timers_enable(False)


def test_knn_two_stage_search_matches_oracle(pkg, gpu, ora):
    """Big squared-L2 launches with k <= 2 take the two-stage search (T = 2 over all queries, T = 4 + merged splits for the
    queries stage 1 cannot prove, exact scan of ALL rows for what stage 2 cannot prove). The codebook holds clusters of 3, 6
    and 20 bit-identical / near-identical rows, each inside one stage-1 lane slot: 3 overflow its top-2, 20 also overflow the merged
    stage-2 slot (16 kept), so all three endings are exercised; answers must be the oracle's, ties to the lowest row."""
    ctx, dev = gpu
    rng = np.random.default_rng(2024)
    words = rng.random((8192, 96)).astype(np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    # rows 16 m + 4 fq + j (m < 8, j < 4) of a 256-row tile share ONE stage-1 lane slot (wave row 0, slot fq): put every cluster
    # into one slot of its tile so that it overflows the slot's top-2
    proto = [260, 1024 + 8, 4096]
    for p_, n in zip(proto, (3, 6, 20)):
        rows = np.asarray([p_ + 16 * (i // 4) + (i % 4) for i in range(n)])
        words[rows] = words[p_]
        words[rows[1::2]] += (1e-7 * rng.random((len(rows[1::2]), 96))).astype(np.float32)    # half of the copies differ in the last bits
    q = rng.random((6000, 96)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    for j, p_ in enumerate(proto):
        q[100 * j:100 * j + 40] = words[p_] + (1e-4 * rng.random((40, 96))).astype(np.float32)
    q[777] = words[proto[2]]
    host, cb = _cb(pkg, gpu, words)
    ctx.timers_enable(True)
    try:
        for k in (1, 2):
            idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
            gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
            n2 = int(ctx.timer("knn_stage2_queries")[0]); nfq, _ = _knn_flagged(ctx)
            widx, wdist = ora.knn(0, words, q, k)
            assert np.array_equal(gi, widx) and np.array_equal(gd, wdist)
            assert n2 >= 41 and nfq >= 1, (n2, nfq)                      # stage 2 ran, and it left work for the exact scan
            assert n2 < 2000                                             # ... but stage 1 proved the bulk
    finally:
        ctx.timers_enable(False)


@pytest.mark.parametrize("mode", ["f16", "bf16x3", "f32"])
@pytest.mark.parametrize("scale", [1.0, 1e-30, 3e-6, 250.0, 1e20])
def test_knn_candidate_modes_and_magnitudes(pkg, gpu, ora, mode, scale, monkeypatch):
    """All three squared-L2 candidate kernels (f16 default, bf16x3, exact f32) must return the oracle's answer for descriptors of any
    magnitude: the f16 image is rescaled by a power of two, tiny and huge values may neither underflow nor overflow it silently."""
    _, dev = gpu
    monkeypatch.setenv("ISMHIP_KNN_MODE", mode)
    ctx = pkg.capi.Ctx(0)                                               # the mode is read when a context is created
    rng = np.random.default_rng(17)
    words = (rng.random((2000, 100)) ** 4).astype(np.float32)           # wide dynamic range inside each descriptor
    q = (words[rng.integers(0, 2000, 400)] + 0.02 * rng.random((400, 100))).astype(np.float32)
    q[:50] = (rng.random((50, 100)) ** 4).astype(np.float32)
    words = (words * np.float32(scale)).astype(np.float32); q = (q * np.float32(scale)).astype(np.float32)
    off = np.arange(len(words) + 1, dtype=np.uint32)
    cb = pkg.capi.Codebook(ctx, words, off, np.zeros((len(words), 3), np.float32), np.zeros(len(words), np.uint32),
                           np.zeros(len(words), np.uint32), 1, np.ones(1, np.float32))
    ctx.timers_enable(True)
    for k in (1, 3):
        idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
        gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
        widx, wdist = ora.knn(0, words, q, k)
        assert np.array_equal(gi, widx)
        assert np.array_equal(gd, wdist)
    if scale in (1.0, 250.0):                                            # ordinary magnitudes must not lean on the exact scan
        assert _knn_flagged(ctx)[0] <= 40, _knn_flagged(ctx)


def test_knn_f16_query_with_outlier_magnitude(pkg, gpu, ora):
    """One query 1e6 times larger than the rest sets the batch's f16 scale; the small queries lose precision in the f16 image
    (absolute error term of the bound) and must be caught by the proof, a query with inf must not poison the others."""
    ctx, dev = gpu
    rng = np.random.default_rng(3)
    words = rng.random((1500, 64)).astype(np.float32)
    q = (words[rng.integers(0, 1500, 200)] + 0.01 * rng.random((200, 64))).astype(np.float32)
    q[5] *= 1e6
    host, cb = _cb(pkg, gpu, words)
    idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), 2)
    widx, wdist = ora.knn(0, words, q, 2)
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(dist.cpu().numpy(), wdist)
    q[9, 3] = np.inf
    idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), 1)
    widx, wdist = ora.knn(0, words, q, 1)
    ok = np.arange(200) != 9
    assert np.array_equal(idx.cpu().numpy()[ok], widx[ok]) and np.array_equal(dist.cpu().numpy()[ok], wdist[ok])


@pytest.mark.parametrize("dim", [352, 33, 16, 100])
def test_knn_large_launch_matches_oracle(pkg, gpu, ora, dim):
    """>= 4096 queries against >= 4096 codewords selects the 256x256 LDS-DMA ring kernel (k_knn_l2_ring); ragged sizes leave a partly
    empty last query tile and padded codeword rows, dim 33 a half-empty last slice, dim 16 a single slice per tile."""
    ctx, dev = gpu
    rng = np.random.default_rng(dim)
    n_words, nq = 4096 + 300, 4096 + 77
    centres = rng.random((64, dim)).astype(np.float32)
    words = (centres[rng.integers(0, 64, n_words)] + 0.05 * rng.random((n_words, dim))).astype(np.float32)
    q = (centres[rng.integers(0, 64, nq)] + 0.05 * rng.random((nq, dim))).astype(np.float32)
    q[:10] = words[-10:]
    host, cb = _cb(pkg, gpu, words)
    for k in (1, 3):
        idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
        widx, wdist = ora.knn(0, words, q, k)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(dist.cpu().numpy(), wdist)


def _steep_spectrum_data(rng, n_words, nq, dim, rank=40, noise=0.02):
    """descriptor-like vectors: a low-rank part + small isotropic noise, non-negative, unit length"""
    basis = rng.random((rank, dim)).astype(np.float32)
    def draw(n):
        x = rng.random((n, rank)).astype(np.float32) ** 3 @ basis + noise * rng.random((n, dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    return draw(n_words), draw(nq)


def _bare_cb(pkg, ctx, words):
    n = len(words)
    return pkg.capi.Codebook(ctx, words, np.arange(n + 1, dtype=np.uint32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32),
                             np.zeros(n, np.uint32), 1, np.ones(1, np.float32))


@pytest.mark.parametrize("forced_m", [None, 96, 256])
def test_knn_rotated_stage1_matches_oracle(pkg, gpu, ora, forced_m, monkeypatch):
    """Stage 1 of the two-stage search on the rotated, truncated image (csrc/pca.hip): partial distances over the leading principal
    coordinates are lower bounds, the re-rank evaluates candidates in bound order and the proof divides by sigma_max(R)^2. Data
    with a steep spectrum takes the path by itself (forced_m None); 96 and 256 are forced truncations, 256 the rotation kernel's widest tile. Answers: the oracle's, bit for bit."""
    _, dev = gpu
    if forced_m is not None:
        monkeypatch.setenv("ISMHIP_KNN_PCA_M", str(forced_m))
    ctx = pkg.capi.Ctx(0)
    rng = np.random.default_rng(99)
    words, q = _steep_spectrum_data(rng, 8192 + 100, 5000, 352)
    words[4000:4003] = words[17]                                          # duplicates: ties to the lowest row
    q[:8] = words[:8]; q[8] = words[17]
    q[9] *= 40.0                                                          # far beyond the fixed query scale's headroom: f16 image overflows
    q[10] *= 1e-6
    cb = _bare_cb(pkg, ctx, words)
    assert cb.stage1_dims == (forced_m or cb.stage1_dims) and 0 < cb.stage1_dims <= 256 and cb.stage1_dims % 32 == 0, cb.stage1_dims
    ctx.timers_enable(True)
    for k in (1, 2):
        idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
        gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
        n2 = int(ctx.timer("knn_stage2_queries")[0])
        widx, wdist = ora.knn(0, words, q, k)
        assert np.array_equal(gi, widx) and np.array_equal(gd, wdist)
        assert n2 >= 1                                                    # at least the overflowing query went to stage 2
        if forced_m is None:
            assert n2 < 500, n2                                           # the chosen truncation proves the bulk
    assert int(ctx.timer("knn_pca_launches")[0]) == 2
    cb.close()


def test_knn_rotated_stage1_on_a_flat_spectrum_falls_back(pkg, gpu, ora, monkeypatch):
    """Random vectors have no leading directions: by itself the codebook gets no rotated image; FORCED onto 192 of 352 coordinates
    the partial distances bound almost nothing, nearly every stage-1 proof fails and stage 2 / the exact scan (original coordinates)
    answer -- slower, never wrong."""
    _, dev = gpu
    rng = np.random.default_rng(5)
    words = rng.normal(size=(6000, 352)).astype(np.float32)
    q = rng.normal(size=(4500, 352)).astype(np.float32)
    q[:5] = words[:5]
    ctx = pkg.capi.Ctx(0)
    cb = _bare_cb(pkg, ctx, words)
    assert cb.stage1_dims == 0                                            # spectrum too flat to pay
    cb.close()
    monkeypatch.setenv("ISMHIP_KNN_PCA_M", "192")
    ctx = pkg.capi.Ctx(0)
    cb = _bare_cb(pkg, ctx, words)
    assert cb.stage1_dims == 192 and cb.stage1_energy < 0.7
    ctx.timers_enable(True)
    idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), 2)
    n2 = int(ctx.timer("knn_stage2_queries")[0])
    widx, wdist = ora.knn(0, words, q, 2)
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(dist.cpu().numpy(), wdist)
    assert n2 > 2000, n2                                                  # the forced truncation really was useless, and it showed
    cb.close()


def test_knn_short_descriptors_take_the_f16_ring_with_identity_rotation(pkg, gpu, ora):
    """FPFH-like rows (33 values up to 100, three blocks summing to 100) in a codebook big enough for the ring kernel: the stage-1
    image is the f16 image itself (R = I, 64 padded coordinates) with the distance-side error bound of csrc/pca.hip instead of the
    exact-f32 MFMA contraction. Answers bit-equal to the oracle; the bulk proven in stage 1."""
    ctx, dev = gpu
    rng = np.random.default_rng(33)
    protos = rng.random((80, 33)).astype(np.float32) ** 2
    def draw(n):
        x = protos[rng.integers(0, 80, n)] * (0.6 + 0.8 * rng.random((n, 33)).astype(np.float32))
        x = x.reshape(n, 3, 11); x = 100.0 * x / x.sum(-1, keepdims=True)
        return x.reshape(n, 33).astype(np.float32)
    words, q = draw(9000), draw(5000)
    q[:4] = words[:4]
    cb = _bare_cb(pkg, ctx, words)
    assert cb.stage1_dims == 64 and cb.stage1_energy == 1.0
    ctx.timers_enable(True)
    try:
        for k in (1, 2):
            idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
            n2 = int(ctx.timer("knn_stage2_queries")[0])
            widx, wdist = ora.knn(0, words, q, k)
            assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(dist.cpu().numpy(), wdist)
            assert n2 < 1000, n2
    finally:
        ctx.timers_enable(False)
    cb.close()


def test_knn_ties_kat_and_ratio(pkg, gpu, ora):
    ctx, dev = gpu
    k = KAT["knn_ties"]
    host, cb = _cb(pkg, gpu, k["words"])
    idx, dist = pkg.capi.knn(ctx, cb, 0, T(np.asarray(k["q"], np.float32), dev), k["k"])
    assert idx.cpu().numpy().tolist() == k["expected_idx"]
    np.testing.assert_allclose(dist.cpu().numpy(), k["expected_l2"], atol=1e-6)
    rng = np.random.default_rng(11)
    words = rng.random((500, 64)).astype(np.float32)
    q = words[:100] + 0.05 * rng.random((100, 64)).astype(np.float32)
    q[50:] = rng.random((50, 64)).astype(np.float32)
    host, cb = _cb(pkg, gpu, words)
    for metric in (0, 1):
        gi, gd = pkg.capi.knn_ratio(ctx, cb, metric, T(q, dev), 0.8)
        wi, wd = ora.knn_ratio(metric, words, q, 0.8)
        assert np.array_equal(gi.cpu().numpy(), wi) and (wi == -1).any() and (wi >= 0).any()
        assert np.array_equal(gd.cpu().numpy(), wd)


# ------------------------------------------------------------------------------------------------ keypoints (the step in front of the path)
@pytest.mark.parametrize("leaf,color", [(0.1, False), (0.05, True), (0.37, True)])
def test_voxel_keypoints_match_oracle(pkg, gpu, ora, leaf, color):
    """KeypointsVoxelGrid (pcl::VoxelGrid centroids in ascending voxel index): counts, order and colours exact, positions to 1e-6
    relative (the device sums in 64-bit fixed point, the reference in float). Ragged batch with an empty object, non-finite
    points, negative coordinates and a single-point object."""
    ctx, dev = gpu
    rng = np.random.default_rng(int(leaf * 1000))
    sizes = [5000, 0, 1, 12345, 777]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(off[-1])
    xyz = (rng.random((n, 3)) * np.array([2.0, 1.0, 3.0]) - np.array([1.0, 0.2, 2.5])).astype(np.float32)
    xyz[off[3]:off[4]] *= 0.3                                             # a denser object
    xyz[17] = np.nan; xyz[int(off[3]) + 5, 1] = np.inf                    # ignored, as pcl::VoxelGrid skips non-finite points
    rgba = rng.integers(0, 1 << 24, n, dtype=np.uint32) if color else None
    ko, kx, ky, kz, kc = pkg.capi.voxel_keypoints(ctx, off, T(xyz[:, 0].copy(), dev), T(xyz[:, 1].copy(), dev), T(xyz[:, 2].copy(), dev), leaf,
                                                  rgba=T(rgba.view(np.int32), dev) if color else None)
    kx, ky, kz = kx.cpu().numpy(), ky.cpu().numpy(), kz.cpu().numpy()
    for o in range(len(sizes)):
        s, e = int(off[o]), int(off[o + 1])
        wx, wy, wz, wc = ora.voxel_grid(xyz[s:e, 0], xyz[s:e, 1], xyz[s:e, 2], leaf, rgba[s:e] if color else None)
        a, b = int(ko[o]), int(ko[o + 1])
        assert b - a == len(wx), (o, b - a, len(wx))
        for got, want in ((kx[a:b], wx), (ky[a:b], wy), (kz[a:b], wz)):
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
        if color:
            assert np.array_equal(kc.cpu().numpy()[a:b].view(np.uint32), wc)
    assert ko[2] - ko[1] == 0 and ko[3] - ko[2] == 1


@pytest.mark.parametrize("orientation", [0, 1])
def test_estimate_normals_pca(pkg, gpu, ora, orientation):
    """ConsistentNormalsMethod 0 / 1: PCA normals (pcl::NormalEstimationOMPWithEigVals) flipped towards the origin / away from the
    object's centroid. The oracle follows PCL's single-pass FLOAT covariance (E[ab] - E[a]E[b] on absolute coordinates) and analytic
    eigen33; the device accumulates centred moments in double and uses the Jacobi solver -- same covariance, better conditioned --
    so the normals agree to the precision of PCL's own arithmetic (2e-3 here), NaN pattern and orientation exactly."""
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(50 + orientation)
    o0 = make_cloud(rng, 4000, "ellipsoid")[0] + np.float32([1.5, -0.7, 0.4])       # off-origin: the origin is a real viewpoint
    o1 = make_cloud(rng, 3000, "sphere", noise=0.005)[0] * 0.8 + np.float32([0.1, 0.2, 2.5])
    o2 = np.array([[0, 0, 0], [5, 5, 5]], np.float32)                                # two isolated points: < 3 neighbours -> NaN
    objs = [o0, o1, o2]
    off = np.concatenate([[0], np.cumsum([len(o) for o in objs])]).astype(np.uint32)
    P = np.concatenate(objs).astype(np.float32)
    P[17] = np.nan
    t = [T(P[:, i].copy(), dev) for i in range(3)]
    zn = [torch.zeros(len(P), dtype=torch.float32, device=dev) for _ in range(3)]
    cloud = pkg.capi.Cloud(ctx, off, *t, *zn, 0.06)
    out = [torch.empty(len(P), dtype=torch.float32, device=dev) for _ in range(3)]
    pkg.capi.estimate_normals_pca(ctx, cloud, 0.15, orientation, *out)
    got = np.stack([a.cpu().numpy() for a in out], 1)
    want = ora.pca_normals(off, P[:, 0], P[:, 1], P[:, 2], 0.15, orientation)
    assert np.array_equal(np.isnan(got).any(1), np.isnan(want).any(1)) and np.isnan(got[-1]).all() and np.isnan(got[17]).all()
    m = ~np.isnan(want).any(1)
    dots = (got[m] * want[m]).sum(1)
    assert (dots > 0).mean() > 0.999                              # same orientation (a flip needs cos_theta ~ 0)
    ang = np.arccos(np.clip(np.abs(dots), 0, 1))
    assert np.quantile(ang, 0.995) < 2e-3, np.quantile(ang, 0.995)
    np.testing.assert_allclose(np.linalg.norm(got[m], axis=1), 1.0, atol=1e-5)
    # the ellipsoid's analytic normals: PCA normals are close to them and point away from its centre (orientation 1) ...
    c0 = o0.mean(0)
    if orientation == 1:
        assert ((got[:4000][~np.isnan(got[:4000]).any(1)] * (o0 - c0)[~np.isnan(got[:4000]).any(1)]).sum(1) > 0).mean() > 0.99
    else:                                                         # ... or face the origin (orientation 0)
        g0 = got[:4000]; ok = ~np.isnan(g0).any(1)
        assert ((g0[ok] * (-P[:4000][ok])).sum(1) >= 0).all()


def test_estimate_normals_from_shot_frames(pkg, gpu, ora):
    """ImplicitShapeModel::computeNormals, ConsistentNormalsMethod 2 (implicit_shape_model.cpp:1014-1018 -> utils/normal_orientation.cpp:
    48-110): PCA normals flipped to the origin first; normal = inverted z axis of the SHOT frame (radius NormalRadius) where the frame
    is valid; a point with an invalid frame (< 5 neighbours) keeps its PCA normal (NaN below 3 neighbours); and the reference's
    mis-indexed patch loop: with k invalid frames in an object, its points 0..k-1 carry the UNFLIPPED PCA normal instead.
    The cloud is created with zero normals; after the call its own (cell-sorted) normals must be the estimated ones too: SHOT-352
    computed on that cloud has to match the oracle fed with the same normals."""
    ctx, dev = gpu
    syn = pkg.synthetic
    ds = syn.Dataset(3, 3, split=0, n_points=3000, n_keypoints=64)
    nb = ds.batch(range(3))
    xyz = nb["xyz"].copy()
    po = nb["pt_off"]
    xyz[5] += 50.0                                                        # object 0: an isolated point: no frame, no PCA normal
    tri = np.float32([[0, 0, 0], [0.05, 0.01, 0], [0.01, 0.04, 0.02]])
    for j, at in enumerate((100, 200)):                                   # object 1: two far-off triplets: PCA normal, no frame
        xyz[po[1] + at:po[1] + at + 3] = xyz[po[1]:po[1] + 3].mean(0) + np.float32([1.5 + j, 1.0, -0.5]) + tri
    x, y, z = (T(xyz[:, i].copy(), dev) for i in range(3))
    zeros = [T(np.zeros(len(xyz), np.float32), dev) for _ in range(3)]
    r_n = 0.12
    cloud = pkg.capi.Cloud(ctx, po, x, y, z, zeros[0], zeros[1], zeros[2], 0.15)
    nx, ny, nz = pkg.capi.estimate_normals(ctx, cloud, r_n, *zeros)
    got = np.stack([nx.cpu().numpy(), ny.cpu().numpy(), nz.cpu().numpy()], 1)
    frames = ora.shot_lrf(po, xyz[:, 0], xyz[:, 1], xyz[:, 2], po, xyz[:, 0], xyz[:, 1], xyz[:, 2], r_n)
    pca = ora.pca_normals(po, xyz[:, 0], xyz[:, 1], xyz[:, 2], r_n, 0)
    raw = ora.pca_normals(po, xyz[:, 0], xyz[:, 1], xyz[:, 2], r_n, 2)
    bad = ~np.isfinite(frames[:, 0])
    want = np.where(bad[:, None], pca, -frames[:, 6:9])
    from_frame = ~bad
    k_inv = [int(bad[po[o]:po[o + 1]].sum()) for o in range(3)]
    assert k_inv[0] >= 1 and k_inv[1] >= 6, k_inv                        # the synthetic clouds have sparse spots of their own
    for o in range(3):
        want[po[o]:po[o] + k_inv[o]] = raw[po[o]:po[o] + k_inv[o]]
        from_frame[po[o]:po[o] + k_inv[o]] = False
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[5]).all()
    # an eigenvector of a near-degenerate covariance may flip or rotate in the last bits; all but a handful agree to 1e-4
    err = np.abs(got[from_frame] - want[from_frame]).max(1)
    assert (err < 1e-4).mean() > 0.999, (err > 1e-4).sum()
    # PCA-derived entries (kept fallbacks, and the first k of each object): PCL's float covariance limits the agreement
    pc = ~from_frame & ~np.isnan(want).any(1)
    assert pc.sum() >= 13
    dots = (got[pc] * want[pc]).sum(1)
    print("method 2: PCA-derived normals", int(pc.sum()), "of", len(want), "min |dot|", float(np.abs(dots).min()), "sign mismatches", int((dots < 0).sum()))
    assert (np.abs(dots) > 1 - 1e-4).mean() > 0.95 and (dots > 0).mean() > 0.95, np.sort(dots)[:10]
    # the triplets' normal is the normal of their plane
    nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0]); nrm /= np.linalg.norm(nrm)
    for at in (100, 200):
        assert (np.abs(got[po[1] + at:po[1] + at + 3] @ nrm) > 1 - 1e-4).all()
    # descriptors on the same cloud use the refreshed normals
    kp, ko = nb["kp"], nb["kp_off"]
    kx, ky, kz = (T(kp[:, i].copy(), dev) for i in range(3))
    lrf = pkg.capi.shot_lrf(ctx, cloud, ko, kx, ky, kz, 0.3)
    desc, _ = pkg.capi.shot352(ctx, cloud, ko, kx, ky, kz, lrf, 0.4, want_counts=True)
    wl = ora.shot_lrf(po, xyz[:, 0], xyz[:, 1], xyz[:, 2], ko, kp[:, 0], kp[:, 1], kp[:, 2], 0.3)
    gn = got.copy()                                                        # the oracle gets the DEVICE normals: isolates the descriptor stage
    wd, _ = ora.shot352(po, xyz[:, 0], xyz[:, 1], xyz[:, 2], gn[:, 0], gn[:, 1], gn[:, 2], ko, kp[:, 0], kp[:, 1], kp[:, 2], wl, 0.4)
    d = desc.cpu().numpy()
    assert np.array_equal(np.isnan(d).any(1), np.isnan(wd).any(1))
    m = ~np.isnan(wd).any(1)
    assert np.abs(d[m] - wd[m]).max() <= 1e-4


# ------------------------------------------------------------------------------------------------ votes + maxima
@pytest.mark.parametrize("flags", [0, 1, 2, 4, 8, 15])
def test_cast_votes_matches_oracle(pkg, gpu, ora, flags):
    ctx, dev = gpu
    rng = np.random.default_rng(20 + flags)
    words = rng.random((200, 32)).astype(np.float32)
    host, cb = _cb(pkg, gpu, words, votes_per_word=3, seed=flags)
    nq, k = 300, 2
    q = rng.random((nq, 32)).astype(np.float32)
    idx, dist = ora.knn(0, words, q, k)
    idx[::17, 1] = -1
    A = rng.normal(size=(nq, 3, 3)); Qm, _ = np.linalg.qr(A)
    Qm[np.linalg.det(Qm) < 0, 2] *= -1
    lrf = Qm.reshape(nq, 9).astype(np.float32)
    kp = rng.normal(size=(nq, 3)).astype(np.float32)
    got = pkg.capi.cast_votes(ctx, cb, flags, T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), T(idx, dev), T(dist, dev), want_bbox=True)
    want = ora.cast_votes(host, flags, lrf, kp[:, 0], kp[:, 1], kp[:, 2], idx, dist)
    for key in ("cls", "inst", "codeword"):
        assert np.array_equal(got[key].cpu().numpy(), want[key]), key
    assert (want["cls"] >= 0).any() and (want["cls"] < 0).any()
    for key in ("pos", "weight", "bbox_quat", "bbox_size"):
        np.testing.assert_allclose(got[key].cpu().numpy(), want[key], atol=1e-5, rtol=1e-5, err_msg=key)


def _vote_scene(rng, n_obj, n_classes, with_empty=True, big=()):
    pos, w, cls, inst, bs, off = [], [], [], [], [], [0]
    for o in range(n_obj):
        if with_empty and o == 1:
            off.append(off[-1]); continue
        n_blobs = rng.integers(1, 4) + (3 if o in big else 0)
        for b in range(n_blobs):
            c = rng.integers(0, n_classes)
            m = rng.integers(5, 120) * (12 if o in big else 1)
            centre = rng.uniform(-2, 2, 3)
            pos.append(centre + 0.15 * rng.normal(size=(m, 3))); w.append(rng.uniform(0.2, 1.0, m))
            cls.append(np.full(m, c)); inst.append(rng.integers(0, 4, m)); bs.append(rng.uniform(0.5, 1.5, (m, 3)))
        m = rng.integers(0, 40)                                   # clutter + slots without a vote
        pos.append(rng.uniform(-3, 3, (m, 3))); w.append(rng.uniform(0.2, 1.0, m)); cls.append(rng.integers(-1, n_classes, m))
        inst.append(rng.integers(0, 4, m)); bs.append(rng.uniform(0.5, 1.5, (m, 3)))
        off.append(off[-1] + sum(len(x) for x in pos) - off[-1])
    v = dict(pos=np.concatenate(pos).astype(np.float32), weight=np.concatenate(w).astype(np.float32), cls=np.concatenate(cls).astype(np.int32),
             inst=np.concatenate(inst).astype(np.int32), bbox_size=np.concatenate(bs).astype(np.float32))
    # shuffle inside every object so that classes interleave like real vote slots
    for o in range(n_obj):
        s, e = off[o], off[o + 1]
        p = s + rng.permutation(e - s)
        for key in v:
            v[key][s:e] = v[key][p]
    return np.asarray(off, np.uint32), v


def test_find_maxima_filter_and_class_bandwidths_match_oracle(pkg, gpu, ora):
    """MaxFilterType "Simple" (greedy NMS over all classes inside the bandwidth, maxima_handler.cpp:227-268) and per-class
    bandwidths (BinOrBandwidthType ObjectRadius / BoundingBoxMedian -> MaximaHandler::getSearchDistForClass, :509-521)"""
    ctx, dev = gpu
    rng = np.random.default_rng(77)
    off, v = _vote_scene(rng, 12, 5)
    # make different classes collide: copy every object's votes once more under another class, slightly shifted
    v2 = {k: np.concatenate([a, a]) for k, a in v.items()}
    n = len(v["weight"])
    v2["cls"][n:] = np.where(v["cls"] >= 0, (v["cls"] + 1) % 5, -1); v2["pos"][n:] += 0.05; v2["weight"][n:] *= 0.7
    off2 = np.concatenate([off, off[1:] + off[-1]]).astype(np.uint32)          # the copies form 12 further objects ...
    order = np.concatenate([np.r_[off[o]:off[o + 1], n + off[o]:n + off[o + 1]] for o in range(12)])   # ... interleave them per object instead
    v3 = {k: a[order] for k, a in v2.items()}
    off3 = (2 * off.astype(np.int64)).astype(np.uint32)
    tv = {k2: T(a, dev) for k2, a in v3.items()}
    for kw in (dict(max_filter=1), dict(class_bandwidth=[0.3, 0.5, 0.8, 0.4, 0.6]), dict(max_filter=1, class_bandwidth=[0.3, 0.5, 0.8, 0.4, 0.6])):
        kw = dict(n_classes=5, bandwidth=0.5, max_maxima=12, min_votes_threshold=2, **kw)
        got = pkg.capi.find_maxima(ctx, off3, tv, **kw)
        want = ora.find_maxima(off3, v3, **kw)
        assert np.array_equal(got["n"].cpu().numpy(), want["n"]) and np.array_equal(got["cls"].cpu().numpy(), want["cls"]), kw
        np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
        np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=2e-3)
    plain = ora.find_maxima(off3, v3, n_classes=5, bandwidth=0.5, max_maxima=12, min_votes_threshold=2)
    filt = ora.find_maxima(off3, v3, n_classes=5, bandwidth=0.5, max_maxima=12, min_votes_threshold=2, max_filter=1)
    assert filt["n"].sum() < plain["n"].sum()                                  # the filter did remove colliding maxima


@pytest.mark.parametrize("suppression,kernel", [(0, 0), (1, 0), (0, 1)])
def test_find_maxima_matches_oracle(pkg, gpu, ora, suppression, kernel):
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(30 + suppression + 2 * kernel)
    off, v = _vote_scene(rng, 12, 5)
    tv = {k2: T(a, dev) for k2, a in v.items()}
    kw = dict(n_classes=5, bandwidth=0.5, suppression=suppression, kernel=kernel, max_maxima=12, min_votes_threshold=2)
    got = pkg.capi.find_maxima(ctx, off, tv, **kw)
    want = ora.find_maxima(off, v, **kw)
    assert np.array_equal(got["n"].cpu().numpy(), want["n"])
    assert np.array_equal(got["cls"].cpu().numpy(), want["cls"])
    assert np.array_equal(got["inst"].cpu().numpy(), want["inst"])
    assert np.array_equal(got["n_votes"].cpu().numpy(), want["n_votes"])
    np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
    np.testing.assert_allclose(got["inst_weight"].cpu().numpy(), want["inst_weight"], atol=TOL)
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=2e-3)          # mean-shift stops at Threshold = 1e-3
    np.testing.assert_allclose(got["bbox_size"].cpu().numpy(), want["bbox_size"], atol=1e-3)
    np.testing.assert_allclose(got["class_score"].cpu().numpy(), want["class_score"], atol=TOL)
    assert want["n"][1] == 0 and want["n"].max() >= 2


def _with_quats(rng, v):
    """bbox quaternions per vote slot: a few base rotations per blob-ish neighbourhood + jitter, unit length"""
    n = len(v["weight"])
    base = rng.normal(size=(6, 4)); base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = base[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 4))
    q *= np.where(rng.random(n) < 0.5, -1.0, 1.0)[:, None]           # q and -q are the same rotation: the scatter matrix does not care
    v = dict(v); v["bbox_quat"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    return v


def _same_rotation(a, b, atol):
    """quaternions equal up to the global sign"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1))
    return bool((d < atol).all()), float(d.max())


@pytest.mark.parametrize("voting", ["meanshift", "hough"])
def test_average_rotation_matches_oracle_and_closed_form(pkg, gpu, ora, voting):
    """Voting.AverageRotation (voting.cpp:186-215 -> Utils::quatWeightedAverage, utils.cpp:617-665): per maximum the dominant
    eigenvector of sum w q q^T over its votes. HIP vs oracle on random scenes; and a closed form: two votes with the SAME weight and
    quaternions q1, q2 average to (q1 + q2) / |q1 + q2| (and one vote returns its own quaternion)."""
    ctx, dev = gpu
    rng = np.random.default_rng(61)
    off, v = _vote_scene(rng, 10, 4)
    v = _with_quats(rng, v)
    tv = {k2: T(a, dev) for k2, a in v.items()}
    if voting == "meanshift":
        kw = dict(n_classes=4, bandwidth=0.5, max_maxima=12, min_votes_threshold=2, average_rotation=True)
        got = pkg.capi.find_maxima(ctx, off, tv, **kw); want = ora.find_maxima(off, v, **kw)
    else:
        kw = dict(n_classes=4, bin_size=0.5, max_maxima=12, min_votes_threshold=2, rel_threshold=0.3, average_rotation=True)
        got = pkg.capi.hough3d_maxima(ctx, off, tv, **kw); want = ora.hough3d_maxima(off, v, **kw)
    assert np.array_equal(got["n"].cpu().numpy(), want["n"]) and want["n"].sum() > 10
    assert np.array_equal(got["cls"].cpu().numpy(), want["cls"])
    ok, err = _same_rotation(got["bbox_quat"].cpu().numpy(), want["bbox_quat"], 2e-4)
    assert ok, err
    # closed form
    q1 = np.asarray([0.9, 0.1, -0.3, 0.2]); q1 /= np.linalg.norm(q1)
    q2 = np.asarray([0.7, -0.4, 0.1, 0.5]); q2 /= np.linalg.norm(q2)
    v2 = dict(pos=np.asarray([[0, 0, 0], [0.01, 0, 0], [3, 3, 3]], np.float32), weight=np.asarray([0.5, 0.5, 1.0], np.float32),
              cls=np.asarray([0, 0, 1], np.int32), inst=np.zeros(3, np.int32), bbox_size=np.ones((3, 3), np.float32),
              bbox_quat=np.stack([q1, q2, q2]).astype(np.float32))
    off2 = np.asarray([0, 3], np.uint32)
    tv2 = {k2: T(a, dev) for k2, a in v2.items()}
    if voting == "meanshift":
        kw = dict(n_classes=2, bandwidth=0.5, max_maxima=4, min_votes_threshold=1, average_rotation=True, kernel=1)   # uniform kernel: equal weights stay equal
        outs = [pkg.capi.find_maxima(ctx, off2, tv2, **kw), ora.find_maxima(off2, v2, **kw)]
    else:
        kw = dict(n_classes=2, bin_size=1.0, max_maxima=4, min_votes_threshold=1, average_rotation=True, use_interpolation=False)
        outs = [pkg.capi.hough3d_maxima(ctx, off2, tv2, **kw), ora.hough3d_maxima(off2, v2, **kw)]
    mean = (q1 + q2) / np.linalg.norm(q1 + q2)
    for o_ in outs:
        cls = np.asarray(o_["cls"].cpu() if hasattr(o_["cls"], "cpu") else o_["cls"])[0]
        bq = np.asarray(o_["bbox_quat"].cpu() if hasattr(o_["bbox_quat"], "cpu") else o_["bbox_quat"])[0]
        assert sorted(cls[:2].tolist()) == [0, 1]
        assert _same_rotation(bq[list(cls[:2]).index(0)], mean, 1e-5)[0]
        assert _same_rotation(bq[list(cls[:2]).index(1)], q2, 1e-5)[0]


@pytest.mark.parametrize("som", [1, 2, 3])
def test_single_object_max_types_match_oracle(pkg, gpu, ora, som):
    """SingleObjectMode with SingleObjectMaxType BANDWIDTH / MODEL_RADIUS / COMPLETE_VOTING_SPACE (voting_mean_shift.cpp:124-157,
    single_object_mode_helper.cpp:15-40): no mean shift, one maximum per class at the centroid of the object's cloud, density and
    reweighting with the type's bandwidth. Centroid and model radius come from the device cloud (ismhip_cloud_centroids / _radii)
    and are checked against numpy first."""
    ctx, dev = gpu
    rng = np.random.default_rng(70 + som)
    off, v = _vote_scene(rng, 8, 4)
    objs = [make_cloud(rng, 1200 + 100 * o, ("sphere", "ellipsoid", "plane")[o % 3], noise=0.01) for o in range(8)]
    objs = [((p + rng.uniform(-1, 1, 3)).astype(np.float32), n_) for p, n_ in objs]            # off-centre clouds
    sc = Scene(pkg, gpu, objs, [p[:8] for p, _ in objs], 0.15)
    cen = pkg.capi.cloud_centroids(ctx, sc.cloud, dev)
    rad = pkg.capi.cloud_radii(ctx, sc.cloud, cen)
    cen_h, rad_h = cen.cpu().numpy(), rad.cpu().numpy()
    for o in range(8):
        s, e = sc.pt_off[o], sc.pt_off[o + 1]
        c64 = sc.p[s:e].astype(np.float64).mean(0)
        assert np.abs(cen_h[o] - c64).max() < 1e-5
        d = np.sqrt(((sc.p[s:e] - cen_h[o]) ** 2).sum(1, dtype=np.float32))
        assert abs(rad_h[o] - d.max()) < 1e-5 * max(1.0, d.max())
    tv = {k2: T(a, dev) for k2, a in v.items()}
    kw = dict(n_classes=4, bandwidth=0.8, max_maxima=8, min_votes_threshold=1, single_object_max_type=som)
    got = pkg.capi.find_maxima(ctx, off, tv, object_centroid=cen, object_radius=rad, **kw)
    want = ora.find_maxima(off, v, object_centroid=cen_h, object_radius=rad_h, **kw)
    assert np.array_equal(got["n"].cpu().numpy(), want["n"])
    assert np.array_equal(got["cls"].cpu().numpy(), want["cls"]) and np.array_equal(got["inst"].cpu().numpy(), want["inst"])
    assert np.array_equal(got["n_votes"].cpu().numpy(), want["n_votes"])
    np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=1e-6)
    np.testing.assert_allclose(got["bbox_size"].cpu().numpy(), want["bbox_size"], atol=1e-3)
    # every maximum sits at its object's centroid, at most one per class that has votes (a class whose votes all lie outside the
    # type's bandwidth gets none)
    n = want["n"]
    assert n.sum() >= 4
    for o in range(8):
        assert n[o] <= len(set(v["cls"][off[o]:off[o + 1]][v["cls"][off[o]:off[o + 1]] >= 0].tolist()))
        assert len(set(want["cls"][o, :n[o]].tolist())) == n[o]
        for m in range(n[o]):
            assert np.abs(want["pos"][o, m] - cen_h[o]).max() < 1e-6


def test_max_filter_merge_matches_oracle(pkg, gpu, ora):
    """MaxFilterType "Merge" (maxima_handler.cpp:300-440): neighbourhoods of maxima closer than the first one's search distance are
    merged per class (running weighted means, instance tallies, quaternion average) and replaced by the heaviest merged maximum.
    Scenes with blobs of DIFFERENT classes on top of each other and same-class blobs just outside the intra-class suppression."""
    ctx, dev = gpu
    rng = np.random.default_rng(83)
    pos, w, cls, inst, bs, off = [], [], [], [], [], [0]
    for o in range(12):
        for b in range(rng.integers(2, 4)):
            centre = rng.uniform(-1.5, 1.5, 3)
            for c in rng.choice(5, size=rng.integers(1, 4), replace=False):      # several classes vote for (almost) the same place
                m = rng.integers(8, 60)
                pos.append(centre + rng.uniform(-0.2, 0.2, 3) + 0.08 * rng.normal(size=(m, 3))); w.append(rng.uniform(0.2, 1.0, m))
                cls.append(np.full(m, c)); inst.append(rng.integers(0, 3, m)); bs.append(rng.uniform(0.5, 1.5, (m, 3)))
        off.append(sum(len(x) for x in pos))
    v = dict(pos=np.concatenate(pos).astype(np.float32), weight=np.concatenate(w).astype(np.float32), cls=np.concatenate(cls).astype(np.int32),
             inst=np.concatenate(inst).astype(np.int32), bbox_size=np.concatenate(bs).astype(np.float32))
    v = _with_quats(rng, v)
    off = np.asarray(off, np.uint32)
    tv = {k2: T(a, dev) for k2, a in v.items()}
    cbw = np.asarray([0.5, 0.6, 0.4, 0.5, 0.7], np.float32)
    for class_bandwidth in (None, cbw):
        kw = dict(n_classes=5, bandwidth=0.5, max_maxima=16, min_votes_threshold=2, max_filter=2, average_rotation=True, class_bandwidth=class_bandwidth)
        got = pkg.capi.find_maxima(ctx, off, tv, **kw)
        want = ora.find_maxima(off, v, **kw)
        plain = ora.find_maxima(off, v, **dict(kw, max_filter=0))
        assert want["n"].sum() < plain["n"].sum()                                # neighbourhoods really were merged away
        assert np.array_equal(got["n"].cpu().numpy(), want["n"])
        assert np.array_equal(got["cls"].cpu().numpy(), want["cls"]) and np.array_equal(got["inst"].cpu().numpy(), want["inst"])
        assert np.array_equal(got["n_votes"].cpu().numpy(), want["n_votes"])
        np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
        np.testing.assert_allclose(got["inst_weight"].cpu().numpy(), want["inst_weight"], atol=TOL)
        np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=2e-3)
        np.testing.assert_allclose(got["bbox_size"].cpu().numpy(), want["bbox_size"], atol=1e-3)
        ok, err = _same_rotation(got["bbox_quat"].cpu().numpy(), want["bbox_quat"], 5e-4)
        assert ok, err


def test_maxima_of_objects_with_more_slots_than_fit_lds(pkg, gpu, ora):
    """More than 2048 vote slots in one object (codewords with many votes, K > 1): the per-class vote arrays move from LDS to a
    workspace in HBM, laid out per (object, class) from a class tally. Same oracle, same tolerances as the LDS-resident kernels;
    small and empty objects ride in the same call."""
    ctx, dev = gpu
    rng = np.random.default_rng(314)
    off, v = _vote_scene(rng, 6, 4, big=(0, 3))
    sizes = np.diff(off.astype(np.int64))
    assert sizes.max() > 2048 and sizes[1] == 0 and sizes[2] < 600, sizes
    tv = {k2: T(a, dev) for k2, a in v.items()}
    kw = dict(n_classes=4, bandwidth=0.5, max_maxima=16, min_votes_threshold=2)
    got = pkg.capi.find_maxima(ctx, off, tv, **kw)
    want = ora.find_maxima(off, v, **kw)
    for key in ("n", "cls", "inst", "n_votes"):
        assert np.array_equal(got[key].cpu().numpy(), want[key]), key
    np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
    np.testing.assert_allclose(got["inst_weight"].cpu().numpy(), want["inst_weight"], atol=TOL)
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=2e-3)
    np.testing.assert_allclose(got["class_score"].cpu().numpy(), want["class_score"], atol=TOL)
    assert want["n"][0] >= 2 and want["n"][3] >= 2
    kw = dict(n_classes=4, bin_size=0.25, use_interpolation=True, rel_threshold=0.2, max_maxima=16, min_votes_threshold=2)
    got = pkg.capi.hough3d_maxima(ctx, off, tv, **kw)
    want = ora.hough3d_maxima(off, v, **kw)
    for key in ("n", "cls", "inst", "n_votes"):
        assert np.array_equal(got[key].cpu().numpy(), want[key]), key
    np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=1e-4)
    np.testing.assert_allclose(got["class_score"].cpu().numpy(), want["class_score"], atol=TOL)
    assert want["n"][0] >= 1


def test_maxima_caps_are_reported_by_sync(pkg, gpu):
    """More than 128 maxima of one class in one object (200 far-apart single-vote blobs): the kernels keep the first 128 and
    ismhip_sync returns ISMHIP_ERR_UNSUPPORTED once -- the reference has no cap, so a truncation must not pass silently."""
    ctx, dev = gpu
    n = 200
    pos = np.stack([np.arange(n) * 5.0, np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    v = dict(pos=T(pos, dev), weight=T(np.ones(n, np.float32), dev), cls=T(np.zeros(n, np.int32), dev), inst=T(np.zeros(n, np.int32), dev))
    out = pkg.capi.find_maxima(ctx, [0, n], v, n_classes=2, bandwidth=0.5, max_maxima=256)
    with pytest.raises(pkg.capi.IsmHipError, match="128 maxima per class"):
        ctx.sync()
    ctx.sync()                                                     # reported once, then cleared
    assert int(out["n"][0]) == 128
    out = pkg.capi.find_maxima(ctx, [0, 100], {k: a[:100] for k, a in v.items()}, n_classes=2, bandwidth=0.5, max_maxima=256)
    ctx.sync()
    assert int(out["n"][0]) == 100


def test_find_maxima_two_blobs_property(pkg, gpu):
    ctx, dev = gpu
    rng = np.random.default_rng(40)
    centres = np.array([[0.0, 0, 0], [3.0, 0, 0]])
    pos = np.concatenate([c + 0.05 * rng.normal(size=(60, 3)) for c in centres]).astype(np.float32)
    v = dict(pos=T(pos, dev), weight=T(np.ones(120, np.float32), dev), cls=T(np.full(120, 1, np.int32), dev), inst=T(np.zeros(120, np.int32), dev))
    out = pkg.capi.find_maxima(ctx, [0, 120], v, n_classes=3, bandwidth=0.5, max_maxima=8)
    assert int(out["n"][0]) == 2
    p = out["pos"][0, :2].cpu().numpy(); p = p[np.argsort(p[:, 0])]
    np.testing.assert_allclose(p, centres, atol=0.03)
    assert abs(float(out["weight"][0, :2].sum()) - 1) < 1e-6


# ------------------------------------------------------------------------------------------------ end to end, full sizes
def test_end_to_end_full_size_properties_and_oracle_sample(pkg, gpu, ora):
    """BASELINE configs[1] sizes (16384 points, 1024 keypoints): properties on all objects, oracle comparison on one."""
    import torch
    ctx, dev = gpu
    capi, pipeline, synthetic = pkg.capi, pkg.pipeline, pkg.synthetic
    cfg = pipeline.IsmConfig(n_classes=4)
    train = synthetic.Dataset(4, 8, split=0)
    test = synthetic.Dataset(4, 4, split=1)
    rec = pipeline.Recognizer(ctx, cfg)
    order = sorted(range(8), key=lambda i: (train.label(i), i))
    cb = rec.train([pipeline.DeviceBatch(train.batch(order), dev)])
    nb = test.batch(range(4))
    out = rec.detect(pipeline.DeviceBatch(nb, dev), keep_intermediates=True)
    f = out["features"]
    desc = f["desc"].cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)               # L2-normalised
    assert desc.min() >= 0
    R = f["lrf"].cpu().numpy().reshape(-1, 3, 3)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (len(R), 1, 1)), atol=5e-6)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)
    # training features are their own nearest codeword at distance 0 (self query)
    self_idx, self_d = capi.knn(ctx, rec.codebook, cfg.metric, T(cb["words"][:2000], dev), 1)
    assert np.array_equal(self_idx.cpu().numpy()[:, 0], np.arange(2000)) and float(self_d.max()) == 0.0
    assert (out["cls"][:, 0].cpu().numpy() == nb["labels"]).all()
    s = out["class_score"].cpu().numpy()
    assert (s >= 0).all() and (s.sum(1) <= 1 + 1e-5).all()
    # oracle on object 0 only (full size)
    o = 0
    ps, pe, ks, ke = nb["pt_off"][o], nb["pt_off"][o + 1], nb["kp_off"][o], nb["kp_off"][o + 1]
    xyz, nrm, kp = nb["xyz"][ps:pe], nb["normals"][ps:pe], nb["kp"][ks:ke]
    lrf = ora.shot_lrf([0, pe - ps], xyz[:, 0], xyz[:, 1], xyz[:, 2], [0, ke - ks], kp[:, 0], kp[:, 1], kp[:, 2], cfg.lrf_radius)
    wdesc, _ = ora.shot352([0, pe - ps], xyz[:, 0], xyz[:, 1], xyz[:, 2], nrm[:, 0], nrm[:, 1], nrm[:, 2], [0, ke - ks], kp[:, 0], kp[:, 1], kp[:, 2], lrf, cfg.radius)
    ok = ~np.isnan(wdesc).any(1)
    n0 = int(f["off"][1])
    assert n0 == ok.sum()
    assert np.abs(desc[:n0] - wdesc[ok]).max() < TOL
    widx, wd = ora.knn(cfg.metric, cb["words"], wdesc[ok], 1)
    gidx = out["idx"].cpu().numpy()[:n0]
    same = gidx[:, 0] == widx[:, 0]
    # the descriptors agree to ~1e-7, so a different winner is only possible on a near tie of the two best distances
    if not same.all():
        _, d2 = ora.knn(cfg.metric, cb["words"], wdesc[ok][~same], 2)
        assert ((d2[:, 1] - d2[:, 0]) < 1e-5).all()
    votes = ora.cast_votes(cb, cfg.weight_flags, lrf[ok], kp[ok, 0], kp[ok, 1], kp[ok, 2], gidx, out["dist"].cpu().numpy()[:n0])
    mx = ora.find_maxima([0, n0], votes, cfg.n_classes, cfg.bandwidth, max_maxima=cfg.max_maxima)
    n = int(mx["n"][0])
    assert n == int(out["n"][0])
    assert np.array_equal(mx["cls"][0, :n], out["cls"][0, :n].cpu().numpy())
    np.testing.assert_allclose(mx["weight"][0, :n], out["weight"][0, :n].cpu().numpy(), atol=TOL)
    np.testing.assert_allclose(mx["pos"][0, :n], out["pos"][0, :n].cpu().numpy(), atol=2e-3)


def test_shot_is_invariant_under_rigid_motion_full_size(pkg, gpu):
    """Size-independent property at BASELINE configs[1] sizes: SHOT-352 is built in the local reference frame, so moving the whole
    object (points, normals, keypoints) by a rigid motion rotates the frames and leaves the descriptors where they were. Rounding of
    the moved coordinates may carry a neighbour across a hard bin or shell boundary, so a small fraction of entries may move."""
    import torch
    ctx, dev = gpu
    capi, synthetic = pkg.capi, pkg.synthetic
    nb = synthetic.Dataset(4, 4, split=1).batch(range(4))
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    R = q.astype(np.float32); tvec = np.array([0.3, -1.1, 0.7], np.float32)

    def run(xyz, nrm, kp):
        x, y, z = (T(xyz[:, i].copy(), dev) for i in range(3)); nx, ny, nz = (T(nrm[:, i].copy(), dev) for i in range(3))
        kx, ky, kz = (T(kp[:, i].copy(), dev) for i in range(3))
        cloud = capi.Cloud(ctx, nb["pt_off"], x, y, z, nx, ny, nz, 0.15)
        lrf = capi.shot_lrf(ctx, cloud, nb["kp_off"], kx, ky, kz, 0.3)
        desc, cnt = capi.shot352(ctx, cloud, nb["kp_off"], kx, ky, kz, lrf, 0.4, want_counts=True)
        return lrf.cpu().numpy().reshape(-1, 3, 3), desc.cpu().numpy(), cnt.cpu().numpy()

    l0, d0, c0 = run(nb["xyz"], nb["normals"], nb["kp"])
    l1, d1, c1 = run(nb["xyz"] @ R.T + tvec, nb["normals"] @ R.T, nb["kp"] @ R.T + tvec)
    ok = ~(np.isnan(d0).any(1) | np.isnan(d1).any(1))
    assert ok.mean() > 0.99
    assert (np.abs(c0.astype(np.int64) - c1.astype(np.int64)) <= 2).mean() > 0.99            # a point on the sphere may fall either side
    # frames rotate with the object: axes' = axes R^T (row vectors); a near-degenerate covariance may flip an axis for a few keypoints
    fr = np.abs(l1[ok] - l0[ok] @ R.T).max((1, 2))
    good = fr < 1e-3
    assert good.mean() > 0.98, good.mean()
    dd = np.abs(d1[ok][good] - d0[ok][good])
    assert (dd < 2e-3).mean() > 0.999 and np.median(dd.max(1)) < 1e-4, ((dd < 2e-3).mean(), np.median(dd.max(1)))


def test_knn_bench_scale_modes_agree(pkg, gpu, ora, monkeypatch):
    """BASELINE configs[1] scale: 102 400 codewords x 352, 32 768 queries of clustered, descriptor-like unit
    vectors (plus exact duplicates of codewords and duplicated codewords); 512 of the queries also go through the oracle. The default path (f16 MFMA candidates -> exact re-rank ->
    proof -> exact scan of unproven slots; both MFMA shapes of the ring kernel, its tile variants, joined and separate codeword streams) and the exact-f32 MFMA
    candidate path are independent routes to the same contract, so
    indices and distances must agree bit for bit; duplicates must resolve to the lowest row at distance 0."""
    import torch
    _, dev = gpu
    g = torch.Generator(device="cpu").manual_seed(7)
    n_words, nq, dim = 102400, 32768, 352
    centres = torch.rand((400, dim), generator=g)
    words = centres[torch.randint(0, 400, (n_words,), generator=g)] + 0.05 * torch.rand((n_words, dim), generator=g)
    words = (words * (torch.rand((n_words, dim), generator=g) > 0.3)).float()              # sparse like SHOT histograms
    words /= words.norm(dim=1, keepdim=True)
    words[50000:50010] = words[10:20]                                                       # duplicated codewords: ties to the lowest row
    q = centres[torch.randint(0, 400, (nq,), generator=g)] + 0.05 * torch.rand((nq, dim), generator=g)
    q = (q * (torch.rand((nq, dim), generator=g) > 0.3)).float()
    q /= q.norm(dim=1, keepdim=True)
    q[:20] = words[:20]
    wn = words.numpy()
    off = np.arange(n_words + 1, dtype=np.uint32)
    res = {}
    modes = ("f16", "f16-nopca", "f16-pca192", "f16-pca128noqp2", "f16-m2off", "f16-m2x224", "f16-ring32", "f16-nojoin", "f16-half", "f16-qpanel", "f32")
    for mode in modes:
        monkeypatch.setenv("ISMHIP_KNN_MODE", mode.split("-")[0])
        if mode.endswith("nopca"):
            monkeypatch.setenv("ISMHIP_KNN_PCA_M", "0")                                       # stage 1 on all 352 dimensions
        elif mode.endswith("pca192"):
            monkeypatch.setenv("ISMHIP_KNN_PCA_M", "192")                                     # stage 1 forced onto 192 rotated coordinates
        elif mode.endswith("pca128noqp2"):
            monkeypatch.setenv("ISMHIP_KNN_PCA_M", "128")                                     # 128 coordinates, query panel NOT resident
        else:
            monkeypatch.delenv("ISMHIP_KNN_PCA_M", raising=False)
        if mode.endswith("m2off"):
            monkeypatch.setenv("ISMHIP_KNN_PCA_M2", "0")                                      # stage 2 on all 352 dimensions
        elif mode.endswith("m2x224"):
            monkeypatch.setenv("ISMHIP_KNN_PCA_M", "128"); monkeypatch.setenv("ISMHIP_KNN_PCA_M2", "224")   # stage 1 / 2 forced onto 128 / 224 coordinates
        else:
            monkeypatch.delenv("ISMHIP_KNN_PCA_M2", raising=False)
        monkeypatch.setenv("ISMHIP_KNN_RING32", "1" if mode.endswith("ring32") else "0")     # the 32x32x16 variant of the ring kernel
        monkeypatch.setenv("ISMHIP_KNN_JOIN", "0" if mode.endswith("nojoin") else "1")       # every workgroup sweeps its split from tile 0
        monkeypatch.setenv("ISMHIP_KNN_HALF", "1" if mode.endswith("half") else "0")         # 128 x 256 tiles, two workgroups per CU
        monkeypatch.setenv("ISMHIP_KNN_QPANEL", "1" if mode.endswith("qpanel") else "0")     # 256 x 128 tiles, query panel resident in LDS
        monkeypatch.setenv("ISMHIP_KNN_QPANEL2", "0" if mode.endswith("noqp2") else "1")     # 256 x 256 tiles with the panel resident (default at <= 160 coordinates)
        ctx = pkg.capi.Ctx(0)
        cb = pkg.capi.Codebook(ctx, wn, off, np.zeros((n_words, 3), np.float32), np.zeros(n_words, np.uint32), np.zeros(n_words, np.uint32), 1,
                               np.ones(1, np.float32))
        ctx.timers_enable(True)
        idx, dist = pkg.capi.knn(ctx, cb, 0, q.to(dev), 2)
        res[mode] = (idx.cpu().numpy(), dist.cpu().numpy(), _knn_flagged(ctx), cb.stage1_dims, cb.stage2_dims)
        cb.close()
    assert res["f16-nopca"][3] == 0 and res["f16-pca192"][3] == 192 and res["f16-pca128noqp2"][3] == 128
    assert res["f16-m2off"][4] == 0 and res["f16-m2x224"][3:5] == (128, 224) and res["f16-nopca"][4] == 0
    for m in modes[:-1]:
        assert np.array_equal(res[m][0], res["f32"][0]), m
        assert np.array_equal(res[m][1], res["f32"][1]), m
    # the oracle leg: every 64th query against all 102 400 words on the CPU
    sel = np.arange(0, nq, 64)
    widx, wdist = ora.knn(0, wn, q.numpy()[sel], 2)
    for m in ("f16", "f16-pca192", "f16-m2x224", "f32"):
        assert np.array_equal(res[m][0][sel], widx) and np.array_equal(res[m][1][sel], wdist), m
    i16, d16, flagged, _, _ = res["f16"]
    assert np.array_equal(i16[:20, 0], np.arange(20)) and (d16[:20, 0] == 0).all()
    assert np.array_equal(i16[10:20, 1], np.arange(50000, 50010)) and (d16[10:20, 1] == 0).all()    # the duplicate is the second neighbour
    assert (d16[:, 0] <= d16[:, 1]).all()
    assert flagged[0] < nq // 20, flagged                                                   # the proof carries the bulk, the scan the rest


# ------------------------------------------------------------------------------------------------ training on the device
@pytest.mark.parametrize("metric,k,clean_up", [(0, 1, True), (1, 1, True), (0, 2, False), (0, 3, False), (1, 2, False), (0, 1, False)])
def test_train_activate_matches_oracle(pkg, gpu, ora, metric, k, clean_up):
    """Codebook::activate on the device (ismhip_train_activate) against ismref_activate: kept words, vote CSR, vote geometry,
    computeWeights medians, statistical class weights (with the reference's class-keyed term3) and class sigma^2."""
    ctx, dev = gpu
    rng = np.random.default_rng(40 + 10 * metric + k)
    n, D = 1500, 48
    proto = rng.random((60, D)).astype(np.float32)
    feats = (proto[rng.integers(0, 60, n)] + 0.05 * rng.random((n, D))).astype(np.float32)      # clustered: shared nearest neighbours
    feats[1000:1030] = feats[5:35]                                # exact duplicates: with k = 1 both copies activate the LOWER row
    cls = np.sort(rng.integers(0, 4, n)).astype(np.uint32)
    model = np.zeros(n, np.uint32)
    for c in range(4):                                            # a few models per class, contiguous
        ids = np.nonzero(cls == c)[0]
        model[ids] = c * 10 + (np.arange(len(ids)) * 3 // max(1, len(ids)))
    A = rng.normal(size=(n, 3, 3)); Q, _ = np.linalg.qr(A); Q[np.linalg.det(Q) < 0, 2] *= -1
    lrf = Q.reshape(n, 9).astype(np.float32); kp = rng.normal(size=(n, 3)).astype(np.float32)
    centre = rng.normal(size=(40, 3)).astype(np.float32)[model]
    got = pkg.capi.train_activate(ctx, metric, T(feats, dev), T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), cls, model, centre,
                                  k=k, clean_up=clean_up, n_classes=4)
    want = ora.activate(metric, feats, lrf, kp, cls, model, centre, k=k, clean_up=clean_up, n_classes=4)
    for key in ("word_src", "vote_offsets", "vote_feature"):
        assert np.array_equal(got[key], want[key]), key
    np.testing.assert_allclose(got["vote_xyz"], want["vote_xyz"], atol=2e-6)
    np.testing.assert_allclose(got["vote_weight"], want["vote_weight"], atol=2e-6)
    np.testing.assert_allclose(got["vote_class_weight"], want["vote_class_weight"], rtol=1e-6, atol=1e-12)
    assert np.array_equal(got["class_sigma"], want["class_sigma"])            # same functor order, same sequential float sums
    if not clean_up:
        assert np.diff(want["vote_offsets"].astype(np.int64)).max() >= (3 if k > 1 else 2)   # multi-vote distributions were exercised
    else:
        assert len(want["word_src"]) < n - 30                                  # the duplicated rows were cleaned away


def _training_set(rng, n, D, n_proto=60, n_classes=4):
    proto = rng.random((n_proto, D)).astype(np.float32)
    feats = (proto[rng.integers(0, n_proto, n)] + 0.05 * rng.random((n, D))).astype(np.float32)
    cls = np.sort(rng.integers(0, n_classes, n)).astype(np.uint32)
    model = np.zeros(n, np.uint32)
    for c in range(n_classes):
        ids = np.nonzero(cls == c)[0]
        model[ids] = c * 10 + (np.arange(len(ids)) * 3 // max(1, len(ids)))
    A = rng.normal(size=(n, 3, 3)); Q, _ = np.linalg.qr(A); Q[np.linalg.det(Q) < 0, 2] *= -1
    lrf = Q.reshape(n, 9).astype(np.float32); kp = rng.normal(size=(n, 3)).astype(np.float32)
    centre = rng.normal(size=(10 * n_classes, 3)).astype(np.float32)[model]
    return feats, cls, model, lrf, kp, centre


def test_train_activate_hub_word_with_more_than_2048_votes(pkg, gpu, ora):
    """Clustering "None", k = 2, no clean-up: 2500 identical descriptors all activate the lowest two of their rows (ties go to the
    lowest row), so those two words carry > 2048 votes each -- the range k_tr_weights leaves to k_tr_weights_big, which used to be
    launched for clustered codebooks only (the vote weights of such a word stayed uninitialised)."""
    ctx, dev = gpu
    rng = np.random.default_rng(77)
    feats, cls, model, lrf, kp, centre = _training_set(rng, 4000, 32)
    feats[700:3200] = feats[700]                                      # 2500 copies of one descriptor
    got = pkg.capi.train_activate(ctx, 0, T(feats, dev), T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), cls, model, centre,
                                  k=2, clean_up=False, n_classes=4)
    want = ora.activate(0, feats, lrf, kp, cls, model, centre, k=2, clean_up=False, n_classes=4)
    assert np.diff(want["vote_offsets"].astype(np.int64)).max() > 2048
    for key in ("word_src", "vote_offsets", "vote_feature"):
        assert np.array_equal(got[key], want[key]), key
    assert np.isfinite(got["vote_weight"]).all()
    np.testing.assert_allclose(got["vote_weight"], want["vote_weight"], atol=2e-6)
    np.testing.assert_allclose(got["vote_class_weight"], want["vote_class_weight"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("init", ["FLANN_CENTERS_RANDOM", "FLANN_CENTERS_GONZALES", "FLANN_CENTERS_KMEANSPP"])
@pytest.mark.parametrize("metric,shape,k", [(0, (3000, 48), 40), (1, (1200, 33), 25), (0, (5000, 352), 300), (0, (400, 16), 400)])
def test_kmeans_matches_oracle(pkg, gpu, ora, init, metric, shape, k):
    """ClusteringKMeans::cluster (clustering_kmeans.h:53-131) on the device against the oracle's sequential restatement of the same
    build-defined k-means (seeded draws, integer potentials and means): centres, assignment, distances and the iteration count are
    bit-identical. FLANN itself is random and approximate here -- parity with the reference is unpinned (DESIGN)."""
    ctx, dev = gpu
    n, D = shape
    rng = np.random.default_rng(n + D + k)
    feats = _training_set(rng, n, D, n_proto=max(5, k // 2))[0]
    feats = (feats + 0.2 * rng.normal(size=feats.shape)).astype(np.float32)     # overlapping clusters: Lloyd needs several rounds
    if metric == 1:
        feats = np.abs(feats)
    feats[7] = feats[3]; feats[n - 1] = feats[3]                  # coinciding points: never two centres on them
    cen, assign, dist, it = pkg.capi.kmeans(ctx, metric, T(feats, dev), k, max_iterations=30, centers_init=init, seed=11)
    wcen, wassign, wdist, wit = ora.kmeans(metric, feats, k, max_iterations=30, centers_init=pkg.capi.CENTERS_INIT[init], seed=11)
    print(f"kmeans {init} metric {metric} {shape} k={k}: {len(wcen)} centres, {wit} iterations, sizes {np.bincount(wassign, minlength=len(wcen)).min()}..{np.bincount(wassign).max()}")
    assert it == wit and tuple(cen.shape) == wcen.shape
    assert np.array_equal(cen.cpu().numpy(), wcen)
    assert np.array_equal(assign.cpu().numpy(), wassign)
    assert np.array_equal(dist.cpu().numpy(), wdist)
    if k < n:                                                      # a k-means result: every feature sits with its nearest centre, no centre is idle
        assert np.bincount(wassign, minlength=len(wcen)).min() >= 1 or wit == 30
    else:
        assert len(wcen) <= n - 2                                  # the coinciding points cannot all become centres


def test_kmeans_on_separated_blobs_finds_them(pkg, gpu):
    """property: well separated blobs, k = number of blobs, k-means++ -> one centre per blob (its mean), every member assigned to it"""
    ctx, dev = gpu
    rng = np.random.default_rng(8)
    mu = rng.normal(size=(12, 64)).astype(np.float32) * 4
    lab = rng.integers(0, 12, 6000)
    x = (mu[lab] + 0.1 * rng.normal(size=(6000, 64))).astype(np.float32)
    cen, assign, dist, it = pkg.capi.kmeans(ctx, 0, T(x, dev), 12, seed=1)
    a = assign.cpu().numpy(); c = cen.cpu().numpy()
    assert len(c) == 12 and it <= 10
    for b in range(12):
        ids = np.nonzero(lab == b)[0]
        assert len(set(a[ids])) == 1
        np.testing.assert_allclose(c[a[ids[0]]], x[ids].mean(0), atol=1e-5)


@pytest.mark.parametrize("metric,k,clean_up", [(0, 2, False), (0, 3, False), (1, 2, False), (0, 1, True), (0, 6, False)])
def test_train_activate_with_cluster_centres_matches_oracle(pkg, gpu, ora, metric, k, clean_up):
    """Codebook::activate with a clustered codebook (implicit_shape_model.cpp:445-490): the codewords are k-means centres, every
    feature activates its K nearest of them (vote fan-out), distributions hold tens to hundreds of votes -- one of them more than
    2048, which takes the workgroup-per-vote median kernel -- and sigma^2 samples feature x centre distances."""
    ctx, dev = gpu
    rng = np.random.default_rng(70 + 10 * metric + k)
    n, D = 6000, 32
    feats, cls, model, lrf, kp, centre = _training_set(rng, n, D, n_proto=30)
    feats[2000:4600] = feats[2000] + 1e-3 * rng.random((2600, D)).astype(np.float32)    # 2600 features around one point: one crowded codeword
    cen, assign, _, _ = pkg.capi.kmeans(ctx, metric, T(feats, dev), 45, max_iterations=20, seed=5)
    cen_h = cen.cpu().numpy()
    got = pkg.capi.train_activate(ctx, metric, T(feats, dev), T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), cls, model, centre,
                                  k=k, clean_up=clean_up, n_classes=4, codewords=cen)
    want = ora.activate(metric, feats, lrf, kp, cls, model, centre, k=k, clean_up=clean_up, n_classes=4, codewords=cen_h)
    for key in ("word_src", "vote_offsets", "vote_feature"):
        assert np.array_equal(got[key], want[key]), key
    np.testing.assert_allclose(got["vote_xyz"], want["vote_xyz"], atol=2e-6)
    np.testing.assert_allclose(got["vote_weight"], want["vote_weight"], atol=2e-6)
    np.testing.assert_allclose(got["vote_class_weight"], want["vote_class_weight"], rtol=1e-6, atol=1e-12)
    assert np.array_equal(got["class_sigma"], want["class_sigma"])
    sizes = np.diff(want["vote_offsets"].astype(np.int64))
    if not clean_up:
        assert len(want["word_src"]) == 45 and sizes.sum() == n * k and sizes.max() > 2048
    else:
        assert len(want["word_src"]) < 45                          # K = 1 clean-up keeps only single-vote codewords (codebook.cpp:201-224)
    if k == 3:                                                     # fewer codewords than K: every feature activates all of them
        two = cen[:2].contiguous()
        got = pkg.capi.train_activate(ctx, metric, T(feats[:500], dev), T(lrf[:500], dev), T(kp[:500, 0], dev), T(kp[:500, 1], dev), T(kp[:500, 2], dev),
                                      cls[:500], model[:500], centre[:500], k=3, clean_up=False, n_classes=4, codewords=two)
        want = ora.activate(metric, feats[:500], lrf[:500], kp[:500], cls[:500], model[:500], centre[:500], k=3, clean_up=False, n_classes=4, codewords=cen_h[:2])
        assert np.array_equal(got["vote_offsets"], want["vote_offsets"]) and list(want["vote_offsets"]) == [0, 500, 1000]
        assert np.array_equal(got["vote_feature"], want["vote_feature"]) and np.array_equal(got["class_sigma"][:1], want["class_sigma"][:1])


# ------------------------------------------------------------------------------------------------ round-2 known-answer vectors
# the same closed-form vectors the oracle is held to (tests/kat_checks.py), here through the C ABI on the GPU
def test_kat_shot_off_centre_interpolation(pkg, gpu):
    ctx, dev = gpu

    def f(pts, nrm, radius):
        s = Scene(pkg, gpu, [(pts, nrm)], [np.zeros((1, 3), np.float32)], 0.4 * radius)
        d, cnt = pkg.capi.shot352(ctx, s.cloud, s.kp_off, *s.tk, T(np.eye(3, dtype=np.float32).reshape(1, 9), dev), radius, want_counts=True)
        assert int(cnt.cpu().numpy()[0]) == len(pts)
        return d.cpu().numpy()[0]
    kat_checks.shot_off_centre(f)


def test_kat_cshot_colour_channel(pkg, gpu):
    ctx, dev = gpu

    def f(pts, nrm, rgba, kp_rgba, radius):
        s = Scene(pkg, gpu, [(pts, nrm)], [np.zeros((1, 3), np.float32)], 0.4 * radius, rgba=[rgba], kp_rgba=[np.asarray([kp_rgba], np.uint32)])
        d = pkg.capi.cshot1344(ctx, s.cloud, s.kp_off, *s.tk, s.t_kp_rgba, T(np.eye(3, dtype=np.float32).reshape(1, 9), dev), radius)
        return d.cpu().numpy()[0]
    kat_checks.cshot_colour_pairs(f)


def test_kat_cast_votes_vector(pkg, gpu):
    ctx, dev = gpu

    def f(cb, flags, lrf, kp, idx, dist):
        dcb = pkg.capi.Codebook(ctx, cb["words"], cb["vote_offsets"], cb["vote_xyz"], cb["vote_class"], cb["vote_instance"], 2, cb["class_sigma"],
                                word_weight=cb["word_weight"], vote_weight=cb["vote_weight"], vote_class_weight=cb["vote_class_weight"],
                                vote_bbox_quat=cb["vote_bbox_quat"], vote_bbox_size=cb["vote_bbox_size"])
        got = pkg.capi.cast_votes(ctx, dcb, flags, T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), T(idx, dev), T(dist, dev), want_bbox=True)
        out = {k: v.cpu().numpy() for k, v in got.items() if v is not None}
        dcb.close()
        return out
    kat_checks.cast_votes_vector(f)


def test_kat_knn_rule_truth_table(pkg, gpu):
    ctx, dev = gpu

    def f(metric, words, wcls, q, thr):
        n = len(words)
        cb = pkg.capi.Codebook(ctx, words, np.arange(n + 1, dtype=np.uint32), np.zeros((n, 3), np.float32), wcls, np.zeros(n, np.uint32),
                               int(wcls.max()) + 1, np.ones(int(wcls.max()) + 1, np.float32))
        i, d = pkg.capi.knn_rule(ctx, cb, metric, T(q, dev), thr)
        return i.cpu().numpy(), d.cpu().numpy()
    kat_checks.knn_rule_table(f)


def test_kat_maxima_thresholds_and_bestk(pkg, gpu):
    ctx, dev = gpu

    def f(off, v, **kw):
        out = pkg.capi.find_maxima(ctx, off, {k: T(a, dev) for k, a in v.items()}, **kw)
        return {k: a.cpu().numpy() for k, a in out.items()}
    kat_checks.maxima_thresholds(f)


def test_kat_hough3d_three_bins(pkg, gpu):
    ctx, dev = gpu

    def f(off, v, **kw):
        out = pkg.capi.hough3d_maxima(ctx, off, {k: T(a, dev) for k, a in v.items()}, **kw)
        return {k: a.cpu().numpy() for k, a in out.items()}
    kat_checks.hough3d_three_bins(f)


def test_kat_activate_weights(pkg, gpu):
    ctx, dev = gpu

    def f(metric, feats, lrf, kp, cls, model, centre, k, clean_up, n_classes):
        return pkg.capi.train_activate(ctx, metric, T(feats, dev), T(lrf, dev), T(kp[:, 0], dev), T(kp[:, 1], dev), T(kp[:, 2], dev), cls, model, centre,
                                       k=k, clean_up=clean_up, n_classes=n_classes)
    kat_checks.activate_weights(f)


def test_kat_meanshift_step_and_double_reweight(pkg, gpu):
    ctx, dev = gpu

    def f(off, v, **kw):
        out = pkg.capi.find_maxima(ctx, off, {k: T(a, dev) for k, a in v.items()}, **kw)
        return {k: a.cpu().numpy() for k, a in out.items()}
    kat_checks.meanshift_step_and_double_reweight(f)


def test_kat_fpfh_three_points(pkg, gpu):
    ctx, dev = gpu

    def f(pts, nrm, kp, radius):
        s = Scene(pkg, gpu, [(pts, nrm)], [kp.reshape(1, 3)], 0.4 * radius)
        d, cnt = pkg.capi.fpfh33(ctx, s.cloud, s.kp_off, *s.tk, radius, want_counts=True)
        assert int(cnt.cpu().numpy()[0]) == len(pts)
        return d.cpu().numpy()[0]
    kat_checks.fpfh_three_points(f)


def test_kat_lrf_majority_sign(pkg, gpu):
    ctx, dev = gpu

    def f(pts, kp, radius):
        s = Scene(pkg, gpu, [(pts, np.zeros_like(pts))], [kp.reshape(1, 3)], 0.4 * radius)
        return pkg.capi.shot_lrf(ctx, s.cloud, s.kp_off, *s.tk, radius).cpu().numpy()[0]
    kat_checks.lrf_majority_sign(f)


def test_kat_pca_normals_slab(pkg, gpu):
    import torch
    ctx, dev = gpu

    def f(pts, radius, orientation):
        s = Scene(pkg, gpu, [(pts, np.zeros_like(pts))], [np.zeros((0, 3), np.float32)], 0.5 * radius)
        n = [torch.empty(len(pts), dtype=torch.float32, device=dev) for _ in range(3)]
        pkg.capi.estimate_normals_pca(ctx, s.cloud, radius, orientation, *n)
        return np.stack([t.cpu().numpy() for t in n], 1)
    kat_checks.pca_normals_slab(f)


def test_kat_kmeans_two_blobs(pkg, gpu):
    ctx, dev = gpu

    def f(pts, n_clusters, init, seed):
        c, asg, d, _ = pkg.capi.kmeans(ctx, 0, T(pts, dev), n_clusters, centers_init=init, seed=seed)
        return c.cpu().numpy(), asg.cpu().numpy(), d.cpu().numpy()
    kat_checks.kmeans_two_blobs(f)


@pytest.mark.parametrize("interp,bin_size,rel", [(True, 0.4, 0.6), (False, 0.4, 0.6), (True, 0.1, 0.3), (True, 0.25, 0.9)])
def test_hough3d_matches_oracle(pkg, gpu, ora, interp, bin_size, rel):
    """VotingHough3D on the device against the oracle on ragged vote sets (empty object, slots without a vote, clutter, votes
    outside the space). bin 0.1 spreads a class over ~60 bins per axis: the accumulator takes several LDS tiles."""
    ctx, dev = gpu
    rng = np.random.default_rng(int(1000 * bin_size) + int(interp))
    off, v = _vote_scene(rng, 10, 4)
    v["pos"][::53] += 20.0                                        # outside [-5, 5]^3: dropped by the accumulator
    tv = {k2: T(a, dev) for k2, a in v.items()}
    kw = dict(n_classes=4, bin_size=bin_size, use_interpolation=interp, rel_threshold=rel, max_maxima=24, min_votes_threshold=2)
    got = pkg.capi.hough3d_maxima(ctx, off, tv, **kw)
    want = ora.hough3d_maxima(off, v, **kw)
    assert np.array_equal(got["n"].cpu().numpy(), want["n"])
    assert np.array_equal(got["cls"].cpu().numpy(), want["cls"])
    assert np.array_equal(got["inst"].cpu().numpy(), want["inst"])
    assert np.array_equal(got["n_votes"].cpu().numpy(), want["n_votes"])
    np.testing.assert_allclose(got["weight"].cpu().numpy(), want["weight"], atol=TOL)
    np.testing.assert_allclose(got["inst_weight"].cpu().numpy(), want["inst_weight"], atol=TOL)
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=1e-4)
    np.testing.assert_allclose(got["bbox_size"].cpu().numpy(), want["bbox_size"], atol=1e-3)
    np.testing.assert_allclose(got["class_score"].cpu().numpy(), want["class_score"], atol=TOL)
    assert want["n"][1] == 0 and want["n"].max() >= 2


def test_all_gather_records_over_rccl_world1(pkg, gpu):
    """the path's one collective through the nccl (= RCCL) backend on this box's GPU: world size 1, device tensors"""
    import torch
    import torch.distributed as dist
    ctx, dev = gpu
    sh = pkg.shard
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=dev)
    try:
        g = torch.Generator().manual_seed(5)
        scores = torch.rand((9, 4), generator=g).to(dev)
        scores[3] = 0
        rec = sh.pack_records(torch.arange(9, device=dev), scores, 12)
        out = sh.all_gather_records(rec, 1)
        assert out.is_cuda and out.shape == (12, 6) and dist.get_backend() == "nccl"
        oi, best, sc = sh.unpack_records(out)
        exp = scores.argmax(1); exp[3] = -1
        assert oi.tolist() == list(range(9)) and best.tolist() == exp.tolist() and torch.equal(sc, scores)
    finally:
        dist.destroy_process_group()


def test_errors_are_loud(pkg, gpu):
    import ctypes as C
    ctx, dev = gpu
    L = pkg.capi.lib()
    h = C.c_void_p()
    assert L.ismhip_cloud_create(ctx._h, 0, None, None, None, None, None, None, None, None, C.c_float(0.1), C.byref(h)) == -1
    assert b"cloud_create" in L.ismhip_last_error(ctx._h)
    with pytest.raises(pkg.capi.IsmHipError):
        pkg.capi.Ctx(99)
    words = np.eye(8, dtype=np.float32)
    _, cb = _cb(pkg, gpu, words)
    with pytest.raises(pkg.capi.IsmHipError):
        pkg.capi.knn(ctx, cb, 0, T(words, dev), 17)         # k > 16 is not built -> ISMHIP_ERR_UNSUPPORTED, not a silent fallback
    x = T(np.arange(8, dtype=np.float32), dev)
    with pytest.raises(pkg.capi.IsmHipError, match="bad argument"):
        pkg.capi.voxel_keypoints(ctx, [0, 8], x, x, x, 0.0)                     # leaf must be positive
    with pytest.raises(pkg.capi.IsmHipError, match="start at 0"):
        pkg.capi.voxel_keypoints(ctx, [1, 8], x, x, x, 0.5)
    with pytest.raises(pkg.capi.IsmHipError, match="leaf too small"):
        pkg.capi.voxel_keypoints(ctx, [0, 8], x * 1e9, x, x, 1e-3)             # PCL: "leaf size is too small for the input dataset"
    ko, kx, _, _, _ = pkg.capi.voxel_keypoints(ctx, [0, 0, 8], x, x, x, 100.0)  # empty object first, one voxel for the rest
    assert ko.tolist() == [0, 0, 1] and abs(float(kx[0]) - 3.5) < 1e-6
