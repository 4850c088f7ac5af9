"""Round-2 known-answer vectors (tests/golden/kat.json: shot_off_centre, cshot_colour_pairs, cast_votes_vector, knn_rule_table,
maxima_thresholds), written once against an abstract back end so that the SAME checks run on the CPU oracle
(tests/test_oracle_kat.py) and on the HIP path through the C ABI (tests/test_gpu_parity.py). The expected values come from
closed-form numpy in tests/golden/make_golden.py, not from the oracle and not from the HIP library."""
import json
import os

import numpy as np

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def shot_off_centre(shot352):
    """shot352(points[n,3], normals[n,3], radius) -> descriptor of a keypoint at the origin with the identity frame"""
    k = KAT["shot_off_centre"]
    for c in k["cases"]:
        pts = np.tile(np.asarray(c["point"], np.float32), (c["copies"], 1))
        nrm = np.tile(np.asarray(c["normal"], np.float32), (c["copies"], 1))
        d = shot352(pts, nrm, c["radius"])
        exp = np.asarray(c["expected"], np.float32)
        np.testing.assert_allclose(d, exp, atol=k["tol"])
        for name, (b, w) in c["parts"].items():          # every interpolation deposit sits in ITS bin with ITS share
            if name != "colour":
                assert abs(d[b] - w / np.sqrt(sum(v[1] ** 2 for kk, v in c["parts"].items() if kk != "colour"))) < 1e-5, name


def cshot_colour_pairs(cshot1344):
    """cshot1344(points, normals, rgba[n], kp_rgba, radius) -> 1344 values"""
    k = KAT["cshot_colour_pairs"]
    for c in k["cases"]:
        n = c["copies"]
        pts = np.tile(np.asarray(c["point"], np.float32), (n, 1)); nrm = np.tile(np.asarray(c["normal"], np.float32), (n, 1))
        d = cshot1344(pts, nrm, np.full(n, c["rgba"], np.uint32), c["kp_rgba"], c["radius"])
        np.testing.assert_allclose(d, np.asarray(c["expected"], np.float32), atol=k["tol"])
        assert abs(np.linalg.norm(d) - 1) < 1e-5
        col = d[352:].reshape(32, 31)
        assert col[:, c["colour_step"]].sum() > 0.5 * col.sum()       # the hard colour step holds most of the colour mass


def cast_votes_vector(cast_votes):
    """cast_votes(cb dict, flags, lrf[nq,9], kp[nq,3], idx[nq,1], dist[nq,1]) -> dict of slot arrays (2 slots per activation)"""
    k = KAT["cast_votes_vector"]
    cb = {kk: np.asarray(v, np.float32) for kk, v in k["codebook"].items()}
    for kk in ("vote_offsets", "vote_class", "vote_instance"):
        cb[kk] = np.asarray(k["codebook"][kk], np.uint32)
    acts = k["activations"]
    lrf = np.asarray([k["frames"][a["frame"]] for a in acts], np.float32)
    kp = np.asarray([a["kp"] for a in acts], np.float32)
    idx = np.asarray([[a["word"]] for a in acts], np.int32)
    dist = np.asarray([[a["dist"]] for a in acts], np.float32)
    for flags, rows in k["expected"].items():
        got = cast_votes(cb, int(flags), lrf, kp, idx, dist)
        assert len(got["cls"]) == len(rows) == 2 * len(acts)
        for s, r in enumerate(rows):
            assert got["cls"][s] == r["cls"], (flags, s)
            if r["cls"] < 0:
                continue
            assert got["inst"][s] == r["inst"] and got["codeword"][s] == r["codeword"], (flags, s)
            assert abs(got["weight"][s] - r["weight"]) <= k["tol"] * max(1.0, r["weight"]), (flags, s, got["weight"][s], r["weight"])
            np.testing.assert_allclose(got["pos"][s], r["pos"], atol=5e-6)
            np.testing.assert_allclose(got["bbox_quat"][s], r["bbox_quat"], atol=5e-6)
            np.testing.assert_allclose(got["bbox_size"][s], r["bbox_size"], atol=0)


def knn_rule_table(knn_rule):
    """knn_rule(metric, words, word_class, q, thr) -> (idx[nq], dist[nq])"""
    k = KAT["knn_rule_table"]
    idx, dist = knn_rule(0, np.asarray(k["words"], np.float32), np.asarray(k["word_class"], np.uint32), np.asarray(k["q"], np.float32), k["threshold"])
    idx = np.asarray(idx).reshape(-1); dist = np.asarray(dist).reshape(-1)
    assert idx.tolist() == k["expected_idx"]
    for g, e in zip(dist, k["expected_dist"]):
        assert (np.isnan(g) if e is None else g == np.float32(e)), (g, e)


def maxima_thresholds(find_maxima):
    """find_maxima(slot_offsets, votes dict, **kw) -> dict as oracle_py.find_maxima"""
    k = KAT["maxima_thresholds"]
    v = dict(pos=np.asarray(k["pos"], np.float32), weight=np.asarray(k["w"], np.float32), cls=np.asarray(k["cls"], np.int32),
             inst=np.asarray(k["inst"], np.int32))
    for c in k["cases"]:
        out = find_maxima([0, len(v["weight"])], v, n_classes=k["n_classes"], bandwidth=k["bandwidth"], max_maxima=8,
                           min_threshold=c["min_threshold"], best_k=c["best_k"])
        n = int(np.asarray(out["n"])[0])
        assert n == c["n"], (c, n)
        np.testing.assert_allclose(np.asarray(out["weight"])[0, :n], k["weights"][:n], atol=k["tol"])
        assert np.asarray(out["cls"])[0, :n].tolist() == k["classes"][:n]
        assert np.asarray(out["inst"])[0, :n].tolist() == k["instances"][:n]
        assert np.asarray(out["n_votes"])[0, :n].tolist() == k["n_votes"][:n]
        np.testing.assert_allclose(np.asarray(out["pos"])[0, :n], np.asarray(k["pos"], np.float32)[[0, 4, 7]][:n], atol=1e-5)
        score = np.zeros(k["n_classes"], np.float32)
        score[k["classes"][:n]] = k["weights"][:n]
        np.testing.assert_allclose(np.asarray(out["class_score"])[0], score, atol=k["tol"])


def hough3d_three_bins(hough):
    """hough(slot_offsets, votes dict, **kw) -> dict as oracle_py.hough3d_maxima"""
    k = KAT["hough3d_three_bins"]
    v = dict(pos=np.asarray(k["pos"], np.float32), weight=np.asarray(k["w"], np.float32), cls=np.asarray(k["cls"], np.int32),
             inst=np.asarray(k["inst"], np.int32))
    for rel, interp in ((k["rel_threshold"], True), (0.3, True), (0.4, False)):      # no interpolation: H = 1, 2, 1 -> threshold 0.8
        out = hough([0, 3], v, n_classes=k["n_classes"], bin_size=k["bin"], min_coord=k["min_coord"], max_coord=k["max_coord"],
                    rel_threshold=rel, use_interpolation=interp, max_maxima=8)
        n = int(np.asarray(out["n"])[0])
        assert n == k["expected_n"], (rel, interp, n)
        np.testing.assert_allclose(np.asarray(out["pos"])[0, :n], np.asarray(k["expected_pos"], np.float32), atol=k["tol"])
        np.testing.assert_allclose(np.asarray(out["weight"])[0, :n], k["expected_weight"], atol=k["tol"])
        assert np.asarray(out["inst"])[0, :n].tolist() == k["expected_inst"] and np.asarray(out["n_votes"])[0, :n].tolist() == k["expected_n_votes"]
        np.testing.assert_allclose(np.asarray(out["class_score"])[0], [0.5, 0.0], atol=k["tol"])


# ---- round-3 vectors ---------------------------------------------------------------------------------------------------------
def activate_weights(activate):
    """activate(metric, feats, lrf, kp, cls, model, centre, k, clean_up, n_classes) -> dict as oracle_py.activate"""
    k = KAT["activate_weights"]
    n = len(k["cls"])
    out = activate(0, np.asarray(k["feats"], np.float32), np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)), np.asarray(k["kp"], np.float32),
                   np.asarray(k["cls"], np.uint32), np.asarray(k["model"], np.uint32), np.asarray(k["centre"], np.float32), k["k"], False, k["n_classes"])
    assert np.asarray(out["word_src"]).tolist() == k["word_src"]
    assert np.asarray(out["vote_offsets"]).tolist() == k["vote_offsets"]
    assert np.asarray(out["vote_feature"]).tolist() == k["vote_feature"]
    np.testing.assert_allclose(np.asarray(out["vote_xyz"]).reshape(-1, 3), k["vote_xyz"], atol=k["tol"])
    np.testing.assert_allclose(out["vote_weight"], k["vote_weight"], atol=k["tol"])             # medians: odd list, even list, single
    np.testing.assert_allclose(out["vote_class_weight"], k["vote_class_weight"], rtol=1e-6)    # term1 * term2 * term3, class-keyed term3


def meanshift_step_and_double_reweight(find_maxima):
    k = KAT["meanshift_step_and_double_reweight"]
    v = dict(pos=np.asarray(k["pos"], np.float32), weight=np.asarray(k["w"], np.float32), cls=np.asarray(k["cls"], np.int32),
             inst=np.asarray(k["inst"], np.int32))
    out = find_maxima([0, len(v["weight"])], v, n_classes=k["n_classes"], bandwidth=k["bandwidth"], max_iter=k["max_iter"], suppression=k["suppression"],
                      max_maxima=8)
    n = int(np.asarray(out["n"])[0])
    assert n == 4
    assert np.asarray(out["cls"])[0, :n].tolist() == k["expected_cls"]
    assert np.asarray(out["n_votes"])[0, :n].tolist() == k["expected_n_votes"]
    np.testing.assert_allclose(np.asarray(out["pos"])[0, :n], k["expected_pos"], atol=5e-6)       # ONE mean-shift step from each seed
    np.testing.assert_allclose(np.asarray(out["weight"])[0, :n], k["expected_weight"], atol=k["tol"])
    assert abs(float(np.asarray(out["weight"])[0, 1]) - k["single_reweight_would_be"]) > 1e-3    # the doubly reweighted vote really is in the second maximum


def fpfh_three_points(fpfh33):
    """fpfh33(points, normals, keypoint, radius) -> 33 values"""
    k = KAT["fpfh_three_points"]
    d = fpfh33(np.asarray(k["points"], np.float32), np.asarray(k["normals"], np.float32), np.asarray(k["keypoint"], np.float32), k["radius"])
    np.testing.assert_allclose(d, k["expected"], atol=k["tol"])
    np.testing.assert_allclose(np.asarray(d).reshape(3, 11).sum(1), 100.0, atol=1e-3)


def lrf_majority_sign(shot_lrf):
    """shot_lrf(points, keypoint, radius) -> 9 values"""
    k = KAT["lrf_majority_sign"]
    f = np.asarray(shot_lrf(np.asarray(k["points"], np.float32), np.asarray(k["keypoint"], np.float32), k["radius"]))
    np.testing.assert_allclose(f, k["expected"], atol=k["tol"])


def pca_normals_slab(pca_normals):
    """pca_normals(points, radius, orientation) -> [n, 3] (orientation 0: towards the origin, 1: away from the centroid)"""
    k = KAT["pca_normals_slab"]
    pts = np.asarray(k["points"], np.float32)
    want = np.asarray([[np.nan if v is None else v for v in r] for r in k["towards_origin"]], np.float64)
    for orientation, sign in ((0, 1.0), (1, -1.0)):
        got = np.asarray(pca_normals(pts, k["radius"], orientation), np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        fin = ~np.isnan(want[:, 0])
        np.testing.assert_allclose(got[fin], sign * want[fin], atol=k["tol"])


def kmeans_two_blobs(kmeans):
    """kmeans(points [n, dim], n_clusters, centers_init (0 random, 1 Gonzales, 2 k-means++), seed) -> (centres [m, dim], assign [n], sqdist [n])"""
    k = KAT["kmeans_two_blobs"]
    pts = np.asarray(k["points"], np.float32)
    optimum = 0
    for init in (0, 1, 2):
        for seed in range(6):
            c, asg, d = kmeans(pts, 2, init, seed)
            c, asg, d = np.asarray(c, np.float64), np.asarray(asg), np.asarray(d, np.float64)
            assert c.shape == (2, 2)
            all_d = ((pts[:, None, :].astype(np.float64) - c[None]) ** 2).sum(2)
            np.testing.assert_allclose(d, all_d[np.arange(8), asg], atol=1e-5)                    # the distance reported is the one to the assigned centre
            assert (all_d[np.arange(8), asg] <= all_d.min(1) + 1e-5).all()                        # ... which is a nearest one
            for t in range(2):                                                                     # every centre is the mean of its members
                assert (asg == t).any()
                np.testing.assert_allclose(c[t], pts[asg == t].mean(0), atol=1e-5)
            if sorted(map(tuple, np.round(c, 5).tolist())) == sorted(map(tuple, np.asarray(k["centres"], float).tolist())):
                optimum += 1
                assert len(set(asg[:4])) == 1 and len(set(asg[4:])) == 1 and asg[0] != asg[4]
                np.testing.assert_allclose(d, k["sqdist"], atol=1e-5)
    # Gonzales picks the farthest point as the second centre: always the other group -> always the optimum; the drawn seedings mostly
    assert optimum >= 12, optimum
