"""GPU parity on reduced-size versions of every BASELINE.json configuration (configs[0], [2], [3], [4]; configs[1] is the
bench workload and is covered at full size in test_gpu_parity.py): the HIP pipeline and the oracle pipeline run on the same
synthetic objects and must agree on the kept features, kNN matches, vote classes, predicted labels and class scores."""
import numpy as np
import pytest

import oracle_pipeline

pytestmark = pytest.mark.gpu


def _run(pkg, gpu, ora, cfg, train, test, n_train, n_test, desc_tol=1e-4):
    ctx, dev = gpu
    pipeline = pkg.pipeline
    rec = pipeline.Recognizer(ctx, cfg)
    order = sorted(range(n_train), key=lambda i: (train.label(i), i))
    cb = rec.train([pipeline.DeviceBatch(train.batch(order), dev)])
    nb = test.batch(range(n_test))
    got = rec.detect(pipeline.DeviceBatch(nb, dev), keep_intermediates=True)
    want = oracle_pipeline.detect(ora, cfg, cb, nb, cfg.n_classes)
    f = got["features"]
    assert np.array_equal(f["off"], want["off"]), "kept feature counts differ"
    assert np.array_equal(f["src"].cpu().numpy(), np.nonzero(want["keep"])[0])
    scale = 100.0 if cfg.feature == "FPFH" else 1.0
    assert np.abs(f["desc"].cpu().numpy() - want["desc"]).max() <= desc_tol * scale
    gi, wi = got["idx"].cpu().numpy(), want["idx"]
    same = (gi == wi).all(1)
    assert same.mean() > 0.995, same.mean()              # a different winner needs a near tie of the two best distances
    if not same.all():
        # recompute both candidates' distances for the GPU's own query rows: the GPU winner must be the exact minimiser of ITS
        # descriptors (which differ from the oracle's by ~1e-7), i.e. only a near tie may separate the two answers
        gq = f["desc"].cpu().numpy()[~same]
        for row, (a, b) in zip(gq, zip(gi[~same][:, 0], wi[~same][:, 0])):
            da, db = ora.distance(cfg.metric, row, cb["words"][a]), ora.distance(cfg.metric, row, cb["words"][b])
            assert da <= db, (da, db)
            assert db - da <= 2e-4 * max(1.0, db), (da, db)
    assert (got["cls"][:, 0].cpu().numpy() == want["cls"][:, 0]).all()
    np.testing.assert_allclose(got["class_score"].cpu().numpy(), want["class_score"], atol=2e-3)
    return got, want, nb


def test_cfg0_quickstart_chi2_model_units(pkg, gpu, ora):
    """configs[0]: the quick-start value set (config/qs_chi2_shot.ism): SHOT, Radius 60, LRF 50, LeafSize 50, Bandwidth 50,
    ChiSquared, K = 1, objects of extent ~350 model units, 5 classes x 1 object."""
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", radius=60.0, lrf_radius=50.0, distance="ChiSquared", k=1, bandwidth=50.0, n_classes=5)
    syn = pkg.synthetic
    train = syn.Dataset(5, 5, split=0, n_points=8192, leaf=50.0, scale=350.0)
    test = syn.Dataset(5, 5, split=1, n_points=8192, leaf=50.0, scale=350.0)
    got, want, nb = _run(pkg, gpu, ora, cfg, train, test, 5, 5)
    assert (got["cls"][:, 0].cpu().numpy() == nb["labels"]).all()


def test_cfg2_random_codebook_2048_keypoints(pkg, gpu, ora):
    """configs[2]: 2048 keypoints/object and a fixed-size codebook = seeded random subset of the training features
    (the reference's UseRandomCodebook mechanism, codebook/codebook.cpp:821-829)."""
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=4, use_random_codebook=True, random_codebook_size=3000)
    syn = pkg.synthetic
    train = syn.Dataset(4, 8, split=0, n_points=16384, n_keypoints=2048)
    test = syn.Dataset(4, 3, split=1, n_points=16384, n_keypoints=2048)
    got, want, nb = _run(pkg, gpu, ora, cfg, train, test, 8, 3)
    assert got["features"]["off"][-1] > 3 * 1900


def test_cfg3_cshot_chi2_partial_views(pkg, gpu, ora):
    """configs[3]: Kinect-like coloured partial views, CSHOT-1344, Radius/LRF 0.05, LeafSize 0.02, Bandwidth 0.045, ChiSquared
    (value set of config/kinect_cshot.ism)."""
    cfg = pkg.pipeline.IsmConfig(feature="CSHOT", radius=0.05, lrf_radius=0.05, distance="ChiSquared", bandwidth=0.045, n_classes=3)
    syn = pkg.synthetic
    kw = dict(n_points=8192, leaf=0.02, scale=0.15, with_color=True, partial_view=True)
    train = syn.Dataset(3, 6, split=0, **kw)
    test = syn.Dataset(3, 3, split=1, **kw)
    _run(pkg, gpu, ora, cfg, train, test, 6, 3)


def test_cfg4_dual_fpfh_shot(pkg, gpu, ora):
    """configs[4]: FPFH-33 and SHOT-352 as two models whose normalised class scores are summed by the harness."""
    syn = pkg.synthetic
    train = syn.Dataset(3, 6, split=0, n_points=8192, n_keypoints=512)
    test = syn.Dataset(3, 3, split=1, n_points=8192, n_keypoints=512)
    scores = []
    for feat in ("FPFH", "SHOT"):
        cfg = pkg.pipeline.IsmConfig(feature=feat, radius=0.3 if feat == "FPFH" else 0.4, n_classes=3)
        got, want, nb = _run(pkg, gpu, ora, cfg, train, test, 6, 3)
        scores.append((got["class_score"].cpu().numpy(), want["class_score"]))
    fused_gpu = scores[0][0] + scores[1][0]
    fused_ora = scores[0][1] + scores[1][1]
    assert (fused_gpu.argmax(1) == fused_ora.argmax(1)).all()
    assert (fused_gpu.argmax(1) == nb["labels"]).all()


def test_cfg1_with_hough3d_voting(pkg, gpu, ora):
    """the ModelNet-like value set with Voting.Type Hough3D (BinSize 0.4, RelThreshold 0.5): the discrete accumulator end to end"""
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=4, voting="Hough3D", hough_bin_size=0.4, hough_rel_threshold=0.5,
                                 hough_min_coord=(-3.0, -3.0, -3.0), hough_max_coord=(3.0, 3.0, 3.0))
    syn = pkg.synthetic
    train = syn.Dataset(4, 8, split=0, n_points=8192, n_keypoints=512)
    test = syn.Dataset(4, 4, split=1, n_points=8192, n_keypoints=512)
    got, want, nb = _run(pkg, gpu, ora, cfg, train, test, 8, 4)
    assert (got["cls"][:, 0].cpu().numpy() == nb["labels"]).all()
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=1e-4)


@pytest.mark.parametrize("kind,n_cols", [("front", 176), ("top", 176), ("dense_x_or_z", 264), ("dense_x_and_z", 88)])
def test_cfg1_with_partial_shot(pkg, gpu, ora, kind, n_cols):
    """Codebook.UsePartialShot (codebook.cpp:416-475, 862-930, mask :952-1036): codewords and features are matched on the histograms
    of a subset of the 32 SHOT signatures; the GPU pipeline against the oracle pipeline with the same column mask."""
    assert len(pkg.capi.partial_shot_columns(kind)) == n_cols
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=3, use_partial_shot=True, partial_shot_type=kind)
    syn = pkg.synthetic
    train = syn.Dataset(3, 6, split=0, n_points=8192, n_keypoints=384)
    test = syn.Dataset(3, 3, split=1, n_points=8192, n_keypoints=384)
    got, want, nb = _run(pkg, gpu, ora, cfg, train, test, 6, 3)
    assert (got["cls"][:, 0].cpu().numpy() == nb["labels"]).all()


def test_cfg1_knn_rule_activation_end_to_end(pkg, gpu, ora):
    """ActivationStrategy KNNRule end to end: trained with plain 1-NN WITHOUT the K = 1 clean-up (multi-vote codewords keep their
    computeWeights / term1*term2*term3 weights), detected with the class-consistency rule."""
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=3, activation="KNNRule", distance_ratio_threshold=0.9, use_class_weight=True, use_vote_weight=True)
    syn = pkg.synthetic
    train = syn.Dataset(3, 6, split=0, n_points=8192, n_keypoints=384)
    test = syn.Dataset(3, 3, split=1, n_points=8192, n_keypoints=384)
    ctx, dev = gpu
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    order = sorted(range(6), key=lambda i: (train.label(i), i))
    cb = rec.train([pkg.pipeline.DeviceBatch(train.batch(order), dev)])
    nb = test.batch(range(3))
    got = rec.detect(pkg.pipeline.DeviceBatch(nb, dev), keep_intermediates=True)
    f = got["features"]
    q = f["desc"].cpu().numpy()
    wi, wd = ora.knn_rule(cfg.metric, cb["words"], cb["word_class"], q, cfg.distance_ratio_threshold)
    assert np.array_equal(got["idx"].cpu().numpy().reshape(-1), wi.reshape(-1))
    votes = ora.cast_votes(cb, cfg.weight_flags, f["lrf"].cpu().numpy(), f["kx"].cpu().numpy(), f["ky"].cpu().numpy(), f["kz"].cpu().numpy(),
                           wi.reshape(-1, 1), wd.reshape(-1, 1))
    assert np.array_equal(got["votes"]["cls"].cpu().numpy(), votes["cls"])
    np.testing.assert_allclose(got["votes"]["weight"].cpu().numpy(), votes["weight"], rtol=1e-5, atol=1e-9)
    assert (got["cls"][:, 0].cpu().numpy() == nb["labels"]).all()


def test_kmeans_codebook_with_vote_fan_out_end_to_end(pkg, gpu, ora):
    """Clustering KMeansCount + KNN activation with K = 2 (the reference's clustered setting, implicit_shape_model.cpp:445-490): the
    training features are clustered on the device, every feature votes through its 2 nearest centres, detection activates 2 centres
    per feature and casts all their votes. Checked stage by stage: centres and distributions against the oracle's k-means /
    activate on the same features, detection against the oracle's kNN + castVotes on the trained codebook; the objects are classified."""
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=3, k=2, clustering="KMeansCount", cluster_count=300, kmeans_iterations=25, kmeans_seed=4,
                                 use_vote_weight=True, use_matching_weight=True, max_maxima=8)   # (the class weights' class-keyed term3 misleads 150-word codebooks)
    syn = pkg.synthetic
    train = syn.Dataset(3, 6, split=0, n_points=8192, n_keypoints=384)
    test = syn.Dataset(3, 3, split=1, n_points=8192, n_keypoints=384)
    ctx, dev = gpu
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    order = sorted(range(6), key=lambda i: (train.label(i), i))
    tb = pkg.pipeline.DeviceBatch(train.batch(order), dev)
    cb = rec.train([tb])
    assert len(cb["words"]) == 300 and rec.kmeans_iterations >= 2
    sizes = np.diff(cb["vote_offsets"].astype(np.int64))
    assert sizes.sum() == 2 * len(rec.cluster_indices) and sizes.max() > 3          # fan-out: every feature left 2 votes
    # the training features again, through the oracle's clustering and activation
    f = rec.compute_features(tb)
    desc = f["desc"].cpu().numpy()
    wcen, wassign, _, wit = ora.kmeans(cfg.metric, desc, 300, max_iterations=25, centers_init=2, seed=4)
    assert np.array_equal(wcen, cb["words"]) and np.array_equal(wassign, rec.cluster_indices.cpu().numpy()) and wit == rec.kmeans_iterations
    # detection on the clustered codebook
    nb = test.batch(range(3))
    got = rec.detect(pkg.pipeline.DeviceBatch(nb, dev), keep_intermediates=True)
    fd = got["features"]
    wi, wd = ora.knn(cfg.metric, cb["words"], fd["desc"].cpu().numpy(), 2)
    assert np.array_equal(got["idx"].cpu().numpy().reshape(-1, 2), wi)
    votes = ora.cast_votes(cb, cfg.weight_flags, fd["lrf"].cpu().numpy(), fd["kx"].cpu().numpy(), fd["ky"].cpu().numpy(), fd["kz"].cpu().numpy(), wi, wd)
    assert np.array_equal(got["votes"]["cls"].cpu().numpy(), votes["cls"])
    np.testing.assert_allclose(got["votes"]["weight"].cpu().numpy(), votes["weight"], rtol=1e-5, atol=1e-9)
    assert (got["cls"][:, 0].cpu().numpy() == nb["labels"]).all()


def test_knn_rule_matches_oracle(pkg, gpu, ora):
    import torch
    ctx, dev = gpu
    rng = np.random.default_rng(77)
    words = rng.random((600, 48)).astype(np.float32)
    wcls = rng.integers(0, 3, 600).astype(np.uint32)
    n = len(words)
    cb = pkg.capi.Codebook(ctx, words, np.arange(n + 1, dtype=np.uint32), np.zeros((n, 3), np.float32), wcls, np.zeros(n, np.uint32), 3,
                           np.ones(3, np.float32))
    q = (words[rng.integers(0, n, 400)] + 0.15 * rng.random((400, 48))).astype(np.float32)
    for metric in (0, 1):
        for thr in (0.8, 0.95):
            gi, gd = pkg.capi.knn_rule(ctx, cb, metric, torch.as_tensor(q).to(dev), thr)
            wi, wd = ora.knn_rule(metric, words, wcls, q, thr)
            assert np.array_equal(gi.cpu().numpy(), wi)
            g, w = gd.cpu().numpy(), wd
            assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(g[~np.isnan(g)], w[~np.isnan(w)])
    assert (wi == -1).any() and (wi >= 0).any()
    cb.set_word_class(np.zeros(n, np.uint32))            # one class everywhere -> the rule always accepts k1
    gi, _ = pkg.capi.knn_rule(ctx, cb, 0, torch.as_tensor(q).to(dev), 0.8)
    assert np.array_equal(gi.cpu().numpy(), ora.knn(0, words, q, 1)[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_functional():
    """The N > 1 path of bench.py executed once on the hardware this box has: `--gpus 2` with both ranks on cuda:0 over gloo (RCCL
    refuses two ranks on one device). Exercises spawn_ranks -> torch.distributed.run -> init_process_group -> per-rank training ->
    plan_shard -> per-rank detect -> record all-gather -> every object exactly once (asserted inside bench.py) -> rank-0 JSON line.
    A FUNCTIONAL test: the line says so, and its value is not a scaling number. Labels must equal the N = 1 run's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--objects", "64", "--steps", "1", "--warmup", "0", "--no-e2e", "--cpu-objects", "0", "--emit-labels", "--train-per-class", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + extra, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]                           # rank 0 prints ONE line
        return json.loads(lines[0])
    one = run([])
    two = run(["--gpus", "2", "--backend", "gloo", "--share-gpu"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["collective_world_size"] == 2 and two["config"]["objects_per_step_this_rank"] == 32
    assert "functional_test_only" in two
    assert len(two["labels_last_step"]) == 64 and two["labels_last_step"] == one["labels_last_step"]
    assert two["accuracy_last_step"] == one["accuracy_last_step"]
    # a WORLD_SIZE that does not match --gpus is refused (exit 2), as the contract says
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, cwd=root,
                       env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
