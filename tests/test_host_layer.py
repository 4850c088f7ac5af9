"""Tests of the C++ host mirror (ism3d.h): configuration / factory / error behaviour on the CPU, and on the GPU
train -> writeObject -> readObject -> detectBatch and the eval_tool harness against the Python harness and the labels."""
import json
import os
import subprocess

import numpy as np
import pytest

import host_binding as hb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "modelnet10_shot.ism")


def _cfg(**edits):
    j = json.load(open(CFG))["ObjectConfig"]
    for path, v in edits.items():
        node = j
        keys = path.split("/")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    return json.dumps(j)


def test_config_roundtrip_defaults_and_unknown_keys():
    m = hb.Model()
    j = json.loads(_cfg())
    j["Parameters"]["SetColorToZero"] = True                     # dead key in the reference's configs: silently ignored
    del j["Children"]["Voting"]["Parameters"]["MaxIter"]         # missing key -> default 1000 (+ warning)
    m.config_from_json(json.dumps(j))
    out = json.loads(m.config_to_json())
    assert out["Type"] == "ImplicitShapeModel"
    assert out["Children"]["Features"]["Type"] == "SHOT" and abs(out["Children"]["Features"]["Parameters"]["Radius"] - 0.4) < 1e-6
    assert out["Children"]["Voting"]["Parameters"]["MaxIter"] == 1000
    assert out["Children"]["Codebook"]["Children"]["ActivationStrategy"]["Type"] == "KNN"
    assert out["Children"]["Keypoints"]["Type"] == "VoxelGrid" and out["Children"]["Clustering"]["Type"] == "None"
    assert "SetColorToZero" not in out["Parameters"]
    m2 = hb.Model()
    m2.config_from_json(m.config_to_json())                      # what we write, we read
    assert json.loads(m2.config_to_json()) == out


@pytest.mark.parametrize("cfgfile", ["modelnet10_shot.ism", "qs_chi2_shot.ism", "kinect_cshot.ism"])
def test_shipped_configs_load(cfgfile):
    m = hb.Model()
    m.read(os.path.join(ROOT, "config", cfgfile), training=True)
    out = json.loads(m.config_to_json())
    assert out["Children"]["Features"]["Type"] in ("SHOT", "CSHOT")


def test_config_errors_match_reference_behaviour():
    for edits, needle in [({"Children/Features/Type": "PFH"}, "outside the MI355X hot path"),
                          ({"Children/Voting/Type": "Hough4D"}, "not built"),
                          ({"Children/Codebook/Children/ActivationStrategy/Type": "INN"}, "not built"),
                          ({"Children/Clustering/Type": "Agglomerative"}, "not built"),
                          ({"Children/Clustering": {"Type": "KMeansCount", "Parameters": {"CentersInit": "FLANN_CENTERS_BEST"}}}, None),
                          ({"Parameters/DistanceType": "Manhattan"}, "invalid distance type"),
                          ({"Children/Voting/Parameters/MaxIter": "many"}, "invalid type for parameter"),
                          ({"Parameters/UseSmoothing": True}, "not built")]:
        m = hb.Model()
        if needle is None:                                        # the value is checked when the clustering runs, as in the reference's trait
            m.config_from_json(_cfg(**edits)); continue
        with pytest.raises(hb.HostError) as e:
            m.config_from_json(_cfg(**edits))
        assert needle in str(e.value), (edits, str(e.value))
    m = hb.Model()
    j = json.loads(_cfg()); del j["Children"]["FeatureWeighting"]
    with pytest.raises(hb.HostError):                            # the reference's loader rejects a config without this child too
        m.config_from_json(json.dumps(j))


def test_pcd_reader_ascii_and_binary(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(50, 3)).astype(np.float32); nrm = rng.normal(size=(50, 3)).astype(np.float32)
    rgba = rng.integers(0, 1 << 24, 50).astype(np.uint32)
    xyz[7, 0] = np.nan                                           # NaN points are removed on load
    for binary in (False, True):
        p = str(tmp_path / ("c%d.pcd" % binary))
        hb.write_pcd(p, xyz, nrm, rgba, binary=binary)
        m = hb.Model()
        m.config_from_json(_cfg())
        m.add_training_file(p, 0, 0)
    with pytest.raises(hb.HostError):
        hb.Model().add_training_file(str(tmp_path / "missing.pcd"), 0, 0)


def test_pcd_reader_three_encodings_agree(tmp_path):
    """ascii, binary and binary_compressed (LZF, field-major) files of the same cloud load to the same points; the compressed
    stream holds literal runs and back references (constant curvature column, repeated coordinates)."""
    rng = np.random.default_rng(1)
    xyz = rng.normal(size=(400, 3)).astype(np.float32); nrm = rng.normal(size=(400, 3)).astype(np.float32)
    xyz[100:200] = xyz[0:100]                                    # long matches for the encoder
    rgba = rng.integers(0, 1 << 24, 400).astype(np.uint32)
    xyz[7, 0] = np.nan                                           # removed on load in every encoding
    pa, pb, pc = (str(tmp_path / n) for n in ("a.pcd", "b.pcd", "c.pcd"))
    hb.write_pcd(pa, xyz, nrm, rgba, binary=False); hb.write_pcd(pb, xyz, nrm, rgba, binary=True); hb.write_pcd_compressed(pc, xyz, nrm, rgba)
    assert os.path.getsize(pc) < os.path.getsize(pb)             # the encoder did find its matches
    a, b, c = hb.load_cloud(pa), hb.load_cloud(pb), hb.load_cloud(pc)
    keep = ~np.isnan(xyz[:, 0])
    for got in (a, b, c):
        assert np.array_equal(got[0], xyz[keep]) and np.array_equal(got[1], nrm[keep]) and np.array_equal(got[2], rgba[keep])
    open(pc, "r+b").truncate(os.path.getsize(pc) - 9)            # a cut stream is an error, not a short cloud
    with pytest.raises(hb.HostError):
        hb.load_cloud(pc)


# ---- .ismd: the byte stream of boost::archive::binary_oarchive in the reference's field order (FORMAT UNPINNED:
#      no file written by the real reference exists here). The independent reader / writer below follows the cited save functions
#      with struct.pack, not the host library's code.
import struct


class _BoostWriter:
    def __init__(self, version=17):
        self.b = bytearray()
        self.b += struct.pack("<Q", 22) + b"serialization::archive" + struct.pack("<H", version) + bytes([4, 8, 4, 8]) + struct.pack("<i", 1)

    def i(self, v): self.b += struct.pack("<i", v)
    def u(self, v): self.b += struct.pack("<I", v)
    def f(self, v): self.b += struct.pack("<f", v)
    def s(self, t): self.b += struct.pack("<Q", len(t)) + t.encode()
    def vf(self, a): a = np.asarray(a, np.float32); self.b += struct.pack("<Q", len(a)) + a.tobytes()
    def vu(self, a): a = np.asarray(a, np.uint32); self.b += struct.pack("<Q", len(a)) + a.tobytes()


class _BoostReader:
    def __init__(self, data):
        self.d, self.o = data, 0
        assert self.q() == 22 and self.raw(22) == b"serialization::archive"
        self.version = struct.unpack_from("<H", self.raw(2))[0]
        assert self.raw(4) == bytes([4, 8, 4, 8]) and self.i() == 1

    def raw(self, n): r = self.d[self.o:self.o + n]; self.o += n; assert len(r) == n; return bytes(r)
    def q(self): return struct.unpack("<Q", self.raw(8))[0]
    def i(self): return struct.unpack("<i", self.raw(4))[0]
    def u(self): return struct.unpack("<I", self.raw(4))[0]
    def f(self): return struct.unpack("<f", self.raw(4))[0]
    def s(self): return self.raw(self.q()).decode()
    def vf(self): return np.frombuffer(self.raw(4 * self.q()), np.float32)
    def vu(self): return np.frombuffer(self.raw(4 * self.q()), np.uint32)


def _toy_codebook(rng, n_words=7, dim=5, n_classes=3):
    cnt = rng.integers(1, 4, n_words)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    nv = int(off[-1])
    vote_class = np.concatenate([np.sort(rng.integers(0, n_classes, c)) for c in cnt]).astype(np.uint32)
    cw = np.zeros(nv, np.float32)
    for w in range(n_words):                                     # one class weight per (word, class), as CodewordDistribution stores them
        for c in set(vote_class[off[w]:off[w + 1]]):
            cw[off[w]:off[w + 1]][vote_class[off[w]:off[w + 1]] == c] = rng.random()
    q = rng.normal(size=(nv, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return dict(words=rng.random((n_words, dim)).astype(np.float32), word_id=(np.arange(n_words) * 3 + 11).astype(np.int32),
                word_class=rng.integers(0, n_classes, n_words).astype(np.uint32), word_weight=rng.random(n_words).astype(np.float32),
                word_keypoint=rng.normal(size=(n_words, 3)).astype(np.float32), vote_offsets=off, vote_xyz=rng.normal(size=(nv, 3)).astype(np.float32),
                vote_weight=rng.random(nv).astype(np.float32), vote_class_weight=cw, vote_class=vote_class,
                vote_instance=rng.integers(0, 9, nv).astype(np.uint32), vote_bbox_quat=q.astype(np.float32),
                vote_bbox_size=rng.random((nv, 3)).astype(np.float32), class_sigma=rng.random(n_classes).astype(np.float32))


def test_ismd_is_a_boost_binary_archive_in_the_reference_field_order(tmp_path):
    rng = np.random.default_rng(3)
    cb = _toy_codebook(rng)
    m = hb.Model()
    m.config_from_json(_cfg())
    m.set_codebook(cb, 3)
    m.set_labels(["chair", "table", "lamp"], ["chair_a", "chair_b", "table_a", "lamp_a"], [0, 0, 1, 2])
    m.set_dimensions(0, 0.9, 0.5, 0.01, 0.02); m.set_dimensions(2, 1.1, 0.7, 0.03, 0.04)
    path = str(tmp_path / "toy.ism")
    m.write(path)
    data = open(str(tmp_path / "toy.ismd"), "rb").read()
    # header: 8-byte length + signature, 2-byte library version, the four type sizes, the endianness probe (40 bytes)
    assert data[:40] == struct.pack("<Q", 22) + b"serialization::archive" + struct.pack("<H", 17) + bytes([4, 8, 4, 8]) + struct.pack("<i", 1)
    r = _BoostReader(data)
    # ImplicitShapeModel::iSaveData (implicit_shape_model.cpp:1144-1179)
    assert [(r.u(), r.u()) for _ in range(r.u())] == [(0, 0), (1, 0), (2, 1), (3, 2)]
    # Codebook::iSaveData (codebook.cpp:739-761)
    assert r.i() == 7
    for w in range(7):
        v0, v1 = int(cb["vote_offsets"][w]), int(cb["vote_offsets"][w + 1])
        assert (r.i(), r.i()) == (int(cb["word_id"][w]), 1) and r.f() == cb["word_weight"][w]                   # Codeword (codeword.cpp:71-83)
        assert np.array_equal(r.vf(), cb["words"][w]) and r.i() == int(cb["word_class"][w])
        assert [r.f(), r.f(), r.f()] == cb["word_keypoint"][w].tolist()
        assert r.i() == v1 - v0                                                                                   # CodewordDistribution (:349-391)
        assert np.array_equal(np.asarray([r.f() for _ in range(3 * (v1 - v0))], np.float32), cb["vote_xyz"][v0:v1].reshape(-1))
        assert np.array_equal(r.vf(), cb["vote_weight"][v0:v1]) and np.array_equal(r.vu(), cb["vote_class"][v0:v1]) and np.array_equal(r.vu(), cb["vote_instance"][v0:v1])
        classes = sorted(set(cb["vote_class"][v0:v1].tolist()))
        assert r.i() == len(classes)
        for c in classes:
            assert r.i() == c and r.f() == cb["vote_class_weight"][v0:v1][cb["vote_class"][v0:v1] == c][0]
        assert r.i() == v1 - v0
        for v in range(v0, v1):
            assert [r.f() for _ in range(4)] == cb["vote_bbox_quat"][v].tolist() and [r.f() for _ in range(3)] == cb["vote_bbox_size"][v].tolist()
    assert r.i() == 3 and [(r.i(), r.f()) for _ in range(3)] == [(c, float(cb["class_sigma"][c])) for c in range(3)]
    # Voting::iSaveData (voting.cpp:559-614): dimension map, variance map, no global features
    assert r.u() == 2 and (r.u(), r.f(), r.f()) == (0, np.float32(0.9), np.float32(0.5)) and (r.u(), r.f(), r.f()) == (2, np.float32(1.1), np.float32(0.7))
    assert r.u() == 2 and (r.u(), r.f(), r.f()) == (0, np.float32(0.01), np.float32(0.02)) and (r.u(), r.f(), r.f()) == (2, np.float32(0.03), np.float32(0.04))
    assert r.u() == 0
    assert r.u() == 3 and [r.s() for _ in range(3)] == ["chair", "table", "lamp"]
    assert r.u() == 4 and [r.s() for _ in range(4)] == ["chair_a", "chair_b", "table_a", "lamp_a"]
    assert r.o == len(data)
    # write -> read round trip through the host library
    m2 = hb.Model()
    m2.read(path)
    got = m2.codebook_all()
    for k, v in cb.items():
        assert np.array_equal(got[k].reshape(-1), np.asarray(v).reshape(-1)), k
    assert m2.label(0, 1) == "table" and m2.label(1, 3) == "lamp_a" and np.allclose(m2.dimensions(2), [1.1, 0.7, 0.03, 0.04]) and m2.dimensions(1) is None


def test_ismd_written_the_reference_way_is_read(tmp_path):
    """A data file assembled with struct.pack as the reference's save functions would write it: arbitrary ascending codeword ids,
    a class without a sigma entry, stored global features (which must be consumed and dropped), Boost library version 18."""
    rng = np.random.default_rng(4)
    cb = _toy_codebook(rng, n_words=5, dim=4, n_classes=3)
    w = _BoostWriter(version=18)
    w.u(2); w.u(0); w.u(0); w.u(1); w.u(2)                                  # instance -> class
    w.i(5)
    for k in range(5):
        v0, v1 = int(cb["vote_offsets"][k]), int(cb["vote_offsets"][k + 1])
        w.i(int(cb["word_id"][k])); w.i(1); w.f(cb["word_weight"][k]); w.vf(cb["words"][k]); w.i(int(cb["word_class"][k]))
        for x in cb["word_keypoint"][k]: w.f(x)
        w.i(v1 - v0)
        for x in cb["vote_xyz"][v0:v1].reshape(-1): w.f(x)
        w.vf(cb["vote_weight"][v0:v1]); w.vu(cb["vote_class"][v0:v1]); w.vu(cb["vote_instance"][v0:v1])
        classes = sorted(set(cb["vote_class"][v0:v1].tolist()))
        w.i(len(classes))
        for c in classes:
            w.i(c); w.f(cb["vote_class_weight"][v0:v1][cb["vote_class"][v0:v1] == c][0])
        w.i(v1 - v0)
        for v in range(v0, v1):
            for x in cb["vote_bbox_quat"][v]: w.f(x)
            for x in cb["vote_bbox_size"][v]: w.f(x)
    w.i(2); w.i(0); w.f(cb["class_sigma"][0]); w.i(2); w.f(cb["class_sigma"][2])   # class 1 has no sigma entry
    w.u(1); w.u(0); w.f(0.8); w.f(0.4); w.u(1); w.u(0); w.f(0.1); w.f(0.2)
    w.u(1); w.u(0); w.u(1); w.u(2)                                            # global features: class 0, one cloud, two features
    for _ in range(2):
        for x in range(9): w.f(float(x))
        w.vf(rng.random(6)); w.f(0.3); w.u(1)
    w.u(2); w.s("mug"); w.s("bowl"); w.u(2); w.s("mug_1"); w.s("bowl_1")
    open(str(tmp_path / "ref.ismd"), "wb").write(bytes(w.b))
    j = {"ObjectConfig": json.loads(_cfg()), "ObjectData": "ref.ismd"}
    open(str(tmp_path / "ref.ism"), "w").write(json.dumps(j))
    m = hb.Model()
    m.read(str(tmp_path / "ref.ism"))
    got = m.codebook_all()
    for k in ("words", "word_id", "word_class", "vote_offsets", "vote_xyz", "vote_weight", "vote_class_weight", "vote_class", "vote_instance", "vote_bbox_quat"):
        assert np.array_equal(got[k].reshape(-1), np.asarray(cb[k]).reshape(-1)), k
    assert got["class_sigma"][0] == cb["class_sigma"][0] and got["class_sigma"][1] == 1.0 and got["class_sigma"][2] == cb["class_sigma"][2]
    assert m.label(0, 1) == "bowl" and np.allclose(m.dimensions(0), [0.8, 0.4, 0.1, 0.2])
    # damaged files are errors, not crashes: cut anywhere, absurd lengths, wrong signature
    good = bytes(w.b)
    for cut in (10, 39, 41, 60, 200, len(good) - 3):
        open(str(tmp_path / "ref.ismd"), "wb").write(good[:cut])
        with pytest.raises(hb.HostError):
            hb.Model().read(str(tmp_path / "ref.ism"))
    bad = bytearray(good); bad[60:68] = struct.pack("<Q", 1 << 60)             # first vector length
    open(str(tmp_path / "ref.ismd"), "wb").write(bytes(bad))
    with pytest.raises(hb.HostError):
        hb.Model().read(str(tmp_path / "ref.ism"))
    open(str(tmp_path / "ref.ismd"), "wb").write(b"ISMDAMD1" + good[8:])
    with pytest.raises(hb.HostError):
        hb.Model().read(str(tmp_path / "ref.ism"))


def _dataset(pkg, n_classes, n_train, n_test):
    syn = pkg.synthetic
    train = syn.Dataset(n_classes, n_train, split=0, n_points=4096, leaf=0.2)
    test = syn.Dataset(n_classes, n_test, split=1, n_points=4096, leaf=0.2)
    return train, test


@pytest.mark.gpu
def test_host_train_write_read_detect_matches_python_harness(pkg, gpu, tmp_path):
    import torch
    ctx, dev = gpu
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    m = hb.Model()
    m.config_from_json(_cfg())
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    # the Python harness on the same data (keypoints = VoxelGrid leaf 0.2 in both)
    cfg = pkg.pipeline.IsmConfig(n_classes=3)
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    cb = rec.train([pkg.pipeline.DeviceBatch(train.batch(order), dev)], instance_ids=order)
    words, vxyz, vcls, sigma = m.codebook(352, 3)
    assert words.shape == cb["words"].shape
    # the two hosts compute the voxel-grid keypoints with different accumulation widths (float vs double): 1-ulp keypoint
    # differences move descriptor entries by a few 1e-6
    np.testing.assert_allclose(words, cb["words"], atol=2e-5)
    np.testing.assert_allclose(vxyz, cb["vote_xyz"], atol=1e-4)
    assert np.array_equal(vcls, cb["vote_class"])
    np.testing.assert_allclose(sigma, cb["class_sigma"], rtol=1e-3)
    # persistence: .ism (JSON) + .ismd
    path = str(tmp_path / "model.ism")
    m.write(path)
    assert os.path.exists(str(tmp_path / "model.ismd")) and json.load(open(path))["ObjectData"] == "model.ismd"
    m2 = hb.Model()
    m2.read(path)
    assert m2.codebook_size() == m.codebook_size()
    nb = test.batch(range(6))
    got = m2.detect_batch(nb["pt_off"], nb["xyz"], nb["normals"], max_maxima=8)
    want = rec.detect(pkg.pipeline.DeviceBatch(nb, dev))
    assert np.array_equal(got["n"], np.minimum(want["n"].cpu().numpy(), 8))
    k = 4
    assert np.array_equal(got["cls"][:, :k], want["cls"][:, :k].cpu().numpy())
    np.testing.assert_allclose(got["weight"][:, :k], want["weight"][:, :k].cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(got["pos"][:, :k], want["pos"][:, :k].cpu().numpy(), atol=2e-3)
    assert (got["cls"][:, 0] == nb["labels"]).all()


@pytest.mark.gpu
def test_host_hough3d_voting_matches_python_harness(pkg, gpu):
    """Factory<Voting> accepts "Hough3D" (voting_factory.h:20-29): the C++ host with the reference's parameter names against
    the Python harness on the same data"""
    ctx, dev = gpu
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    m = hb.Model()
    m.config_from_json(_cfg(**{"Children/Voting/Type": "Hough3D",
                               "Children/Voting/Parameters": {"UseInterpolation": True, "MinCoord": [-3, -3, -3], "MaxCoord": [3, 3, 3],
                                                              "BinSize": [0.4, 0.4, 0.4], "RelThreshold": 0.5, "MinThreshold": 0.0,
                                                              "MinVotesThreshold": 1, "BestK": -1}}))
    out = json.loads(m.config_to_json())
    assert out["Children"]["Voting"]["Type"] == "Hough3D" and out["Children"]["Voting"]["Parameters"]["BinSize"] == [0.4, 0.4, 0.4]
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    cfg = pkg.pipeline.IsmConfig(n_classes=3, voting="Hough3D", hough_bin_size=0.4, hough_rel_threshold=0.5,
                                 hough_min_coord=(-3.0, -3.0, -3.0), hough_max_coord=(3.0, 3.0, 3.0))
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    rec.train([pkg.pipeline.DeviceBatch(train.batch(order), dev)], instance_ids=order)
    nb = test.batch(range(6))
    got = m.detect_batch(nb["pt_off"], nb["xyz"], nb["normals"], max_maxima=8)
    want = rec.detect(pkg.pipeline.DeviceBatch(nb, dev))
    assert np.array_equal(got["n"], np.minimum(want["n"].cpu().numpy(), 8))
    k = 3
    assert np.array_equal(got["cls"][:, :k], want["cls"][:, :k].cpu().numpy())
    np.testing.assert_allclose(got["weight"][:, :k], want["weight"][:, :k].cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(got["pos"][:, :k], want["pos"][:, :k].cpu().numpy(), atol=1e-3)
    assert (got["cls"][:, 0] == nb["labels"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", ["average_rotation+merge", "single_object_bandwidth", "single_object_radius", "single_object_space", "many_maxima"])
def test_host_voting_options_match_python_harness(pkg, gpu, opts):
    """Voting options the shipped configs set and round 2 refused or ignored, through the C++ host with the reference's key names:
    AverageRotation (bounding-box quaternion per maximum), MaxFilterType "Merge", SingleObjectMode with SingleObjectMaxType
    BandwidthVotes / ModelRadiusVotes / VotingSpaceVotes; and an object with more maxima than the host's first 32-slot request
    (BestK -1, MinThreshold 0, tiny bandwidth): the reference returns all of them, so must the host (second request with 1024)."""
    ctx, dev = gpu
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    vp = {"MinThreshold": 0.0, "MinVotesThreshold": 1, "BestK": -1, "Bandwidth": 0.6}
    kw = dict(n_classes=3, max_maxima=1024 if opts == "many_maxima" else 16)
    if opts == "average_rotation+merge":
        vp.update(AverageRotation=True, MaxFilterType="Merge"); kw.update(average_rotation=True, max_filter="Merge")
    elif opts.startswith("single_object"):
        t = {"single_object_bandwidth": "BandwidthVotes", "single_object_radius": "ModelRadiusVotes", "single_object_space": "VotingSpaceVotes"}[opts]
        vp.update(SingleObjectMode=True, SingleObjectMaxType=t, MaxFilterType="Simple")          # the filter is skipped in single-object mode
        kw.update(single_object_mode=True, single_object_max_type=t, max_filter="Simple")
    else:
        vp.update(Bandwidth=0.08, MaximaSuppression="Suppress"); kw.update(bandwidth=0.08, maxima_suppression="Suppress")
    m = hb.Model()
    m.config_from_json(_cfg(**{"Children/Voting/Parameters": vp}))
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    rec = pkg.pipeline.Recognizer(ctx, pkg.pipeline.IsmConfig(**kw))
    rec.train([pkg.pipeline.DeviceBatch(train.batch(order), dev)], instance_ids=order)
    if opts == "average_rotation+merge":
        # the harness trains without bounding boxes; the host stores box quaternion * conj(q(LRF)) and the box size per vote
        # (codeword_distribution.cpp:63-70): run the harness on the HOST's codebook so that both see the same vote boxes
        cbh = m.codebook_all()
        assert cbh["words"].shape == rec.cb_host["words"].shape and np.abs(cbh["words"] - rec.cb_host["words"]).max() < 1e-4
        rec.load_codebook({k: cbh[k] for k in ("words", "vote_offsets", "vote_xyz", "vote_class", "vote_instance", "class_sigma", "word_weight",
                                               "vote_weight", "vote_class_weight", "vote_bbox_quat", "vote_bbox_size", "word_class")})
    nb = test.batch(range(6))
    M = kw["max_maxima"]
    got = m.detect_batch(nb["pt_off"], nb["xyz"], nb["normals"], max_maxima=M)
    want = rec.detect(pkg.pipeline.DeviceBatch(nb, dev))
    ctx.sync()
    wn = want["n"].cpu().numpy()
    assert np.array_equal(got["n_total"], wn), (got["n_total"], wn)
    if opts == "many_maxima":
        assert wn.max() > 32, wn                                          # the construction really exceeded the first request
    for o in range(6):
        k = int(wn[o])
        assert np.array_equal(got["cls"][o, :k], want["cls"][o, :k].cpu().numpy())
        # (VotingSpaceVotes: the bandwidth IS the distance of the farthest vote, so that vote sits on the d^2 < h^2 edge and the 1e-5
        # differences between the host's and the harness's keypoints can flip it -- the device itself is checked against the oracle
        # on identical votes in test_single_object_max_types_match_oracle)
        np.testing.assert_allclose(got["weight"][o, :k], want["weight"][o, :k].cpu().numpy(), atol=1e-2 if opts == "single_object_space" else 1e-4)
        np.testing.assert_allclose(got["pos"][o, :k], want["pos"][o, :k].cpu().numpy(), atol=1e-3)
        if opts == "average_rotation+merge":
            a, b = got["bbox_quat"][o, :k].astype(np.float64), want["bbox_quat"][o, :k].cpu().numpy().astype(np.float64)
            assert np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max() < 1e-3
            assert np.abs(np.linalg.norm(a, axis=1) - 1).max() < 1e-4
    if opts.startswith("single_object"):
        assert (wn <= 3).all() and wn.sum() >= 6                             # at most one maximum per class, at the cloud centroid
        cen = np.stack([nb["xyz"][nb["pt_off"][o]:nb["pt_off"][o + 1]].astype(np.float64).mean(0) for o in range(6)])
        for o in range(6):
            for i in range(int(wn[o])):
                assert np.abs(got["pos"][o, i] - cen[o]).max() < 1e-4


@pytest.mark.gpu
def test_host_partial_shot_and_bandwidth_types(pkg, gpu, tmp_path):
    """UsePartialShot / PartialShotType, BinOrBandwidthType ObjectRadius and MaxFilterType Simple through the C++ host (the size
    hints travel in the .ismd): the model still tells the synthetic classes apart, and the per-class hints are the training objects'."""
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    m = hb.Model()
    m.config_from_json(_cfg(**{"Children/Codebook/Parameters/UsePartialShot": True, "Children/Codebook/Parameters/PartialShotType": "top",
                               "Children/Voting/Parameters/BinOrBandwidthType": "ObjectRadius", "Children/Voting/Parameters/BinOrBandwidthFactor": 0.6,
                               "Children/Voting/Parameters/MaxFilterType": "Simple"}))
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    d0 = m.dimensions(0)
    radii = [np.linalg.norm(train.get(i)["xyz"] - train.get(i)["xyz"].mean(0), axis=1).max() for i in order if train.label(i) == 0]
    assert abs(d0[0] - np.mean(radii)) < 1e-4 and d0[1] > 0
    path = str(tmp_path / "p.ism")
    m.write(path)
    m2 = hb.Model()
    m2.read(path)
    assert np.allclose(m2.dimensions(0), d0)
    nb = test.batch(range(6))
    got = m2.detect_batch(nb["pt_off"], nb["xyz"], nb["normals"], max_maxima=4)
    assert (got["cls"][:, 0] == nb["labels"]).all()


@pytest.mark.gpu
def test_host_kmeans_clustering_matches_python_harness(pkg, gpu, tmp_path):
    """Clustering KMeansCount / KMeansFactor / KMeansThumbRule / KMeansHartigan through the C++ host (clustering/*.cpp): codewords are
    the device k-means' centres, KNN activation with K = 2 fans every feature's vote out; the codebook equals the Python harness's on
    the same data (both drive ismhip_kmeans + ismhip_train_activate), survives the .ismd round trip and classifies the test objects."""
    import torch
    ctx, dev = gpu
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    edits = {"Children/Clustering": {"Type": "KMeansCount", "Parameters": {"ClusterCount": 300, "Iterations": 25, "Seed": 4}},
             "Children/Codebook/Children/ActivationStrategy/Parameters/K": 2,
             "Children/Codebook/Parameters/UseVoteWeight": True, "Children/Codebook/Parameters/UseMatchingWeight": True}
    m = hb.Model()
    m.config_from_json(_cfg(**edits))
    assert json.loads(m.config_to_json())["Children"]["Clustering"]["Parameters"]["CentersInit"] == "FLANN_CENTERS_KMEANSPP"
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    cfg = pkg.pipeline.IsmConfig(n_classes=3, k=2, clustering="KMeansCount", cluster_count=300, kmeans_iterations=25, kmeans_seed=4,
                                 use_vote_weight=True, use_matching_weight=True)
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    cb = rec.train([pkg.pipeline.DeviceBatch(train.batch(order), dev)], instance_ids=order)
    got = m.codebook_all()
    assert got["words"].shape == cb["words"].shape == (300, 352)
    # (1-ulp keypoint differences between the two hosts' voxel grids move descriptor entries by a few 1e-6, see the test above)
    np.testing.assert_allclose(got["words"], cb["words"], atol=2e-5)
    assert np.array_equal(got["vote_offsets"], cb["vote_offsets"]) and np.array_equal(got["vote_class"], cb["vote_class"])
    np.testing.assert_allclose(got["vote_xyz"], cb["vote_xyz"], atol=1e-4)
    np.testing.assert_allclose(got["vote_weight"], cb["vote_weight"], atol=1e-4)
    assert got["vote_offsets"][-1] == 2 * len(rec.cluster_indices)
    path = str(tmp_path / "km.ism")
    m.write(path)
    m2 = hb.Model()
    m2.read(path)
    nb = test.batch(range(6))
    out = m2.detect_batch(nb["pt_off"], nb["xyz"], nb["normals"], max_maxima=4)
    assert (out["cls"][:, 0] == nb["labels"]).all()
    # the other cluster-count rules: count from the feature total
    n_feat = len(rec.cluster_indices)
    for kind, params, want in [("KMeansFactor", {"ClusterFactor": 0.1, "Iterations": 3}, int(round(n_feat * np.float32(0.1)))),
                               ("KMeansThumbRule", {"Iterations": 3}, int(round(float(np.sqrt(np.float32(n_feat / 2.0)))))),
                               ("KMeansHartigan", {"MaxK": 4, "Iterations": 5}, None)]:
        mk = hb.Model()
        mk.config_from_json(_cfg(**{"Children/Clustering": {"Type": kind, "Parameters": params}, "Children/Codebook/Children/ActivationStrategy/Parameters/K": 2}))
        for i in order:
            o = train.get(i)
            mk.add_training(o["xyz"], o["normals"], o["label"], i)
        mk.train()
        if want is not None:
            assert mk.codebook_size() == want, (kind, mk.codebook_size(), want)
        else:
            assert 1 <= mk.codebook_size() <= 4


@pytest.mark.gpu
def test_host_clouds_without_normals_get_them_on_the_device(pkg, gpu, monkeypatch):
    """Inputs whose first normal is zero/NaN count as normal-less (implicit_shape_model.cpp:615-625): the host layer then estimates
    normals on the device (ConsistentNormalsMethod 2: inverted z axis of a SHOT frame with NormalRadius at every point), drops points
    with NaN normals and carries on. Trained and tested without a single input normal, the synthetic objects must still be told apart."""
    train, test = _dataset(pkg, 3, 9, 6)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    m = hb.Model()
    m.config_from_json(_cfg(**{"Parameters/NormalRadius": 0.15}))
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], np.zeros_like(o["normals"]), o["label"], i)
    m.train()
    assert m.codebook_size() > 100
    nb = test.batch(range(6))
    got = m.detect_batch(nb["pt_off"], nb["xyz"], np.zeros_like(nb["normals"]), max_maxima=4)
    assert (got["cls"][:, 0] == nb["labels"]).all()
    # the batch above never left HBM between normal estimation and description (ismhip_filter_normals); the host-side filter
    # (mixed batches, host keypoint detectors) must give the same maxima
    monkeypatch.setenv("ISM3D_HOST_NORMAL_FILTER", "1")
    via_host = m.detect_batch(nb["pt_off"], nb["xyz"], np.zeros_like(nb["normals"]), max_maxima=4)
    monkeypatch.delenv("ISM3D_HOST_NORMAL_FILTER")
    for k in ("cls", "weight", "pos"):
        if k in got:
            np.testing.assert_array_equal(got[k], via_host[k])
    # a sparse tail of isolated points: their frames are invalid (< 5 neighbours), their normals NaN, and they leave the cloud on the device
    xyz2 = nb["xyz"].copy()
    first = int(nb["pt_off"][1])
    xyz2[first - 3:first] += np.array([[40, 0, 0], [0, 55, 0], [0, 0, 70]], np.float32)
    got2 = m.detect_batch(nb["pt_off"], xyz2, np.zeros_like(nb["normals"]), max_maxima=4)
    monkeypatch.setenv("ISM3D_HOST_NORMAL_FILTER", "1")
    via_host2 = m.detect_batch(nb["pt_off"], xyz2, np.zeros_like(nb["normals"]), max_maxima=4)
    monkeypatch.delenv("ISM3D_HOST_NORMAL_FILTER")
    for k in ("cls", "weight", "pos"):
        if k in got2:
            np.testing.assert_array_equal(got2[k], via_host2[k])
    assert (got2["cls"][:, 0] == nb["labels"]).all()
    m.config_from_json(_cfg(**{"Parameters/NormalRadius": 0.15, "Parameters/ConsistentNormalsMethod": 7}))
    with pytest.raises(hb.HostError, match="ConsistentNormalsMethod 7 is not built"):
        m.detect_batch(nb["pt_off"], nb["xyz"], np.zeros_like(nb["normals"]), max_maxima=4)
    # method 1 (PCA normals pointing away from the centroid): train and detect without input normals
    m1 = hb.Model()
    m1.config_from_json(_cfg(**{"Parameters/NormalRadius": 0.15, "Parameters/ConsistentNormalsMethod": 1}))
    for i in order:
        o = train.get(i)
        m1.add_training(o["xyz"], np.zeros_like(o["normals"]), o["label"], i)
    m1.train()
    got = m1.detect_batch(nb["pt_off"], nb["xyz"], np.zeros_like(nb["normals"]), max_maxima=4)
    assert (got["cls"][:, 0] == nb["labels"]).all()


@pytest.mark.gpu
def test_host_detect_file_decides_normals_from_the_first_point(pkg, gpu, tmp_path, capfd):
    """ImplicitShapeModel::detect(filename) (implicit_shape_model.cpp:556-573, 615-625): the cloud "has normals" when its FIRST point
    carries one; otherwise they are estimated (on the device). A dense HEIGHT > 1 file stays organized in the reference, which then
    takes IntegralImageNormalEstimation -- not built here: the unorganized method runs and says so."""
    train, test = _dataset(pkg, 3, 9, 3)
    order = sorted(range(9), key=lambda i: (train.label(i), i))
    m = hb.Model()
    m.config_from_json(_cfg(**{"Parameters/NormalRadius": 0.15}))
    for i in order:
        o = train.get(i)
        m.add_training(o["xyz"], o["normals"], o["label"], i)
    m.train()
    for i in range(3):
        o = test.get(i)
        n = len(o["xyz"]) // 2 * 2
        xyz, nrm = o["xyz"][:n], o["normals"][:n]
        off = np.array([0, n], np.uint32)
        with_n = str(tmp_path / f"with_{i}.pcd"); hb.write_pcd(with_n, xyz, nrm, binary=True)
        cls, w = m.detect_file(with_n)
        ref = m.detect_batch(off, xyz, nrm, max_maxima=4)
        assert cls == o["label"] == ref["cls"][0, 0] and w == ref["weight"][0, 0]
        # first normal zero -> every normal of the file is ignored and estimated again, exactly as for an all-zero input
        first0 = nrm.copy(); first0[0] = 0
        no_n = str(tmp_path / f"first0_{i}.pcd"); hb.write_pcd(no_n, xyz, first0, binary=True)
        cls0, w0 = m.detect_file(no_n)
        ref0 = m.detect_batch(off, xyz, np.zeros_like(nrm), max_maxima=4)
        assert cls0 == ref0["cls"][0, 0] and w0 == ref0["weight"][0, 0]
        capfd.readouterr()
        org = str(tmp_path / f"organized_{i}.pcd"); hb.write_pcd(org, xyz, np.zeros_like(nrm), binary=True, height=2)
        cls1, w1 = m.detect_file(org)
        assert (cls1, w1) == (cls0, w0)
        assert "organized input cloud without normals" in capfd.readouterr().out
        m.detect_file(no_n)
        assert "organized input cloud" not in capfd.readouterr().out


@pytest.mark.gpu
def test_eval_tool_end_to_end(pkg, tmp_path):
    train, test = _dataset(pkg, 3, 6, 6)
    names = ["chair", "table", "lamp"]
    def dump(ds, n, tag, header):
        lines = [header]
        for i in range(n):
            o = ds.get(i)
            p = str(tmp_path / f"{tag}_{i}.pcd")
            hb.write_pcd(p, o["xyz"], o["normals"], binary=True)
            lines.append(f"{p} {names[o['label']]}")
        lp = str(tmp_path / f"{tag}_list.txt")
        open(lp, "w").write("\n".join(lines) + "\n")
        return lp
    ltrain = dump(train, 6, "train", "# train")
    ltest = dump(test, 6, "test", "# test")
    tool = os.path.join(ROOT, "point-cloud-donkey_amd", "eval_tool")
    out = str(tmp_path / "out")
    r = subprocess.run([tool, "-t", CFG, "-f", ltrain, "-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    model = os.path.join(out, "modelnet10_shot.ism")
    assert os.path.exists(model)
    r = subprocess.run([tool, "-d", model, "-f", ltest, "-o", out, "-b", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    s = open(os.path.join(out, "summary.txt")).read()
    assert s.count("ground truth class:") == 6 and "Accuracy: 100 %" in s and "compute features:" in s and "0: chair" in s
    # wrong mode is refused like the reference does
    r = subprocess.run([tool, "-d", model, "-f", ltrain, "-o", out], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
