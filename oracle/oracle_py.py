"""ctypes/numpy binding of the CPU oracle (oracle/libism_oracle.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package. See oracle/ism_oracle.h for the parity statement (parity unpinned by the reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libism_oracle.so")
_lib = None


class MaximaParams(C.Structure):
    _fields_ = [("n_classes", C.c_int), ("class_bandwidth", C.c_void_p), ("bandwidth", C.c_float),
                ("threshold", C.c_float), ("max_iter", C.c_int), ("kernel", C.c_int), ("suppression", C.c_int),
                ("min_votes_threshold", C.c_int), ("min_threshold", C.c_float), ("best_k", C.c_int),
                ("max_maxima", C.c_int), ("max_filter", C.c_int),
                ("vote_bbox_quat", C.c_void_p), ("max_bbox_quat_out", C.c_void_p), ("single_object_max_type", C.c_int),
                ("object_centroid", C.c_void_p), ("object_radius", C.c_void_p)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ismref_distance.restype = C.c_float
        _lib = L
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _u(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _i(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


def set_num_threads(n):
    lib().ismref_set_num_threads(C.c_int(n))


def get_num_threads():
    return lib().ismref_get_num_threads()


def radius_search(x, y, z, q, radius):
    x, y, z = _f(x), _f(y), _f(z)
    n = len(x)
    idx = np.empty(n, np.int32)
    d2 = np.empty(n, np.float32)
    m = lib().ismref_radius_search(C.c_int(n), _p(x), _p(y), _p(z), C.c_float(q[0]), C.c_float(q[1]), C.c_float(q[2]),
                                   C.c_float(radius), C.c_int(n), _p(idx), _p(d2))
    return idx[:m].copy(), d2[:m].copy()


def shot_lrf(pt_off, x, y, z, kp_off, kx, ky, kz, radius):
    po, ko = _u(pt_off), _u(kp_off)
    x, y, z, kx, ky, kz = map(_f, (x, y, z, kx, ky, kz))
    out = np.empty((int(ko[-1]), 9), np.float32)
    lib().ismref_shot_lrf(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(ko), _p(kx), _p(ky), _p(kz), C.c_float(radius), _p(out))
    return out


def shot352(pt_off, x, y, z, nx, ny, nz, kp_off, kx, ky, kz, lrf, radius):
    po, ko = _u(pt_off), _u(kp_off)
    x, y, z, nx, ny, nz, kx, ky, kz, lrf = map(_f, (x, y, z, nx, ny, nz, kx, ky, kz, lrf))
    n = int(ko[-1])
    out = np.empty((n, 352), np.float32)
    cnt = np.empty(n, np.uint32)
    lib().ismref_shot352(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(ko), _p(kx), _p(ky), _p(kz),
                         _p(lrf), C.c_float(radius), _p(out), _p(cnt))
    return out, cnt


def cshot1344(pt_off, x, y, z, nx, ny, nz, rgba, kp_off, kx, ky, kz, kp_rgba, lrf, radius):
    po, ko = _u(pt_off), _u(kp_off)
    x, y, z, nx, ny, nz, kx, ky, kz, lrf = map(_f, (x, y, z, nx, ny, nz, kx, ky, kz, lrf))
    rgba, kp_rgba = _u(rgba), _u(kp_rgba)
    n = int(ko[-1])
    out = np.empty((n, 1344), np.float32)
    cnt = np.empty(n, np.uint32)
    rc = lib().ismref_cshot1344(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(rgba), _p(ko), _p(kx),
                                _p(ky), _p(kz), _p(kp_rgba), _p(lrf), C.c_float(radius), _p(out), _p(cnt))
    assert rc == 0
    return out, cnt


def fpfh33(pt_off, x, y, z, nx, ny, nz, kp_off, kx, ky, kz, radius):
    po, ko = _u(pt_off), _u(kp_off)
    x, y, z, nx, ny, nz, kx, ky, kz = map(_f, (x, y, z, nx, ny, nz, kx, ky, kz))
    n = int(ko[-1])
    out = np.empty((n, 33), np.float32)
    cnt = np.empty(n, np.uint32)
    lib().ismref_fpfh33(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(ko), _p(kx), _p(ky), _p(kz),
                        C.c_float(radius), _p(out), _p(cnt))
    return out, cnt


def centroids(pt_off, x, y, z):
    po = _u(pt_off)
    x, y, z = map(_f, (x, y, z))
    out = np.empty((len(po) - 1, 3), np.float32)
    lib().ismref_centroids(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(out))
    return out


def center_dist(pt_off, x, y, z, kp_off, kx, ky, kz):
    po, ko = _u(pt_off), _u(kp_off)
    x, y, z, kx, ky, kz = map(_f, (x, y, z, kx, ky, kz))
    out = np.empty(int(ko[-1]), np.float32)
    lib().ismref_center_dist(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), _p(ko), _p(kx), _p(ky), _p(kz), _p(out))
    return out


def rgb2lab(rgba):
    L, a, b = C.c_float(), C.c_float(), C.c_float()
    lib().ismref_rgb2lab(C.c_uint32(int(rgba)), C.byref(L), C.byref(a), C.byref(b))
    return L.value, a.value, b.value


def pair_features(p1, n1, p2, n2):
    p1, n1, p2, n2 = map(_f, (p1, n1, p2, n2))
    out = np.zeros(4, np.float32)
    ok = lib().ismref_pair_features(_p(p1), _p(n1), _p(p2), _p(n2), _p(out))
    return bool(ok), out


def distance(metric, a, b):
    a, b = _f(a), _f(b)
    return float(lib().ismref_distance(C.c_int(metric), C.c_int(len(a)), _p(a), _p(b)))


def knn(metric, words, q, k=1):
    words, q = _f(words), _f(q)
    nq = q.shape[0]
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    lib().ismref_knn(C.c_int(metric), C.c_int(words.shape[0]), C.c_int(words.shape[1]), _p(words), C.c_int(nq), _p(q), C.c_int(k),
                     _p(idx), _p(dist))
    return idx, dist


def knn_ratio(metric, words, q, thr):
    words, q = _f(words), _f(q)
    nq = q.shape[0]
    idx = np.empty((nq, 1), np.int32)
    dist = np.empty((nq, 1), np.float32)
    lib().ismref_knn_ratio(C.c_int(metric), C.c_int(words.shape[0]), C.c_int(words.shape[1]), _p(words), C.c_int(nq), _p(q), C.c_float(thr),
                           _p(idx), _p(dist))
    return idx, dist


def knn_rule(metric, words, word_class, q, thr):
    words, q = _f(words), _f(q)
    nq = q.shape[0]
    idx = np.empty((nq, 1), np.int32)
    dist = np.empty((nq, 1), np.float32)
    lib().ismref_knn_rule(C.c_int(metric), C.c_int(words.shape[0]), C.c_int(words.shape[1]), _p(words), _p(_u(word_class)), C.c_int(nq), _p(q),
                          C.c_float(thr), _p(idx), _p(dist))
    return idx, dist


def rot_quaternion(lrf9):
    l = _f(lrf9).reshape(9)
    out = np.empty(4, np.float32)
    lib().ismref_rot_quaternion(_p(l), _p(out))
    return out


def rotate_into(lrf9, v):
    l, v = _f(lrf9).reshape(9), _f(v)
    out = np.empty(3, np.float32)
    lib().ismref_rotate_into(_p(l), _p(v), _p(out))
    return out


def rotate_back(lrf9, v):
    l, v = _f(lrf9).reshape(9), _f(v)
    out = np.empty(3, np.float32)
    lib().ismref_rotate_back(_p(l), _p(v), _p(out))
    return out


def cast_votes(cb, weight_flags, lrf, kx, ky, kz, idx, dist):
    """cb: dict with words, vote_offsets, vote_xyz, vote_class, vote_instance, class_sigma (+ optional weights/bbox)"""
    words = _f(cb["words"])
    vo = _u(cb["vote_offsets"])
    idx, dist = _i(idx), _f(dist)
    nq, k = idx.shape
    maxv = int(np.max(np.diff(vo))) if len(vo) > 1 else 0
    ns = nq * k * maxv
    pos = np.empty((ns, 3), np.float32); w = np.empty(ns, np.float32)
    cls = np.empty(ns, np.int32); inst = np.empty(ns, np.int32); cw = np.empty(ns, np.int32)
    bq = np.empty((ns, 4), np.float32); bs = np.empty((ns, 3), np.float32)
    sig = _f(cb["class_sigma"])
    args = [_f(cb.get("word_weight")), vo, _f(cb["vote_xyz"]), _f(cb.get("vote_weight")), _f(cb.get("vote_class_weight")),
            _u(cb["vote_class"]), _u(cb["vote_instance"]), _f(cb.get("vote_bbox_quat")), _f(cb.get("vote_bbox_size"))]
    lrf, kx, ky, kz = map(_f, (lrf, kx, ky, kz))
    lib().ismref_cast_votes(C.c_int(words.shape[0]), C.c_int(words.shape[1]), *[_p(a) for a in args], C.c_int(len(sig)), _p(sig),
                            C.c_uint32(weight_flags), C.c_int(nq), _p(lrf), _p(kx), _p(ky), _p(kz), C.c_int(k), _p(idx), _p(dist),
                            _p(pos), _p(w), _p(cls), _p(inst), _p(cw), _p(bq), _p(bs))
    return dict(pos=pos, weight=w, cls=cls, inst=inst, codeword=cw, bbox_quat=bq, bbox_size=bs)


def find_maxima(slot_offsets, votes, n_classes, bandwidth, threshold=1e-3, max_iter=1000, kernel=0, suppression=0,
                min_votes_threshold=1, min_threshold=0.0, best_k=-1, max_maxima=16, class_bandwidth=None, max_filter=0,
                average_rotation=False, single_object_max_type=0, object_centroid=None, object_radius=None):
    so = _u(slot_offsets)
    n_obj = len(so) - 1
    cbw = _f(class_bandwidth)
    bq = _f(votes["bbox_quat"]) if average_rotation else None
    bq_out = np.empty((n_obj, max_maxima, 4), np.float32) if average_rotation else None
    oc, orad = _f(object_centroid), _f(object_radius)
    P = MaximaParams(n_classes, cbw.ctypes.data if cbw is not None else None, bandwidth, threshold, max_iter, kernel, suppression,
                     min_votes_threshold, min_threshold, best_k, max_maxima, max_filter,
                     bq.ctypes.data if bq is not None else None, bq_out.ctypes.data if bq_out is not None else None, single_object_max_type,
                     oc.ctypes.data if oc is not None else None, orad.ctypes.data if orad is not None else None)
    out = dict(n=np.empty(n_obj, np.int32), pos=np.empty((n_obj, max_maxima, 3), np.float32),
               weight=np.empty((n_obj, max_maxima), np.float32), cls=np.empty((n_obj, max_maxima), np.int32),
               inst=np.empty((n_obj, max_maxima), np.int32), inst_weight=np.empty((n_obj, max_maxima), np.float32),
               bbox_size=np.empty((n_obj, max_maxima, 3), np.float32), n_votes=np.empty((n_obj, max_maxima), np.int32),
               class_score=np.empty((n_obj, n_classes), np.float32))
    pos, w, cls, inst = _f(votes["pos"]), _f(votes["weight"]), _i(votes["cls"]), _i(votes["inst"])
    bs = _f(votes.get("bbox_size"))
    lib().ismref_find_maxima(C.c_int(n_obj), _p(so), _p(pos), _p(w), _p(cls), _p(inst), _p(bs), C.byref(P), _p(out["n"]), _p(out["pos"]),
                             _p(out["weight"]), _p(out["cls"]), _p(out["inst"]), _p(out["inst_weight"]), _p(out["bbox_size"]),
                             _p(out["n_votes"]), _p(out["class_score"]))
    if bq_out is not None:
        out["bbox_quat"] = bq_out
    return out


class HoughParams(C.Structure):
    _fields_ = [("n_classes", C.c_int), ("min_coord", C.c_float * 3), ("max_coord", C.c_float * 3), ("bin_size", C.c_float),
                ("class_bin", C.c_void_p), ("use_interpolation", C.c_int), ("rel_threshold", C.c_float),
                ("min_votes_threshold", C.c_int), ("min_threshold", C.c_float), ("best_k", C.c_int), ("max_maxima", C.c_int), ("max_filter", C.c_int),
                ("vote_bbox_quat", C.c_void_p), ("max_bbox_quat_out", C.c_void_p)]


def hough3d_maxima(slot_offsets, votes, n_classes, bin_size, min_coord=(-5, -5, -5), max_coord=(5, 5, 5), use_interpolation=True,
                   rel_threshold=0.8, min_votes_threshold=1, min_threshold=0.0, best_k=-1, max_maxima=16, class_bin=None, max_filter=0,
                   average_rotation=False):
    so = _u(slot_offsets)
    n_obj = len(so) - 1
    cb = _f(class_bin)
    bq = _f(votes["bbox_quat"]) if average_rotation else None
    bq_out = np.empty((n_obj, max_maxima, 4), np.float32) if average_rotation else None
    P = HoughParams(n_classes, (C.c_float * 3)(*min_coord), (C.c_float * 3)(*max_coord), bin_size, cb.ctypes.data if cb is not None else None,
                    1 if use_interpolation else 0, rel_threshold, min_votes_threshold, min_threshold, best_k, max_maxima, max_filter,
                    bq.ctypes.data if bq is not None else None, bq_out.ctypes.data if bq_out is not None else None)
    out = dict(n=np.empty(n_obj, np.int32), pos=np.empty((n_obj, max_maxima, 3), np.float32),
               weight=np.empty((n_obj, max_maxima), np.float32), cls=np.empty((n_obj, max_maxima), np.int32),
               inst=np.empty((n_obj, max_maxima), np.int32), inst_weight=np.empty((n_obj, max_maxima), np.float32),
               bbox_size=np.empty((n_obj, max_maxima, 3), np.float32), n_votes=np.empty((n_obj, max_maxima), np.int32),
               class_score=np.empty((n_obj, n_classes), np.float32))
    pos, w, cls, inst = _f(votes["pos"]), _f(votes["weight"]), _i(votes["cls"]), _i(votes["inst"])
    bs = _f(votes.get("bbox_size"))
    lib().ismref_hough3d_maxima(C.c_int(n_obj), _p(so), _p(pos), _p(w), _p(cls), _p(inst), _p(bs), C.byref(P), _p(out["n"]), _p(out["pos"]),
                                _p(out["weight"]), _p(out["cls"]), _p(out["inst"]), _p(out["inst_weight"]), _p(out["bbox_size"]),
                                _p(out["n_votes"]), _p(out["class_score"]))
    if bq_out is not None:
        out["bbox_quat"] = bq_out
    return out


def create_seeds(pos, w, bin_size):
    pos, w = _f(pos), _f(w)
    n = len(w)
    sp = np.empty((max(n, 1), 3), np.float32); sw = np.empty(max(n, 1), np.float32)
    m = lib().ismref_create_seeds(C.c_int(n), _p(pos), _p(w), C.c_float(bin_size), C.c_int(n), _p(sp), _p(sw))
    return sp[:m].copy(), sw[:m].copy()


def voxel_grid(x, y, z, leaf, rgba=None):
    x, y, z = map(_f, (x, y, z))
    rgba = _u(rgba)
    n = len(x)
    kx = np.empty(n, np.float32); ky = np.empty(n, np.float32); kz = np.empty(n, np.float32); kc = np.empty(n, np.uint32)
    m = lib().ismref_voxel_grid(C.c_int(n), _p(x), _p(y), _p(z), _p(rgba), C.c_float(leaf), C.c_int(n), _p(kx), _p(ky), _p(kz), _p(kc))
    return kx[:m].copy(), ky[:m].copy(), kz[:m].copy(), kc[:m].copy()


def class_sigmas(metric, feats, feat_class, feat_model, activated_word, words, n_classes):
    feats, words = _f(feats), _f(words)
    out = np.empty(n_classes, np.float32)
    lib().ismref_class_sigmas(C.c_int(metric), C.c_int(feats.shape[1]), C.c_int(feats.shape[0]), _p(feats), _p(_u(feat_class)),
                              _p(_u(feat_model)), _p(_i(activated_word)), C.c_int(words.shape[0]), _p(words), C.c_int(n_classes), _p(out))
    return out


def activate(metric, feats, lrf, kp, feat_class, feat_model, feat_center, k=1, clean_up=True, n_classes=None, codewords=None):
    """Codebook::activate (codewords None: one codeword per training feature) -> dict(word_src, vote_offsets, vote_feature, vote_xyz,
    vote_weight, vote_class_weight, class_sigma)"""
    feats, lrf, kp, feat_center = _f(feats), _f(lrf), _f(kp), _f(feat_center)
    n, dim = feats.shape
    fc, fm = _u(feat_class), _u(feat_model)
    C_ = int(n_classes if n_classes is not None else fc.max() + 1)
    kx, ky, kz = (np.ascontiguousarray(kp[:, i]) for i in range(3))
    cw = None if codewords is None else _f(codewords)
    ncw = n if cw is None else len(cw)
    nw = C.c_int32(0)
    word_src = np.empty(ncw, np.uint32); vo = np.empty(ncw + 1, np.uint32); vf = np.empty(n * k, np.uint32)
    vxyz = np.empty((n * k, 3), np.float32); vw = np.empty(n * k, np.float32); vcw = np.empty(n * k, np.float32); sig = np.empty(C_, np.float32)
    rc = lib().ismref_activate(C.c_int(metric), C.c_int(dim), C.c_int(n), _p(feats), _p(lrf), _p(kx), _p(ky), _p(kz), _p(fc), _p(fm), _p(feat_center),
                               C.c_int(ncw), _p(cw) if cw is not None else C.c_void_p(0),
                               C.c_int(k), C.c_int(1 if clean_up else 0), C.c_int(C_), C.byref(nw), _p(word_src), _p(vo), _p(vf), _p(vxyz), _p(vw), _p(vcw), _p(sig))
    assert rc == 0
    m = nw.value; nv = int(vo[m])
    return dict(word_src=word_src[:m].copy(), vote_offsets=vo[:m + 1].copy(), vote_feature=vf[:nv].copy(), vote_xyz=vxyz[:nv].copy(),
                vote_weight=vw[:nv].copy(), vote_class_weight=vcw[:nv].copy(), class_sigma=sig)


def kmeans(metric, feats, n_clusters, max_iterations=1000, centers_init=2, seed=0):
    """ClusteringKMeans::cluster -> (centers [m, dim], assign [n], dist [n], iterations)"""
    feats = _f(feats)
    n, dim = feats.shape
    kc = min(int(n_clusters), n)
    centers = np.empty((kc, dim), np.float32); assign = np.empty(n, np.int32); dist = np.empty(n, np.float32)
    m = C.c_int32(0); it = C.c_int32(0)
    rc = lib().ismref_kmeans(C.c_int(metric), C.c_int(n), C.c_int(dim), _p(feats), C.c_int(kc), C.c_int(max_iterations), C.c_int(centers_init),
                             C.c_uint64(seed), _p(centers), _p(assign), _p(dist), C.byref(m), C.byref(it))
    assert rc == 0
    return centers[:m.value].copy(), assign, dist, it.value


def pca_normals(pt_off, x, y, z, radius, orientation):
    po = _u(pt_off); x, y, z = map(_f, (x, y, z))
    n = len(x)
    nx = np.empty(n, np.float32); ny = np.empty(n, np.float32); nz = np.empty(n, np.float32)
    lib().ismref_pca_normals(C.c_int(len(po) - 1), _p(po), _p(x), _p(y), _p(z), C.c_float(radius), C.c_int(orientation), _p(nx), _p(ny), _p(nz))
    return np.stack([nx, ny, nz], 1)
