"""ctypes binding of libismhip.so (include/ismhip.h) for the Python test / bench harness.

The product is the shared library; this module only marshals pointers. torch is used for device memory
(tensor.data_ptr()) and streams; no torch type crosses the C ABI. There is no CPU fallback: if the
library or a gfx950 device is missing, loading / ctx creation raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libismhip.so")

METRIC_L2SQ, METRIC_CHI2 = 0, 1
W_CLASS, W_VOTE, W_MATCHING, W_CODEWORD = 1, 2, 4, 8
KERNEL_GAUSSIAN, KERNEL_UNIFORM = 0, 1
SUPPRESS_AVERAGE, SUPPRESS_SUPPRESS, SUPPRESS_NONE = 0, 1, 2
MAXFILTER_NONE, MAXFILTER_SIMPLE, MAXFILTER_MERGE = 0, 1, 2
SOM_MEANSHIFT, SOM_BANDWIDTH, SOM_MODEL_RADIUS, SOM_COMPLETE_VOTING_SPACE = 0, 1, 2, 3
ERR_NODEVICE = -5

EXPORTS = [
    "ismhip_abi_version", "ismhip_ctx_create", "ismhip_ctx_create_on_stream", "ismhip_ctx_destroy", "ismhip_sync", "ismhip_last_error",
    "ismhip_timers_enable", "ismhip_timers_reset", "ismhip_timer_get",
    "ismhip_cloud_create", "ismhip_cloud_destroy", "ismhip_cloud_centroids", "ismhip_cloud_radii", "ismhip_estimate_normals", "ismhip_estimate_normals_pca",
    "ismhip_shot_lrf", "ismhip_shot352", "ismhip_cshot1344", "ismhip_fpfh33", "ismhip_center_dist",
    "ismhip_compact_features", "ismhip_compact_descriptor_rows", "ismhip_filter_normals", "ismhip_voxel_keypoints", "ismhip_gather_columns",
    "ismhip_codebook_create", "ismhip_codebook_set_word_class", "ismhip_codebook_destroy", "ismhip_codebook_max_votes_per_word", "ismhip_codebook_stage1_dims", "ismhip_codebook_stage2_dims",
    "ismhip_knn", "ismhip_knn_ratio", "ismhip_knn_rule", "ismhip_cast_votes", "ismhip_find_maxima", "ismhip_hough3d_maxima", "ismhip_train_activate", "ismhip_kmeans",
]


class MaximaParams(C.Structure):
    _fields_ = [("n_classes", C.c_int), ("class_bandwidth_h", C.c_void_p), ("bandwidth", C.c_float),
                ("threshold", C.c_float), ("max_iter", C.c_int), ("kernel", C.c_int), ("suppression", C.c_int),
                ("min_votes_threshold", C.c_int), ("min_threshold", C.c_float), ("best_k", C.c_int),
                ("max_maxima", C.c_int), ("max_filter", C.c_int),
                ("vote_bbox_quat", C.c_void_p), ("max_bbox_quat_out", C.c_void_p), ("single_object_max_type", C.c_int),
                ("object_centroid", C.c_void_p), ("object_radius", C.c_void_p)]


class HoughParams(C.Structure):
    _fields_ = [("n_classes", C.c_int), ("min_coord", C.c_float * 3), ("max_coord", C.c_float * 3), ("bin_size", C.c_float),
                ("class_bin_h", C.c_void_p), ("use_interpolation", C.c_int), ("rel_threshold", C.c_float),
                ("min_votes_threshold", C.c_int), ("min_threshold", C.c_float), ("best_k", C.c_int), ("max_maxima", C.c_int), ("max_filter", C.c_int),
                ("vote_bbox_quat", C.c_void_p), ("max_bbox_quat_out", C.c_void_p)]


class IsmHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libismhip.so (once). Raises if it has not been built — the hot path has no other back end."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IsmHipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C point-cloud-donkey_amd/csrc)")
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's): import it FIRST so that the process holds exactly
        # one HIP runtime and libismhip.so binds to the one that owns torch's device memory and streams.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.ismhip_last_error.restype = C.c_char_p
        L.ismhip_last_error.argtypes = [C.c_void_p]
        for name in EXPORTS:
            fn = getattr(L, name)
            if name != "ismhip_last_error":
                fn.restype = C.c_int
        _lib = L
    return _lib


def _p(t):
    """device pointer of a torch tensor / host pointer of a numpy array / None"""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


class Ctx:
    def __init__(self, device=0, stream="torch"):
        """stream: "torch" = torch's current stream on the device (library work is then ordered with torch ops and
        tensor.cpu() waits for it), None = a private non-blocking stream, or a raw hipStream_t value."""
        self._h = C.c_void_p()
        L = lib()
        if stream is None:
            rc = L.ismhip_ctx_create(C.c_int(device), C.c_void_p(0), C.byref(self._h))
        else:
            if stream == "torch":
                import torch
                if not torch.cuda.is_available():
                    raise IsmHipError("no gfx950 device visible: the hot path has no CPU fallback (ISMHIP_ERR_NODEVICE)")
                if not (0 <= device < torch.cuda.device_count()):
                    raise IsmHipError(f"device {device} out of range (ISMHIP_ERR_INVALID)")
                stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.ismhip_ctx_create_on_stream(C.c_int(device), C.c_void_p(stream), C.byref(self._h))
        if rc != 0:
            raise IsmHipError(f"ismhip_ctx_create failed ({rc}); a gfx950 device is required, there is no CPU fallback")
        self.device = device

    def check(self, rc, what):
        if rc != 0:
            raise IsmHipError(f"{what} failed ({rc}): {lib().ismhip_last_error(self._h).decode()}")

    def sync(self):
        self.check(lib().ismhip_sync(self._h), "ismhip_sync")

    def timers_enable(self, on=True):
        self.check(lib().ismhip_timers_enable(self._h, C.c_int(1 if on else 0)), "timers_enable")

    def timers_reset(self):
        self.check(lib().ismhip_timers_reset(self._h), "timers_reset")

    def timer(self, name):
        ms, n = C.c_double(), C.c_int64()
        self.check(lib().ismhip_timer_get(self._h, name.encode(), C.byref(ms), C.byref(n)), "timer_get")
        return ms.value, n.value

    def close(self):
        if self._h:
            lib().ismhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cloud:
    """ismhip_cloud: search surface of a batch of objects (device SoA tensors are borrowed, keep them alive)."""

    def __init__(self, ctx, pt_offsets, x, y, z, nx, ny, nz, cell_size, rgba=None):
        self.ctx = ctx
        self.pt_offsets = _u32(pt_offsets)
        self.n_obj = len(self.pt_offsets) - 1
        self._keep = (x, y, z, nx, ny, nz, rgba)
        self._h = C.c_void_p()
        rc = lib().ismhip_cloud_create(ctx._h, C.c_int(self.n_obj), _p(self.pt_offsets), _p(x), _p(y), _p(z), _p(nx), _p(ny),
                                       _p(nz), _p(rgba), C.c_float(cell_size), C.byref(self._h))
        ctx.check(rc, "ismhip_cloud_create")

    def close(self):
        if self._h:
            lib().ismhip_cloud_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Codebook:
    def __init__(self, ctx, words, vote_offsets, vote_xyz, vote_class, vote_instance, n_classes, class_sigma,
                 word_weight=None, vote_weight=None, vote_class_weight=None, vote_bbox_quat=None, vote_bbox_size=None):
        f32 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        self.ctx = ctx
        words = f32(words)
        self.n_words, self.dim = words.shape
        self.n_classes = int(n_classes)
        self._h = C.c_void_p()
        rc = lib().ismhip_codebook_create(ctx._h, C.c_int(self.n_words), C.c_int(self.dim), _p(words), _p(f32(word_weight)),
                                          _p(_u32(vote_offsets)), _p(f32(vote_xyz)), _p(f32(vote_weight)),
                                          _p(f32(vote_class_weight)), _p(_u32(vote_class)), _p(_u32(vote_instance)),
                                          _p(f32(vote_bbox_quat)), _p(f32(vote_bbox_size)), C.c_int(self.n_classes),
                                          _p(f32(class_sigma)), C.byref(self._h))
        ctx.check(rc, "ismhip_codebook_create")
        self.max_votes = lib().ismhip_codebook_max_votes_per_word(self._h)
        e = C.c_float(1.0)
        self.stage1_dims = int(lib().ismhip_codebook_stage1_dims(self._h, C.byref(e)))     # 0: squared-L2 candidates on all dimensions
        self.stage1_energy = float(e.value)
        self.stage2_dims = int(lib().ismhip_codebook_stage2_dims(self._h, C.byref(e)))     # 0: stage 2 of the squared-L2 search on all dimensions
        self.stage2_energy = float(e.value)

    def set_word_class(self, word_class):
        self.ctx.check(lib().ismhip_codebook_set_word_class(self.ctx._h, self._h, _p(_u32(word_class))), "ismhip_codebook_set_word_class")

    def close(self):
        if self._h:
            lib().ismhip_codebook_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch():
    import torch
    return torch


def shot_lrf(ctx, cloud, kp_offsets, kpx, kpy, kpz, radius):
    torch = _torch()
    ko = _u32(kp_offsets)
    out = torch.empty((int(ko[-1]), 9), dtype=torch.float32, device=kpx.device)
    ctx.check(lib().ismhip_shot_lrf(ctx._h, cloud._h, _p(ko), _p(kpx), _p(kpy), _p(kpz), C.c_float(radius), _p(out)), "ismhip_shot_lrf")
    return out


def shot352(ctx, cloud, kp_offsets, kpx, kpy, kpz, lrf, radius, want_counts=False):
    torch = _torch()
    ko = _u32(kp_offsets)
    n = int(ko[-1])
    out = torch.empty((n, 352), dtype=torch.float32, device=kpx.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=kpx.device) if want_counts else None
    ctx.check(lib().ismhip_shot352(ctx._h, cloud._h, _p(ko), _p(kpx), _p(kpy), _p(kpz), _p(lrf), C.c_float(radius), _p(out), _p(cnt)),
              "ismhip_shot352")
    return (out, cnt) if want_counts else out


def cshot1344(ctx, cloud, kp_offsets, kpx, kpy, kpz, kp_rgba, lrf, radius, want_counts=False):
    torch = _torch()
    ko = _u32(kp_offsets)
    n = int(ko[-1])
    out = torch.empty((n, 1344), dtype=torch.float32, device=kpx.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=kpx.device) if want_counts else None
    ctx.check(lib().ismhip_cshot1344(ctx._h, cloud._h, _p(ko), _p(kpx), _p(kpy), _p(kpz), _p(kp_rgba), _p(lrf), C.c_float(radius),
                                     _p(out), _p(cnt)), "ismhip_cshot1344")
    return (out, cnt) if want_counts else out


def fpfh33(ctx, cloud, kp_offsets, kpx, kpy, kpz, radius, want_counts=False):
    torch = _torch()
    ko = _u32(kp_offsets)
    n = int(ko[-1])
    out = torch.empty((n, 33), dtype=torch.float32, device=kpx.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=kpx.device) if want_counts else None
    ctx.check(lib().ismhip_fpfh33(ctx._h, cloud._h, _p(ko), _p(kpx), _p(kpy), _p(kpz), C.c_float(radius), _p(out), _p(cnt)),
              "ismhip_fpfh33")
    return (out, cnt) if want_counts else out


def cloud_centroids(ctx, cloud, device):
    torch = _torch()
    out = torch.empty((cloud.n_obj, 3), dtype=torch.float32, device=device)
    ctx.check(lib().ismhip_cloud_centroids(ctx._h, cloud._h, _p(out)), "ismhip_cloud_centroids")
    return out


def cloud_radii(ctx, cloud, centroid):
    """SingleObjectHelper::getModelRadius per object: farthest point from centroid [n_obj, 3] (device)"""
    torch = _torch()
    out = torch.empty((cloud.n_obj,), dtype=torch.float32, device=centroid.device)
    ctx.check(lib().ismhip_cloud_radii(ctx._h, cloud._h, _p(centroid), _p(out)), "ismhip_cloud_radii")
    return out


def center_dist(ctx, cloud, kp_offsets, kpx, kpy, kpz):
    torch = _torch()
    ko = _u32(kp_offsets)
    out = torch.empty((int(ko[-1]),), dtype=torch.float32, device=kpx.device)
    ctx.check(lib().ismhip_center_dist(ctx._h, cloud._h, _p(ko), _p(kpx), _p(kpy), _p(kpz), _p(out)), "ismhip_center_dist")
    return out


def estimate_normals(ctx, cloud, radius, nx, ny, nz):
    """ImplicitShapeModel::computeNormals (method 2): fills nx, ny, nz (device tensors, original order) and the cloud's own copies"""
    ctx.check(lib().ismhip_estimate_normals(ctx._h, cloud._h, C.c_float(radius), _p(nx), _p(ny), _p(nz)), "ismhip_estimate_normals")
    cloud._keep = getattr(cloud, "_keep", ()) + (nx, ny, nz)
    return nx, ny, nz


def estimate_normals_pca(ctx, cloud, radius, orientation, nx, ny, nz):
    """ConsistentNormalsMethod 0 (orientation 0: towards the origin) / 1 (orientation 1: away from the object's centroid)"""
    ctx.check(lib().ismhip_estimate_normals_pca(ctx._h, cloud._h, C.c_float(radius), C.c_int(orientation), _p(nx), _p(ny), _p(nz)), "ismhip_estimate_normals_pca")
    cloud._keep = getattr(cloud, "_keep", ()) + (nx, ny, nz)
    return nx, ny, nz


class _PointArrays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "y", "z", "nx", "ny", "nz", "rgba")]


def filter_normals(ctx, pt_offsets, x, y, z, nx, ny, nz, rgba=None):
    """ImplicitShapeModel::filterNormals on the device: returns (new_offsets numpy, x, y, z, nx, ny, nz, rgba or None) without the
    points whose normal holds a NaN (order kept)"""
    torch = _torch()
    po = _u32(pt_offsets)
    n_obj = len(po) - 1
    ins = [x, y, z, nx, ny, nz] + ([rgba] if rgba is not None else [])
    outs = [torch.empty_like(t) for t in ins]
    a_in = _PointArrays(*[t.data_ptr() for t in ins], *([None] if rgba is None else []))
    a_out = _PointArrays(*[t.data_ptr() for t in outs], *([None] if rgba is None else []))
    new = np.zeros(n_obj + 1, dtype=np.uint32)
    ctx.check(lib().ismhip_filter_normals(ctx._h, C.c_int(n_obj), _p(po), C.byref(a_in), C.byref(a_out), _p(new)), "ismhip_filter_normals")
    m = int(new[-1])
    outs = [t[:m] for t in outs]
    return (new, *outs, *([None] if rgba is None else []))


def voxel_keypoints(ctx, pt_offsets, x, y, z, leaf, rgba=None):
    """KeypointsVoxelGrid on the device: returns (kp_offsets[n_obj+1] numpy, kx, ky, kz, krgba or None) packed object after object"""
    torch = _torch()
    po = _u32(pt_offsets)
    n_obj = len(po) - 1
    n = int(po[-1])
    kx = torch.empty((n,), dtype=torch.float32, device=x.device); ky = torch.empty_like(kx); kz = torch.empty_like(kx)
    kc = torch.empty((n,), dtype=torch.int32, device=x.device) if rgba is not None else None
    ko = np.zeros(n_obj + 1, dtype=np.uint32)
    ctx.check(lib().ismhip_voxel_keypoints(ctx._h, C.c_int(n_obj), _p(po), _p(x), _p(y), _p(z), _p(rgba), C.c_float(leaf), C.c_uint32(n),
                                           _p(kx), _p(ky), _p(kz), _p(kc), _p(ko)), "ismhip_voxel_keypoints")
    m = int(ko[-1])
    return ko, kx[:m], ky[:m], kz[:m], (kc[:m] if kc is not None else None)


def compact_descriptor_rows(ctx, kp_offsets, desc, lrf, kpx, kpy, kpz):
    """compact_features for descriptor matrices of this library (rows are NaN as a whole): when nothing is dropped the INPUT tensors are
    returned, no copy is made"""
    torch = _torch()
    ko = _u32(kp_offsets)
    n_obj = len(ko) - 1
    n, dim = desc.shape
    d_o = torch.empty_like(desc)
    l_o = torch.empty_like(lrf) if lrf is not None else None
    x_o, y_o, z_o = torch.empty_like(kpx), torch.empty_like(kpy), torch.empty_like(kpz)
    src = torch.empty((n,), dtype=torch.int32, device=desc.device)
    keep = np.zeros(n_obj + 1, dtype=np.uint32)
    all_kept = C.c_int(0)
    ctx.check(lib().ismhip_compact_descriptor_rows(ctx._h, C.c_int(n_obj), _p(ko), C.c_int(dim), _p(desc), _p(lrf), _p(kpx), _p(kpy), _p(kpz),
                                                   _p(d_o), _p(l_o), _p(x_o), _p(y_o), _p(z_o), _p(src), _p(keep), C.byref(all_kept)),
              "ismhip_compact_descriptor_rows")
    if all_kept.value:
        return keep, desc, lrf, kpx, kpy, kpz, src
    m = int(keep[-1])
    return keep, d_o[:m], (l_o[:m] if l_o is not None else None), x_o[:m], y_o[:m], z_o[:m], src[:m]


def compact_features(ctx, kp_offsets, desc, lrf, kpx, kpy, kpz):
    """returns (keep_offsets, desc, lrf, kpx, kpy, kpz, src_index) with NaN rows removed (order preserved)"""
    torch = _torch()
    ko = _u32(kp_offsets)
    n_obj = len(ko) - 1
    n, dim = desc.shape
    d_o = torch.empty_like(desc)
    l_o = torch.empty_like(lrf) if lrf is not None else None
    x_o, y_o, z_o = torch.empty_like(kpx), torch.empty_like(kpy), torch.empty_like(kpz)
    src = torch.empty((n,), dtype=torch.int32, device=desc.device)
    keep = np.zeros(n_obj + 1, dtype=np.uint32)
    ctx.check(lib().ismhip_compact_features(ctx._h, C.c_int(n_obj), _p(ko), C.c_int(dim), _p(desc), _p(lrf), _p(kpx), _p(kpy), _p(kpz),
                                            _p(d_o), _p(l_o), _p(x_o), _p(y_o), _p(z_o), _p(src), _p(keep)), "ismhip_compact_features")
    m = int(keep[-1])
    return keep, d_o[:m], (l_o[:m] if l_o is not None else None), x_o[:m], y_o[:m], z_o[:m], src[:m]


def knn(ctx, cb, metric, q, k=1):
    torch = _torch()
    nq = q.shape[0]
    idx = torch.empty((nq, k), dtype=torch.int32, device=q.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    ctx.check(lib().ismhip_knn(ctx._h, cb._h, C.c_int(metric), C.c_int(nq), _p(q), C.c_int(k), _p(idx), _p(dist)), "ismhip_knn")
    return idx, dist


def knn_ratio(ctx, cb, metric, q, ratio_threshold):
    torch = _torch()
    nq = q.shape[0]
    idx = torch.empty((nq, 1), dtype=torch.int32, device=q.device)
    dist = torch.empty((nq, 1), dtype=torch.float32, device=q.device)
    ctx.check(lib().ismhip_knn_ratio(ctx._h, cb._h, C.c_int(metric), C.c_int(nq), _p(q), C.c_float(ratio_threshold), _p(idx), _p(dist)),
              "ismhip_knn_ratio")
    return idx, dist


def knn_rule(ctx, cb, metric, q, ratio_threshold):
    torch = _torch()
    nq = q.shape[0]
    idx = torch.empty((nq, 1), dtype=torch.int32, device=q.device)
    dist = torch.empty((nq, 1), dtype=torch.float32, device=q.device)
    ctx.check(lib().ismhip_knn_rule(ctx._h, cb._h, C.c_int(metric), C.c_int(nq), _p(q), C.c_float(ratio_threshold), _p(idx), _p(dist)),
              "ismhip_knn_rule")
    return idx, dist


def cast_votes(ctx, cb, weight_flags, lrf, kpx, kpy, kpz, idx, dist, want_bbox=False):
    torch = _torch()
    nq, k = idx.shape
    ns = nq * k * max(cb.max_votes, 0)
    dev = idx.device
    pos = torch.empty((ns, 3), dtype=torch.float32, device=dev)
    w = torch.empty((ns,), dtype=torch.float32, device=dev)
    cls = torch.empty((ns,), dtype=torch.int32, device=dev)
    inst = torch.empty((ns,), dtype=torch.int32, device=dev)
    cw = torch.empty((ns,), dtype=torch.int32, device=dev)
    bq = torch.empty((ns, 4), dtype=torch.float32, device=dev) if want_bbox else None
    bs = torch.empty((ns, 3), dtype=torch.float32, device=dev) if want_bbox else None
    ctx.check(lib().ismhip_cast_votes(ctx._h, cb._h, C.c_uint32(weight_flags), C.c_int(nq), _p(lrf), _p(kpx), _p(kpy), _p(kpz), C.c_int(k),
                                      _p(idx), _p(dist), _p(pos), _p(w), _p(cls), _p(inst), _p(cw), _p(bq), _p(bs)), "ismhip_cast_votes")
    return dict(pos=pos, weight=w, cls=cls, inst=inst, codeword=cw, bbox_quat=bq, bbox_size=bs)


def find_maxima(ctx, slot_offsets, votes, n_classes, bandwidth, threshold=1e-3, max_iter=1000, kernel=KERNEL_GAUSSIAN,
                suppression=SUPPRESS_AVERAGE, min_votes_threshold=1, min_threshold=0.0, best_k=-1, max_maxima=16,
                class_bandwidth=None, max_filter=0, average_rotation=False, single_object_max_type=SOM_MEANSHIFT,
                object_centroid=None, object_radius=None):
    """average_rotation: votes["bbox_quat"] in, out["bbox_quat"] per maximum; single_object_max_type != SOM_MEANSHIFT needs
    object_centroid [n_obj,3] (and object_radius [n_obj] for SOM_MODEL_RADIUS) as device tensors"""
    torch = _torch()
    so = _u32(slot_offsets)
    n_obj = len(so) - 1
    dev = votes["pos"].device
    cbw = None if class_bandwidth is None else np.ascontiguousarray(np.asarray(class_bandwidth, dtype=np.float32))
    bq_out = torch.empty((n_obj, max_maxima, 4), dtype=torch.float32, device=dev) if average_rotation else None
    P = MaximaParams(n_classes, cbw.ctypes.data if cbw is not None else None, bandwidth, threshold, max_iter, kernel, suppression,
                     min_votes_threshold, min_threshold, best_k, max_maxima, max_filter,
                     _p(votes["bbox_quat"]).value if average_rotation else None, _p(bq_out).value, single_object_max_type,
                     _p(object_centroid).value, _p(object_radius).value)
    out = dict(
        n=torch.empty((n_obj,), dtype=torch.int32, device=dev),
        pos=torch.empty((n_obj, max_maxima, 3), dtype=torch.float32, device=dev),
        weight=torch.empty((n_obj, max_maxima), dtype=torch.float32, device=dev),
        cls=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        inst=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        inst_weight=torch.empty((n_obj, max_maxima), dtype=torch.float32, device=dev),
        bbox_size=torch.empty((n_obj, max_maxima, 3), dtype=torch.float32, device=dev),
        n_votes=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        class_score=torch.empty((n_obj, n_classes), dtype=torch.float32, device=dev),
    )
    ctx.check(lib().ismhip_find_maxima(ctx._h, C.c_int(n_obj), _p(so), _p(votes["pos"]), _p(votes["weight"]), _p(votes["cls"]),
                                       _p(votes["inst"]), _p(votes.get("bbox_size")), C.byref(P), _p(out["n"]), _p(out["pos"]),
                                       _p(out["weight"]), _p(out["cls"]), _p(out["inst"]), _p(out["inst_weight"]), _p(out["bbox_size"]),
                                       _p(out["n_votes"]), _p(out["class_score"])), "ismhip_find_maxima")
    if bq_out is not None:
        out["bbox_quat"] = bq_out
    return out


def hough3d_maxima(ctx, slot_offsets, votes, n_classes, bin_size, min_coord=(-5, -5, -5), max_coord=(5, 5, 5), use_interpolation=True,
                   rel_threshold=0.8, min_votes_threshold=1, min_threshold=0.0, best_k=-1, max_maxima=16, class_bin=None, max_filter=0,
                   average_rotation=False):
    """VotingHough3D on the device: same outputs as find_maxima"""
    torch = _torch()
    so = _u32(slot_offsets)
    n_obj = len(so) - 1
    dev = votes["pos"].device
    cb = None if class_bin is None else np.ascontiguousarray(np.asarray(class_bin, dtype=np.float32))
    bq_out = torch.empty((n_obj, max_maxima, 4), dtype=torch.float32, device=dev) if average_rotation else None
    P = HoughParams(n_classes, (C.c_float * 3)(*min_coord), (C.c_float * 3)(*max_coord), bin_size, cb.ctypes.data if cb is not None else None,
                    1 if use_interpolation else 0, rel_threshold, min_votes_threshold, min_threshold, best_k, max_maxima, max_filter,
                    _p(votes["bbox_quat"]).value if average_rotation else None, _p(bq_out).value)
    out = dict(
        n=torch.empty((n_obj,), dtype=torch.int32, device=dev),
        pos=torch.empty((n_obj, max_maxima, 3), dtype=torch.float32, device=dev),
        weight=torch.empty((n_obj, max_maxima), dtype=torch.float32, device=dev),
        cls=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        inst=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        inst_weight=torch.empty((n_obj, max_maxima), dtype=torch.float32, device=dev),
        bbox_size=torch.empty((n_obj, max_maxima, 3), dtype=torch.float32, device=dev),
        n_votes=torch.empty((n_obj, max_maxima), dtype=torch.int32, device=dev),
        class_score=torch.empty((n_obj, n_classes), dtype=torch.float32, device=dev),
    )
    ctx.check(lib().ismhip_hough3d_maxima(ctx._h, C.c_int(n_obj), _p(so), _p(votes["pos"]), _p(votes["weight"]), _p(votes["cls"]),
                                          _p(votes["inst"]), _p(votes.get("bbox_size")), C.byref(P), _p(out["n"]), _p(out["pos"]),
                                          _p(out["weight"]), _p(out["cls"]), _p(out["inst"]), _p(out["inst_weight"]), _p(out["bbox_size"]),
                                          _p(out["n_votes"]), _p(out["class_score"])), "ismhip_hough3d_maxima")
    if bq_out is not None:
        out["bbox_quat"] = bq_out
    return out


def train_activate(ctx, metric, desc, lrf, kx, ky, kz, feat_class, feat_model, feat_center, k=1, clean_up=True, n_classes=None, codewords=None):
    """Codebook::activate on the device (features class-major; codewords = device matrix of cluster centres, None = the features
    themselves) -> dict of host arrays (word_src, vote_offsets, vote_feature, vote_xyz, vote_weight, vote_class_weight, class_sigma)"""
    n, dim = desc.shape
    fc, fm = _u32(feat_class), _u32(feat_model)
    ctr = np.ascontiguousarray(np.asarray(feat_center, dtype=np.float32))
    C_ = int(n_classes if n_classes is not None else fc.max() + 1)
    ncw = n if codewords is None else int(codewords.shape[0])
    nw = C.c_int32(0)
    word_src = np.empty(ncw, np.uint32); vo = np.empty(ncw + 1, np.uint32); vf = np.empty(n * k, np.uint32)
    vxyz = np.empty((n * k, 3), np.float32); vw = np.empty(n * k, np.float32); vcw = np.empty(n * k, np.float32); sig = np.empty(C_, np.float32)
    ctx.check(lib().ismhip_train_activate(ctx._h, C.c_int(metric), C.c_int(n), C.c_int(dim), _p(desc), _p(lrf), _p(kx), _p(ky), _p(kz), _p(fc), _p(fm),
                                          _p(ctr), C.c_int(ncw), _p(codewords), C.c_int(k), C.c_int(1 if clean_up else 0), C.c_int(C_), C.byref(nw), _p(word_src),
                                          _p(vo), _p(vf), _p(vxyz), _p(vw), _p(vcw), _p(sig)), "ismhip_train_activate")
    m = nw.value; nv = int(vo[m])
    return dict(word_src=word_src[:m].copy(), vote_offsets=vo[:m + 1].copy(), vote_feature=vf[:nv].copy(), vote_xyz=vxyz[:nv].copy(),
                vote_weight=vw[:nv].copy(), vote_class_weight=vcw[:nv].copy(), class_sigma=sig)


CENTERS_INIT = {"FLANN_CENTERS_RANDOM": 0, "FLANN_CENTERS_GONZALES": 1, "FLANN_CENTERS_KMEANSPP": 2}


def kmeans(ctx, metric, desc, n_clusters, max_iterations=1000, centers_init="FLANN_CENTERS_KMEANSPP", seed=0):
    """ClusteringKMeans::cluster on the device -> (centers [m, dim] device, assign [n] device int32, dist [n] device, iterations)"""
    torch = _torch()
    n, dim = desc.shape
    kc = min(int(n_clusters), n)
    centers = torch.empty((kc, dim), dtype=torch.float32, device=desc.device)
    assign = torch.empty(n, dtype=torch.int32, device=desc.device)
    dist = torch.empty(n, dtype=torch.float32, device=desc.device)
    m = C.c_int32(0); it = C.c_int32(0)
    ctx.check(lib().ismhip_kmeans(ctx._h, C.c_int(metric), C.c_int(n), C.c_int(dim), _p(desc), C.c_int(kc), C.c_int(max_iterations),
                                  C.c_int(CENTERS_INIT[centers_init] if isinstance(centers_init, str) else int(centers_init)), C.c_ulonglong(seed),
                                  _p(centers), _p(assign), _p(dist), C.byref(m), C.byref(it)), "ismhip_kmeans")
    return centers[:m.value], assign, dist, it.value


PARTIAL_SHOT_SIGNATURES = {          # Codebook::getSignatureMask (codebook/codebook.cpp:952-1036): kept signatures of the 32
    "front": range(8, 24), "dense_x": range(8, 24), "back": list(range(0, 8)) + list(range(24, 32)), "sparse_x": list(range(0, 8)) + list(range(24, 32)),
    "left": range(16, 32), "positive_y": range(16, 32), "right": range(0, 16), "negative_y": range(0, 16),
    "top": range(1, 32, 2), "dense_z": range(1, 32, 2), "bottom": range(0, 32, 2), "sparse_z": range(0, 32, 2),
    "dense_x_or_z": sorted(set(range(8, 24)) | set(range(1, 32, 2))), "dense_x_and_z": range(9, 24, 2),
    "front_turn_left": range(12, 28), "front_turn_right": range(4, 20),
}


def partial_shot_columns(kind):
    """descriptor columns of SHOT-352 kept by UsePartialShot / PartialShotType (unknown type: the complete descriptor, with the reference's warning)"""
    sig = PARTIAL_SHOT_SIGNATURES.get(kind, range(32))
    return np.asarray([s * 11 + j for s in sig for j in range(11)], np.int32)


def gather_columns(ctx, src, cols):
    torch = _torch()
    cols = np.ascontiguousarray(cols, np.int32)
    out = torch.empty((src.shape[0], len(cols)), dtype=torch.float32, device=src.device)
    ctx.check(lib().ismhip_gather_columns(ctx._h, C.c_int(src.shape[0]), C.c_int(src.shape[1]), _p(src), C.c_int(len(cols)), _p(cols), _p(out)), "ismhip_gather_columns")
    return out
