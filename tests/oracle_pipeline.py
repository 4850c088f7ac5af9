"""The oracle's version of Recognizer.detect (checker for the end-to-end configuration tests)."""
import numpy as np


def detect(ora, cfg, cb, nb, n_classes):
    """nb: numpy batch (pt_off, kp_off, xyz, normals, kp [, rgba, kp_rgba]); cb: host codebook dict. Returns oracle maxima dict + extras."""
    xyz, nrm, kp = nb["xyz"], nb["normals"], nb["kp"]
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    kx, ky, kz = kp[:, 0], kp[:, 1], kp[:, 2]
    lrf = ora.shot_lrf(nb["pt_off"], x, y, z, nb["kp_off"], kx, ky, kz, cfg.lrf_radius)
    if cfg.feature == "SHOT":
        desc, _ = ora.shot352(nb["pt_off"], x, y, z, nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["kp_off"], kx, ky, kz, lrf, cfg.radius)
    elif cfg.feature == "CSHOT":
        desc, _ = ora.cshot1344(nb["pt_off"], x, y, z, nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["rgba"], nb["kp_off"], kx, ky, kz, nb["kp_rgba"], lrf, cfg.radius)
    else:
        desc, _ = ora.fpfh33(nb["pt_off"], x, y, z, nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["kp_off"], kx, ky, kz, cfg.radius)
    ok = ~np.isnan(desc).any(1) & ~np.isnan(lrf[:, 0]) & ~np.isnan(lrf[:, 3]) & ~np.isnan(lrf[:, 6])
    n_obj = len(nb["pt_off"]) - 1
    off = np.concatenate([[0], np.cumsum([ok[nb["kp_off"][o]:nb["kp_off"][o + 1]].sum() for o in range(n_obj)])]).astype(np.uint32)
    q = desc[ok]
    words = cb["words"]
    if getattr(cfg, "use_partial_shot", False):                  # codebook.cpp:416-475 / 862-930: the same signature mask on both sides
        import importlib
        cols = __import__("__graft_entry__").load_package().capi.partial_shot_columns(cfg.partial_shot_type)
        q, words = np.ascontiguousarray(q[:, cols]), np.ascontiguousarray(np.asarray(words)[:, cols])
    if cfg.use_distance_ratio and cfg.k == 1:
        idx, dist = ora.knn_ratio(cfg.metric, words, q, cfg.distance_ratio_threshold)
    else:
        idx, dist = ora.knn(cfg.metric, words, q, cfg.k)
    votes = ora.cast_votes(cb, cfg.weight_flags, lrf[ok], kx[ok], ky[ok], kz[ok], idx, dist)
    maxv = int(np.max(np.diff(np.asarray(cb["vote_offsets"], np.int64))))
    slot_off = (off.astype(np.int64) * idx.shape[1] * maxv).astype(np.uint32)
    if getattr(cfg, "voting", "MeanShift") == "Hough3D":
        mx = ora.hough3d_maxima(slot_off, votes, n_classes, cfg.hough_bin_size, cfg.hough_min_coord, cfg.hough_max_coord, cfg.hough_use_interpolation,
                                cfg.hough_rel_threshold, cfg.min_votes_threshold, cfg.min_threshold, cfg.best_k, cfg.max_maxima)
    else:
        mx = ora.find_maxima(slot_off, votes, n_classes, cfg.bandwidth, cfg.threshold, cfg.max_iter, 0 if cfg.kernel == "Gaussian" else 1,
                             {"Average": 0, "Suppress": 1}.get(cfg.maxima_suppression, 2), cfg.min_votes_threshold, cfg.min_threshold, cfg.best_k,
                             cfg.max_maxima)
    mx.update(desc=desc[ok], lrf=lrf[ok], off=off, idx=idx, dist=dist, votes=votes, keep=ok)
    return mx
