"""Seeded synthetic stand-ins for the BASELINE.json datasets.

Objects are unions of 1-4 superquadrics drawn from a class-specific parameter distribution, sampled with N surface
points and analytic outward normals, scaled to unit bounding-sphere radius. Keypoints are pcl::VoxelGrid-style voxel
centroids (reference: keypoints/keypoints_voxel_grid.cpp:30-46) with the leaf bisected until >= K voxels are
occupied; the first K in voxel-index order are kept. Everything is numpy and a pure function of the seeds.
"""
import numpy as np

BASE_SEED = 0x5EED0000


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def class_params(class_id, dataset_seed=0):
    """class-specific part list: (scale[3], e1, e2, offset[3], R[3,3])"""
    rng = np.random.default_rng([BASE_SEED, dataset_seed, 7919, class_id])
    n_parts = 1 + int(rng.integers(0, 4))
    parts = []
    for p in range(n_parts):
        scale = rng.uniform(0.25, 1.0, size=3)
        e1, e2 = rng.uniform(0.25, 1.75, size=2)
        offset = rng.uniform(-0.6, 0.6, size=3) if p > 0 else np.zeros(3)
        parts.append((scale, e1, e2, offset, _rot(rng)))
    return parts


def _spow(v, e):
    return np.sign(v) * np.abs(v) ** e


def make_object(class_id, split, index, n_points=16384, dataset_seed=0, with_color=False):
    """returns xyz[n,3], normals[n,3] (float32) and, optionally, rgba[n] (uint32, 0x00RRGGBB)"""
    rng = np.random.default_rng([BASE_SEED + split * 1_000_000 + index, dataset_seed, class_id])
    parts = class_params(class_id, dataset_seed)
    areas = np.array([np.prod(np.sort(s)[1:]) for s, *_ in parts])
    counts = np.floor(areas / areas.sum() * n_points).astype(int)
    counts[0] += n_points - counts.sum()
    P, Nn = [], []
    for (scale, e1, e2, offset, R), cnt in zip(parts, counts):
        scale = scale * (1.0 + 0.08 * rng.normal(size=3))
        e1j = float(np.clip(e1 + 0.05 * rng.normal(), 0.2, 1.9))
        e2j = float(np.clip(e2 + 0.05 * rng.normal(), 0.2, 1.9))
        d = rng.normal(size=(cnt, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        eta = np.arcsin(np.clip(d[:, 2], -1, 1))
        om = np.arctan2(d[:, 1], d[:, 0])
        ce, se, co, so = np.cos(eta), np.sin(eta), np.cos(om), np.sin(om)
        pts = np.stack([scale[0] * _spow(ce, e1j) * _spow(co, e2j), scale[1] * _spow(ce, e1j) * _spow(so, e2j),
                        scale[2] * _spow(se, e1j)], axis=1)
        nrm = np.stack([_spow(ce, 2 - e1j) * _spow(co, 2 - e2j) / scale[0], _spow(ce, 2 - e1j) * _spow(so, 2 - e2j) / scale[1],
                        _spow(se, 2 - e1j) / scale[2]], axis=1)
        nl = np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = np.where(nl > 1e-12, nrm / np.maximum(nl, 1e-12), d)
        P.append(pts @ R.T + offset + 0.02 * rng.normal(size=3))
        Nn.append(nrm @ R.T)
    P = np.concatenate(P)
    Nn = np.concatenate(Nn)
    P -= P.mean(axis=0)
    P /= np.linalg.norm(P, axis=1).max()
    xyz, normals = P.astype(np.float32), Nn.astype(np.float32)
    if not with_color:
        return xyz, normals
    # procedural texture: class-dependent base hue modulated along the object
    base = np.random.default_rng([BASE_SEED, dataset_seed, 104729, class_id]).uniform(40, 215, size=3)
    tex = 40.0 * np.sin(6.0 * P[:, :1] + np.array([[0.0, 2.0, 4.0]])) + 10.0 * rng.normal(size=(len(P), 3))
    rgb = np.clip(base + tex, 0, 255).astype(np.uint32)
    return xyz, normals, (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]


def voxel_grid(xyz, leaf, rgba=None):
    """pcl::VoxelGrid centroids, ordered by voxel linear index (x fastest)"""
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(xyz * inv).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    uniq, inv_idx, cnt = np.unique(lin, return_inverse=True, return_counts=True)
    cen = np.zeros((len(uniq), 3), np.float64)
    np.add.at(cen, inv_idx, xyz)
    cen = (cen / cnt[:, None]).astype(np.float32)
    if rgba is None:
        return cen
    ch = np.stack([(rgba >> 16) & 0xff, (rgba >> 8) & 0xff, rgba & 0xff], axis=1).astype(np.float64)
    acc = np.zeros((len(uniq), 3), np.float64)
    np.add.at(acc, inv_idx, ch)
    c = (acc / cnt[:, None]).astype(np.uint32)
    return cen, ((c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]).astype(np.uint32)


def keypoints_fixed(xyz, k, rgba=None):
    """bisect the leaf size until >= k voxels are occupied, keep the first k voxel centroids"""
    lo, hi = 1e-3, 1.0          # lo: many voxels, hi: few
    for _ in range(14):
        mid = 0.5 * (lo + hi)
        n = len(voxel_grid(xyz, mid))
        if n >= k:
            lo = mid
        else:
            hi = mid
    out = voxel_grid(xyz, lo, rgba)
    if rgba is None:
        return out[:k]
    return out[0][:k], out[1][:k]


class Dataset:
    """A split of a synthetic dataset, generated object by object (nothing large is stored on disk)."""

    def __init__(self, n_classes, n_objects, split, n_points=16384, n_keypoints=1024, dataset_seed=0, with_color=False,
                 leaf=None, scale=1.0, partial_view=False):
        """scale: metric size of the bounding-sphere radius (1 = ModelNet-like, ~350 = the quick-start "model units",
        ~0.15 = Kinect-like metres). partial_view: keep only the points whose normal faces a per-object viewpoint (single-view
        partial surfaces of BASELINE configs[3]); leaf: fixed VoxelGrid LeafSize instead of a fixed keypoint count."""
        self.n_classes, self.n_objects, self.split = n_classes, n_objects, split
        self.n_points, self.n_keypoints, self.dataset_seed, self.with_color, self.leaf = n_points, n_keypoints, dataset_seed, with_color, leaf
        self.scale, self.partial_view = float(scale), partial_view

    def label(self, i):
        return i % self.n_classes

    def get(self, i):
        c = self.label(i)
        o = make_object(c, self.split, i, self.n_points, self.dataset_seed, self.with_color)
        xyz, nrm = o[0], o[1]
        rgba = o[2] if self.with_color else None
        if self.partial_view:
            view = np.random.default_rng([BASE_SEED, self.dataset_seed, 31337, self.split, i]).normal(size=3)
            view /= np.linalg.norm(view)
            keep = (nrm @ view) > 0.0                       # points with n.view < 0 are culled
            xyz, nrm = xyz[keep], nrm[keep]
            rgba = rgba[keep] if rgba is not None else None
        if self.scale != 1.0:
            xyz = (xyz * np.float32(self.scale)).astype(np.float32)
        if self.leaf is not None:
            kp = voxel_grid(xyz, self.leaf, rgba)
        else:
            kp = keypoints_fixed(xyz, self.n_keypoints, rgba)
        if self.with_color:
            return dict(xyz=xyz, normals=nrm, rgba=rgba.astype(np.uint32), kp=kp[0], kp_rgba=kp[1], label=c)
        return dict(xyz=xyz, normals=nrm, kp=kp, label=c)

    def batch(self, indices):
        """concatenated SoA numpy arrays of several objects + offsets"""
        objs = [self.get(i) for i in indices]
        pt_off = np.zeros(len(objs) + 1, np.uint32)
        kp_off = np.zeros(len(objs) + 1, np.uint32)
        for j, ob in enumerate(objs):
            pt_off[j + 1] = pt_off[j] + len(ob["xyz"])
            kp_off[j + 1] = kp_off[j] + len(ob["kp"])
        out = dict(pt_off=pt_off, kp_off=kp_off,
                   xyz=np.concatenate([ob["xyz"] for ob in objs]), normals=np.concatenate([ob["normals"] for ob in objs]),
                   kp=np.concatenate([ob["kp"] for ob in objs]), labels=np.array([ob["label"] for ob in objs], np.int32))
        if self.with_color:
            out["rgba"] = np.concatenate([ob["rgba"] for ob in objs])
            out["kp_rgba"] = np.concatenate([ob["kp_rgba"] for ob in objs])
        return out
