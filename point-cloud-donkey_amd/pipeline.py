"""Batch recognition driver over the C ABI (include/ismhip.h): the Python twin of ImplicitShapeModel::train()/detect().

Reference call stack mirrored here (paths relative to /root/reference/src/implicit_shape_model):
  detect : implicit_shape_model.cpp:583-712  computeFeatures -> removeNaNFeatures -> castVotes -> findMaxima
  train  : implicit_shape_model.cpp:252-500 with Clustering "None" (clustering_none.cpp:25-35), KNN K=1 activation
           (codebook.cpp:64-224: activate, per-class sigma, k=1 clean-up) and Uniform feature ranking.
All heavy work is inside libismhip.so; this file only sequences the calls and keeps every intermediate on the device
(one host sync per batch, for the NaN-feature compaction counts). torch is device memory + streams, nothing else.
"""
import os
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class IsmConfig:
    feature: str = "SHOT"            # "SHOT" | "CSHOT" | "FPFH"   (Features.Type)
    radius: float = 0.4              # Features.Radius
    lrf_radius: float = 0.3          # Features.ReferenceFrameRadius
    distance: str = "Euclidean"      # DistanceType: "Euclidean" (FLANN L2, squared) | "ChiSquared"
    activation: str = "KNN"          # ActivationStrategy.Type: "KNN" | "KNNRule"
    k: int = 1                       # ActivationStrategy.K
    use_distance_ratio: bool = False
    distance_ratio_threshold: float = 0.95
    use_class_weight: bool = False
    use_vote_weight: bool = False
    use_matching_weight: bool = False
    use_codeword_weight: bool = False
    bandwidth: float = 0.6           # Voting.Bandwidth
    threshold: float = 1e-3
    max_iter: int = 1000
    kernel: str = "Gaussian"
    maxima_suppression: str = "Average"
    min_votes_threshold: int = 1
    min_threshold: float = 0.0
    best_k: int = -1
    max_maxima: int = 16
    average_rotation: bool = False   # Voting.AverageRotation (voting.cpp:210-215): out["bbox_quat"]
    max_filter: str = "None"         # Voting.MaxFilterType: "None" | "Simple" | "Merge" (ignored in single-object mode, voting.cpp:262-268)
    single_object_mode: bool = False # Voting.SingleObjectMode
    single_object_max_type: str = "Default"   # Voting.SingleObjectMaxType: "Default" | "BandwidthVotes" | "ModelRadiusVotes" | "VotingSpaceVotes"
    use_partial_shot: bool = False   # Codebook.UsePartialShot / PartialShotType (SHOT-352 only, codebook.cpp:416-475)
    partial_shot_type: str = "front"
    voting: str = "MeanShift"        # Voting.Type: "MeanShift" | "Hough3D"
    hough_min_coord: tuple = (-5.0, -5.0, -5.0)   # Voting(Hough3D).MinCoord / MaxCoord / BinSize[0] / UseInterpolation / RelThreshold
    hough_max_coord: tuple = (5.0, 5.0, 5.0)
    hough_bin_size: float = 0.2
    hough_use_interpolation: bool = True
    hough_rel_threshold: float = 0.8
    clustering: str = "None"         # Clustering.Type: "None" | "KMeansCount" | "KMeansFactor" | "KMeansThumbRule"
    cluster_count: int = 10          # Clustering.ClusterCount (KMeansCount)
    cluster_factor: float = 0.2      # Clustering.ClusterFactor (KMeansFactor)
    kmeans_iterations: int = 1000    # Clustering.Iterations
    kmeans_centers_init: str = "FLANN_CENTERS_KMEANSPP"
    kmeans_seed: int = 0             # this build's draws (the reference draws from rand())
    n_classes: int = 10
    use_random_codebook: bool = False
    random_codebook_size: int = 0    # fixed-size seeded subset (reference: UseRandomCodebook/RandomCodebookFactor, codebook.cpp:821-829)
    random_codebook_seed: int = 0x5EED

    @property
    def dim(self):
        return {"SHOT": 352, "CSHOT": 1344, "FPFH": 33}[self.feature]

    @property
    def metric(self):
        return {"Euclidean": capi.METRIC_L2SQ, "ChiSquared": capi.METRIC_CHI2}[self.distance]

    @property
    def weight_flags(self):
        return ((capi.W_CLASS if self.use_class_weight else 0) | (capi.W_VOTE if self.use_vote_weight else 0) |
                (capi.W_MATCHING if self.use_matching_weight else 0) | (capi.W_CODEWORD if self.use_codeword_weight else 0))


class DeviceBatch:
    """A batch of objects resident in HBM as SoA float tensors (+ host offset arrays)."""

    def __init__(self, np_batch, device):
        import torch
        t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)
        self.pt_off = np.asarray(np_batch["pt_off"], np.uint32)
        self.kp_off = np.asarray(np_batch["kp_off"], np.uint32)
        xyz, nrm, kp = np_batch["xyz"], np_batch["normals"], np_batch["kp"]
        self.x, self.y, self.z = t(xyz[:, 0]), t(xyz[:, 1]), t(xyz[:, 2])
        self.nx, self.ny, self.nz = t(nrm[:, 0]), t(nrm[:, 1]), t(nrm[:, 2])
        self.kx, self.ky, self.kz = t(kp[:, 0]), t(kp[:, 1]), t(kp[:, 2])
        self.rgba = t(np_batch["rgba"].astype(np.int64), torch.int32) if "rgba" in np_batch else None
        self.kp_rgba = t(np_batch["kp_rgba"].astype(np.int64), torch.int32) if "kp_rgba" in np_batch else None
        self.labels = np.asarray(np_batch.get("labels", np.zeros(len(self.pt_off) - 1)), np.int32)
        self.n_obj = len(self.pt_off) - 1


class HostStager:
    """Inputs that start in (pinned) HOST memory: uploads run on a copy stream into one of two device buffer sets while the
    previous chunk computes on the ctx stream (double buffering). Used by bench.py's PCIe-inclusive leg and by callers that
    stream a test list through detect(); the resident path (DeviceBatch) is unchanged."""
    FIELDS = ("x", "y", "z", "nx", "ny", "nz", "kx", "ky", "kz", "rgba", "kp_rgba")

    def __init__(self, np_batches, device):
        import torch
        self.device = device
        self.host, self.meta = [], []
        for nb in np_batches:
            xyz, nrm, kp = nb["xyz"], nb["normals"], nb["kp"]
            h = dict(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], nx=nrm[:, 0], ny=nrm[:, 1], nz=nrm[:, 2], kx=kp[:, 0], ky=kp[:, 1], kz=kp[:, 2])
            if "rgba" in nb:
                h["rgba"] = nb["rgba"].astype(np.int64).astype(np.int32); h["kp_rgba"] = nb["kp_rgba"].astype(np.int64).astype(np.int32)
            self.host.append({k: torch.as_tensor(np.ascontiguousarray(v)).pin_memory() for k, v in h.items()})
            self.meta.append((np.asarray(nb["pt_off"], np.uint32), np.asarray(nb["kp_off"], np.uint32), np.asarray(nb.get("labels", np.zeros(len(nb["pt_off"]) - 1)), np.int32)))
        self.bytes_per_pass = int(sum(t.numel() * t.element_size() for h in self.host for t in h.values()))
        cap = {k: max((h[k].numel() for h in self.host if k in h), default=0) for k in self.FIELDS}
        self.slots = [{k: torch.empty((n,), dtype=torch.int32 if "rgba" in k else torch.float32, device=device) for k, n in cap.items() if n}
                      for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self._used = [False, False]

    def _upload(self, j):
        import torch
        s = j % 2
        with torch.cuda.stream(self.copy_stream):
            if self._used[s]:
                self.copy_stream.wait_event(self.free[s])          # the compute stream is done with this buffer set
            for k, t in self.host[j].items():
                self.slots[s][k][:t.numel()].copy_(t, non_blocking=True)
            self.ready[s].record(self.copy_stream)

    def begin(self):
        self._upload(0)

    def get(self, j):
        import torch
        s = j % 2
        torch.cuda.current_stream(self.device).wait_event(self.ready[s])
        if j + 1 < len(self.host):
            self._upload(j + 1)
        b = DeviceBatch.__new__(DeviceBatch)
        b.pt_off, b.kp_off, b.labels = self.meta[j]
        b.n_obj = len(b.pt_off) - 1
        h = self.host[j]
        for k in self.FIELDS:
            setattr(b, k, self.slots[s][k][:h[k].numel()] if k in h else None)
        return b

    def release(self, j):
        import torch
        s = j % 2
        self.free[s].record(torch.cuda.current_stream(self.device))
        self._used[s] = True


class Recognizer:
    def __init__(self, ctx, cfg: IsmConfig):
        self.ctx, self.cfg = ctx, cfg
        self.codebook = None

    # -- ImplicitShapeModel::computeFeatures step f + removeNaNFeatures -------------------------------------
    def compute_features(self, b: DeviceBatch, want_counts=False):
        c, ctx = self.cfg, self.ctx
        cell = min(c.radius, c.lrf_radius if c.feature != "FPFH" else c.radius) * float(os.environ.get("ISMHIP_CELL_SCALE", "0.4"))
        cloud = capi.Cloud(ctx, b.pt_off, b.x, b.y, b.z, b.nx, b.ny, b.nz, cell, rgba=b.rgba if c.feature == "CSHOT" else None)
        lrf = capi.shot_lrf(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, c.lrf_radius)   # Features::operator() always computes LRFs
        if c.feature == "SHOT":
            desc, cnt = capi.shot352(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, lrf, c.radius, want_counts=True)
        elif c.feature == "CSHOT":
            desc, cnt = capi.cshot1344(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, b.kp_rgba, lrf, c.radius, want_counts=True)
        else:
            # FPFH ignores the frames for description, but keypoints with an invalid frame are still dropped first
            desc, cnt = capi.fpfh33(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, c.radius, want_counts=True)
        keep, desc, lrf, kx, ky, kz, src = capi.compact_descriptor_rows(ctx, b.kp_off, desc, lrf, b.kx, b.ky, b.kz)
        out = dict(off=keep, desc=desc, lrf=lrf, kx=kx, ky=ky, kz=kz, src=src, cloud=cloud)
        if want_counts:
            out["counts"] = cnt
        return out

    # -- ImplicitShapeModel::train() with Clustering None / KNN K=1 / Uniform ranking ---------------------------
    def train(self, batches, instance_ids=None):
        """batches: iterable of DeviceBatch whose objects are ordered class-major (the reference iterates std::map by class)."""
        import torch
        c, ctx = self.cfg, self.ctx
        descs, lrfs, kps, cls, inst, model, centers = [], [], [], [], [], [], []
        obj_base = 0
        for b in batches:
            f = self.compute_features(b)
            # AABB centre per object (BoundingBoxType "AABB"; MVBB is not built)
            for o in range(b.n_obj):
                s, e = int(b.pt_off[o]), int(b.pt_off[o + 1])
                mn = torch.stack([b.x[s:e].min(), b.y[s:e].min(), b.z[s:e].min()])
                mx = torch.stack([b.x[s:e].max(), b.y[s:e].max(), b.z[s:e].max()])
                n = int(f["off"][o + 1] - f["off"][o])
                centers.append(((mn + mx) * 0.5).expand(n, 3))
                cls.append(np.full(n, b.labels[o], np.uint32))
                inst.append(np.full(n, (instance_ids[obj_base + o] if instance_ids is not None else obj_base + o), np.uint32))
                model.append(np.full(n, obj_base + o, np.uint32))
            obj_base += b.n_obj
            descs.append(f["desc"]); lrfs.append(f["lrf"]); kps.append(torch.stack([f["kx"], f["ky"], f["kz"]], 1))
            ctx.sync()
        desc = torch.cat(descs); lrf = torch.cat(lrfs); kp = torch.cat(kps); center = torch.cat(centers)
        cls = np.concatenate(cls); inst = np.concatenate(inst); model = np.concatenate(model)
        n = desc.shape[0]
        # Codebook::activate on the device (ismhip_train_activate): one codeword per training feature (clustering_none.cpp), exact
        # kNN activation, class sigma^2, K = 1 clean-up, vote CSR, computeWeights and the statistical class weights
        assert (np.diff(cls.astype(np.int64)) >= 0).all(), "training objects must be ordered class-major"
        knn_rule = getattr(c, "activation", "KNN") == "KNNRule"
        k_act = 1 if knn_rule else c.k                       # KNNRule trains with plain 1-NN and keeps multi-vote codewords
        # clustering (implicit_shape_model.cpp:445-475): k-means centres become the codewords; "None" = every feature its own codeword
        centres = None
        kind = getattr(c, "clustering", "None")
        if kind != "None":
            count = {"KMeansCount": lambda: c.cluster_count, "KMeansFactor": lambda: int(round(n * min(c.cluster_factor, 1.0))),
                     "KMeansThumbRule": lambda: int(round(float(np.sqrt(np.float32(n / 2.0)))))}[kind]()
            centres, self.cluster_indices, _, self.kmeans_iterations = capi.kmeans(ctx, c.metric, desc, max(1, count), c.kmeans_iterations,
                                                                                   c.kmeans_centers_init, c.kmeans_seed)
        act = capi.train_activate(ctx, c.metric, desc, lrf, kp[:, 0].contiguous(), kp[:, 1].contiguous(), kp[:, 2].contiguous(), cls, model,
                                  center.cpu().numpy(), k=k_act, clean_up=(not knn_rule and c.k == 1), n_classes=c.n_classes, codewords=centres)
        words_h = (desc if centres is None else centres).cpu().numpy()
        keep_words, vote_off, src = act["word_src"].astype(np.int64), act["vote_offsets"], act["vote_feature"].astype(np.int64)
        vote_xyz, vote_w, vote_cw = act["vote_xyz"], act["vote_weight"], act["vote_class_weight"]
        if c.use_random_codebook and 0 < c.random_codebook_size < len(keep_words):
            # fixed-size seeded subset of the codewords (reference: UseRandomCodebook at load time, codebook.cpp:821-829)
            rng = np.random.default_rng(c.random_codebook_seed)
            sel = np.sort(rng.choice(len(keep_words), c.random_codebook_size, replace=False))
            cnt = np.diff(vote_off.astype(np.int64))
            vsel = np.concatenate([np.arange(vote_off[e], vote_off[e + 1]) for e in sel]) if len(sel) else np.zeros(0, np.int64)
            keep_words = keep_words[sel]; vote_off = np.concatenate([[0], np.cumsum(cnt[sel])]).astype(np.uint32)
            src, vote_xyz, vote_w, vote_cw = src[vsel], vote_xyz[vsel], vote_w[vsel], vote_cw[vsel]
        m = len(keep_words)
        self.cb_host = dict(words=words_h[keep_words], vote_offsets=vote_off.astype(np.uint32), vote_xyz=vote_xyz,
                            vote_class=cls[src], vote_instance=inst[src], class_sigma=act["class_sigma"],
                            vote_class_weight=vote_cw.astype(np.float32), vote_weight=vote_w.astype(np.float32), word_weight=np.ones(m, np.float32),
                            word_class=cls[keep_words])
        self.load_codebook(self.cb_host)
        return self.cb_host

    def load_codebook(self, cb):
        if self.codebook is not None:
            self.codebook.close()
        self.cb_host = cb
        self.partial_cols = None
        if self.cfg.use_partial_shot:
            # Codebook::iLoadData builds the partial codewords when the model is loaded (codebook.cpp:862-930); detection then masks every
            # feature the same way. Only plain SHOT: the reference's CSHOT branch leaks hist_size = 31 into the next feature's shape part.
            if self.cfg.feature != "SHOT":
                raise capi.IsmHipError("UsePartialShot is built for SHOT-352 only (the reference's partial CSHOT yields descriptors of unequal length)")
            self.partial_cols = capi.partial_shot_columns(self.cfg.partial_shot_type)
            cb = dict(cb, words=np.ascontiguousarray(np.asarray(cb["words"], np.float32)[:, self.partial_cols]))
        self.codebook = capi.Codebook(self.ctx, cb["words"], cb["vote_offsets"], cb["vote_xyz"], cb["vote_class"], cb["vote_instance"],
                                      self.cfg.n_classes, cb["class_sigma"], word_weight=cb.get("word_weight"),
                                      vote_weight=cb.get("vote_weight"), vote_class_weight=cb.get("vote_class_weight"),
                                      vote_bbox_quat=cb.get("vote_bbox_quat"), vote_bbox_size=cb.get("vote_bbox_size"))
        if cb.get("word_class") is not None:
            self.codebook.set_word_class(cb["word_class"])       # Codeword::getClassId = class of the feature the word was made from

    # -- ImplicitShapeModel::detect() over a batch --------------------------------------------------------------
    def detect(self, b: DeviceBatch, keep_intermediates=False):
        c, ctx, cb = self.cfg, self.ctx, self.codebook
        f = self.compute_features(b)
        q = f["desc"] if getattr(self, "partial_cols", None) is None else capi.gather_columns(ctx, f["desc"], self.partial_cols)
        if getattr(c, "activation", "KNN") == "KNNRule":
            idx, dist = capi.knn_rule(ctx, cb, c.metric, q, c.distance_ratio_threshold)
        elif c.use_distance_ratio and c.k == 1:
            idx, dist = capi.knn_ratio(ctx, cb, c.metric, q, c.distance_ratio_threshold)
        else:
            idx, dist = capi.knn(ctx, cb, c.metric, q, c.k)
        votes = capi.cast_votes(ctx, cb, c.weight_flags, f["lrf"], f["kx"], f["ky"], f["kz"], idx, dist, want_bbox=c.average_rotation)
        slot_off = f["off"].astype(np.uint64) * (c.k * cb.max_votes)
        max_filter = capi.MAXFILTER_NONE if c.single_object_mode else {"Simple": capi.MAXFILTER_SIMPLE, "Merge": capi.MAXFILTER_MERGE}.get(c.max_filter, capi.MAXFILTER_NONE)
        if c.voting == "Hough3D":
            mx = capi.hough3d_maxima(ctx, slot_off.astype(np.uint32), votes, c.n_classes, c.hough_bin_size, c.hough_min_coord, c.hough_max_coord,
                                     c.hough_use_interpolation, c.hough_rel_threshold, c.min_votes_threshold, c.min_threshold, c.best_k, c.max_maxima,
                                     max_filter=max_filter, average_rotation=c.average_rotation)
        else:
            som = capi.SOM_MEANSHIFT if not c.single_object_mode else {"BandwidthVotes": capi.SOM_BANDWIDTH, "ModelRadiusVotes": capi.SOM_MODEL_RADIUS,
                                                                      "VotingSpaceVotes": capi.SOM_COMPLETE_VOTING_SPACE}.get(c.single_object_max_type, capi.SOM_MEANSHIFT)
            cen = rad = None
            if som != capi.SOM_MEANSHIFT:                      # the query point is the centroid of the object's cloud (voting_mean_shift.cpp:126-132)
                cen = capi.cloud_centroids(ctx, f["cloud"], b.x.device)
                rad = capi.cloud_radii(ctx, f["cloud"], cen)
            mx = capi.find_maxima(ctx, slot_off.astype(np.uint32), votes, c.n_classes, c.bandwidth, c.threshold, c.max_iter,
                                  capi.KERNEL_GAUSSIAN if c.kernel == "Gaussian" else capi.KERNEL_UNIFORM,
                                  {"Average": capi.SUPPRESS_AVERAGE, "Suppress": capi.SUPPRESS_SUPPRESS}.get(c.maxima_suppression, capi.SUPPRESS_NONE),
                                  c.min_votes_threshold, c.min_threshold, c.best_k, c.max_maxima, max_filter=max_filter,
                                  average_rotation=c.average_rotation, single_object_max_type=som, object_centroid=cen, object_radius=rad)
        if keep_intermediates:
            mx.update(features=f, idx=idx, dist=dist, votes=votes, slot_off=slot_off.astype(np.uint32))
        else:
            mx["_keep"] = (f, idx, dist, votes)   # outputs are produced asynchronously: keep inputs alive until the caller syncs
        return mx
