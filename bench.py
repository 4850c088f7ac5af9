#!/usr/bin/env python
"""bench.py — objects/sec classified on the ModelNet10-like SHOT-352 workload (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL. Either the driver starts the ranks (`python -m torch.distributed.run ... bench.py --gpus N`:
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the environment) or, when RANK is unset, this script starts them itself as a CHILD
`torch.distributed.run` process BEFORE anything in this process touches a GPU (a process that has initialised the GPU is never
re-exec'ed) and exits with the child's code. A WORLD_SIZE that differs from --gpus, or fewer visible GPUs than --gpus, is an error.

A step = one pass of the recognition hot path (grid -> LRF -> SHOT-352 -> NaN compaction -> exact kNN -> vote casting ->
mean-shift maxima) over the FIXED test split of --objects objects (908 = ModelNet10's test split), inputs already resident in
HBM, followed by the path's one exchange: an all-gather of the per-object class-score records (RCCL). The split is cut into
contiguous per-rank shards (shard.shard_range; the reference loops over the objects serially, eval_classification.cpp:347-356),
so the total work is fixed as N grows: STRONG scaling. value = objects classified by all ranks / max-over-ranks time.

The JSON line also carries
  roofline      : the dominant kernel (k_knn_l2_ring16, the f16-MFMA candidate stage of the exact kNN, MFMA-bound): algorithmic
                  flop (2 * queries * codewords * 352) / launch time, measured with HIP events on the stream the kernel runs on
                  (ismhip timers), vs the dense F16/BF16 MFMA peak
  roofline_shot : descriptor extraction (k_shot<false>, HBM-bound, gather-model bytes of SURVEY.md §8d)
  value_end_to_end : the same step with every input batch coming from pinned host memory (H2D on a copy stream, double-buffered)
                  and the class scores copied back (D2H) inside the clock — SURVEY §8d's "first H2D -> last result D2H";
                  reported beside `value`, never instead of it
  cpu_baseline  : the CPU oracle (kind "port": the reference itself cannot be built here) timed on this host's cores on a
                  bounded sample of the same workload; checker/baseline only, never on the measured path.

--config {1,2,3,4} selects BASELINE.json configs[1..4] (1 = default headline; 2 = 2048 keypoints + 10k-word codebook;
3 = CSHOT-1344 chi-squared partial views; 4 = FPFH-33 + SHOT-352 dual models, 50k-word codebooks), each with its own roofline.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (ISMHIP_KNN_MODE=f32)
PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector peak (k_knn_chi2)
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configs[i]")
    ap.add_argument("--objects", type=int, default=0, help="objects of the test split classified per step by ALL ranks together (0 = the config's default; 908 for config 1)")
    ap.add_argument("--batch", type=int, default=1024, help="largest launch: a rank's shard is processed in chunks of at most this many objects (the 908-object split is one launch)")
    ap.add_argument("--train-per-class", type=int, default=0, help="training objects per class (0 = the config's default)")
    ap.add_argument("--cpu-objects", type=int, default=24, help="objects of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--points", type=int, default=0)
    ap.add_argument("--keypoints", type=int, default=0)
    ap.add_argument("--classes", type=int, default=0)
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive (value_end_to_end) leg")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo only for --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="FUNCTIONAL run of the N > 1 path on one GPU: every rank uses cuda:0 (needs --backend gloo: "
                    "RCCL refuses two ranks on one device); exercises spawn -> shard -> per-rank training -> gather, measures nothing")
    ap.add_argument("--emit-labels", action="store_true", help="add the predicted class of every object of the last step to the JSON line")
    return ap.parse_args(argv)


CONFIGS = {
    # objects: the real split sizes 908 / 2468 / 153 / 2468 (ModelNet10 test, ModelNet40 test, 51 classes x 3 views, ModelNet40 test);
    # a rank's shard is processed in launches of at most --batch objects
    1: dict(name="configs[1]: ModelNet10-like test split, 16384 pts, 1024 uniform keypoints/object, SHOT-352 (Radius 0.4, LRF 0.3), "
                 "exact kNN K=1 squared-L2, mean-shift bandwidth 0.6",
            classes=10, objects=908, points=16384, keypoints=1024, train_per_class=10, models=[dict(feature="SHOT")]),
    2: dict(name="configs[2]: ModelNet40-like, 2048 keypoints/object, SHOT-352, 10k-word codebook (seeded random subset of the training "
                 "features, the reference's UseRandomCodebook mechanism), exact kNN K=1 squared-L2",
            classes=40, objects=2468, points=16384, keypoints=2048, train_per_class=1,
            models=[dict(feature="SHOT", use_random_codebook=True, random_codebook_size=10000)]),
    3: dict(name="configs[3]: Washington-like coloured partial views, CSHOT-1344, Radius/LRF 0.05, LeafSize 0.02, bandwidth 0.045, "
                 "chi-squared exact kNN K=1, 10k-word codebook",
            classes=51, objects=153, points=8192, keypoints=0, train_per_class=1, dataset=dict(leaf=0.02, scale=0.15, with_color=True, partial_view=True),
            models=[dict(feature="CSHOT", radius=0.05, lrf_radius=0.05, distance="ChiSquared", bandwidth=0.045,
                         use_random_codebook=True, random_codebook_size=10000)]),
    4: dict(name="configs[4]: ModelNet40-like, 2048 keypoints/object, FPFH-33 + SHOT-352 as two models (class scores summed), 50k-word "
                 "codebooks, exact kNN K=1 squared-L2",
            classes=40, objects=2468, points=16384, keypoints=2048, train_per_class=1,
            models=[dict(feature="FPFH", radius=0.3, use_random_codebook=True, random_codebook_size=50000),
                    dict(feature="SHOT", use_random_codebook=True, random_codebook_size=50000)]),
}


def visible_gpus():
    """GPUs this process may use, WITHOUT touching the HIP runtime (a launcher that initialised the GPU must not start the ranks):
    the visible-devices variables if set, else the KFD topology (nodes with SIMDs are GPUs). None = unknown (the ranks then validate
    their own LOCAL_RANK)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    try:
        for node in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(node):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return None
    return n or None


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child torch.distributed.run BEFORE this process touches a GPU."""
    have = visible_gpus()
    if args.share_gpu:
        have = None                                # every rank maps to cuda:0
    if have is not None and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; refusing to run fewer ranks than asked\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def plan_shard(n_objects, rank, world, batch):
    """Strong scaling: the fixed split of n_objects is cut into contiguous per-rank blocks (shard_range of shard.py: rank r owns
    [r*ceil(n/R), (r+1)*ceil(n/R))), and a rank's block into launches of at most `batch` objects of (almost) equal size.
    Returns (lo, hi, records per rank in the all-gather, launch boundaries)."""
    per = (n_objects + world - 1) // world
    lo = min(n_objects, rank * per)
    hi = min(n_objects, lo + per)
    n_chunks = max(1, -(-(hi - lo) // batch))
    bounds = [lo + (hi - lo) * j // n_chunks for j in range(n_chunks + 1)]
    return lo, hi, per, bounds


def latest_traffic_json(config=1):
    """beyond-L2 bytes per launch from the newest committed PMC pass of this very command (profiles/round*_pmc_traffic.json for the
    headline config, profiles/round*_cfgN_pmc_traffic.json for --config N)"""
    pat = "round*_pmc_traffic.json" if config == 1 else f"round*_cfg{config}_pmc_traffic.json"
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pat)) if config != 1 or "_cfg" not in os.path.basename(f))
    for f in reversed(files):
        try:
            return os.path.relpath(f, ROOT), json.load(open(f))
        except (OSError, ValueError):
            continue
    return None, None


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ        # started by torch.distributed.run
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start one rank per GPU\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    assert torch.cuda.is_available(), "bench.py needs an MI355X; the hot path has no CPU fallback"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu and args.backend != "gloo":
        sys.stderr.write("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)\n")
        sys.exit(2)
    dev_index = 0 if args.share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} has no GPU {dev_index} (visible: {torch.cuda.device_count()})\n")
        sys.exit(2)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    distributed = launched
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world

    pkg = ge.load_package()
    capi, pipeline, synthetic, shard = pkg.capi, pkg.pipeline, pkg.synthetic, pkg.shard
    ctx = capi.Ctx(dev_index)             # on torch's current stream
    cdef = CONFIGS[args.config]
    C = args.classes or cdef["classes"]
    G = args.objects or cdef["objects"]
    n_points = args.points or cdef["points"]
    n_kp = args.keypoints or cdef["keypoints"]
    tpc = args.train_per_class or cdef["train_per_class"]
    ds_kw = dict(cdef.get("dataset", {}))
    if n_kp:
        ds_kw["n_keypoints"] = n_kp

    # ---- models: codebooks trained the reference's way from the synthetic training split (replicated per rank)
    t0 = time.time()
    n_train = tpc * C
    train = synthetic.Dataset(C, n_train, split=0, n_points=n_points, **ds_kw)
    order = sorted(range(n_train), key=lambda i: (train.label(i), i))
    recs, cbs, cfgs = [], [], []
    tb = [pipeline.DeviceBatch(train.batch(order[s:s + 32]), dev) for s in range(0, n_train, 32)]
    ctxs = []
    for mi, m in enumerate(cdef["models"]):
        cfg = pipeline.IsmConfig(k=1, n_classes=C, max_maxima=16, **m)
        mctx = ctx if mi == 0 else capi.Ctx(dev_index)        # one library context per model: separate kernel timers, same (torch current) stream
        ctxs.append(mctx)
        rec = pipeline.Recognizer(mctx, cfg)
        cbs.append(rec.train(tb))
        recs.append(rec); cfgs.append(cfg)
    del tb
    n_words = [int(cb["words"].shape[0]) for cb in cbs]
    t_train = time.time() - t0

    # ---- inputs: this rank's shard of the fixed test split, resident in HBM before the clock starts
    test = synthetic.Dataset(C, G, split=1, n_points=n_points, **ds_kw)
    lo, hi, pad, bounds = plan_shard(G, rank, world, args.batch)
    assert (lo, hi) == shard.shard_range(G, rank, world)
    chunks_h, chunks, chunk_ids = [], [], []
    for j in range(len(bounds) - 1):
        ids = list(range(bounds[j], bounds[j + 1]))
        if not ids:
            continue
        nb = test.batch(ids)
        chunks_h.append(nb)
        chunks.append(pipeline.DeviceBatch(nb, dev))
        chunk_ids.append(torch.as_tensor(ids, device=dev))
    torch.cuda.synchronize()

    def classify(b):
        outs = [r.detect(b) for r in recs]
        score = outs[0]["class_score"] if len(outs) == 1 else sum(o["class_score"] for o in outs)   # configs[4]: harness-level score fusion
        return score, outs

    def step(batches):
        keep, rows, n = [], torch.full((pad, 2 + C), -1.0, dtype=torch.float32, device=dev), 0
        for b, ids in zip(batches, chunk_ids):
            score, outs = classify(b)
            rows[n:n + len(ids)] = shard.pack_records(ids, score, len(ids))
            n += len(ids)
            keep.append(outs)
        return shard.all_gather_records(rows, world), keep

    for i in range(args.warmup):
        g, _ = step(chunks)
    torch.cuda.synchronize()
    for c_ in ctxs:
        c_.timers_enable(True)
        c_.timers_reset()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        g, _ = step(chunks)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- accuracy of the last gathered step (sanity, not part of the metric): every object of the split must be there once
    oi, best, _ = shard.unpack_records(g)
    assert oi.tolist() == list(range(G)), "the all-gather did not return every object of the split exactly once"
    labels = torch.as_tensor([test.label(int(k)) for k in oi.tolist()], device=best.device)
    correct, total = int((best == labels).sum().item()), int(len(oi))

    # ---- per-kernel device time (HIP events on the ctx stream, timed region only; this rank), one timer set per model
    names = ["grid", "lrf", "shot352", "cshot1344", "fpfh33", "knn", "knn_rotate", "knn_l2_mfma", "knn_chi2", "knn_rerank", "knn_stage2", "knn_fallback", "cast_votes", "maxima"]
    tms = [{n: c_.timer(n) for n in names} for c_ in ctxs]
    tm = {n: (sum(t[n][0] for t in tms), sum(t[n][1] for t in tms)) for n in names}
    knn_fb = {"queries": int(ctx.timer("knn_flagged_queries")[0]), "slot_items": int(ctx.timer("knn_flagged_items")[0]),
              "stage2_queries": int(ctx.timer("knn_stage2_queries")[0])}
    for c_ in ctxs:
        c_.timers_enable(False)
    rooflines = {}
    if rank == 0:
        # features actually searched (NaN rows removed) and neighbour visits come from one extra, untimed pass over the shard
        knn_lines, desc_lines = [], []
        for rec, cfg, nw, tmm in zip(recs, cfgs, n_words, tms):
            nq_sum = m_sum = nkp = npts = 0
            for b in chunks:
                f = rec.compute_features(b, want_counts=True)
                nq_sum += int(f["off"][-1]); m_sum += int(f["counts"].to(torch.int64).sum().item()); nkp += int(b.kp_off[-1]); npts += int(b.pt_off[-1])
            nq_l = nq_sum / len(chunks)
            if cfg.distance == "Euclidean" and tmm["knn_l2_mfma"][1] > 0:
                ms_knn = tmm["knn_l2_mfma"][0] / tmm["knn_l2_mfma"][1]
                flop_eff = 2.0 * nq_l * nw * cfg.dim
                knn_mode = os.environ.get("ISMHIP_KNN_MODE", "f16")
                if cfg.dim <= 64 and knn_mode == "f16" and not rec.codebook.stage1_dims:
                    # short descriptors (FPFH-33): the exact-f32 MFMA contraction is the candidate kernel
                    kname, peak, flop, extra = "k_knn_l2_mfma", PEAK_FP32_MFMA_TFLOPS, flop_eff, {}
                else:
                    # stage 1 runs on the m leading rotated coordinates of the codebook (csrc/pca.hip; m = dim when the codebook has no
                    # rotated image) plus a sampling pre-pass over every 32nd codeword tile: `achieved` counts the flops ISSUED by
                    # those launches; the 2*Nq*Nc*dim figure of SURVEY 8(d) is reported beside it as `effective`, NOT as the fraction
                    m1 = rec.codebook.stage1_dims or cfg.dim
                    pre = (1.0 + 1.0 / float(os.environ.get("ISMHIP_KNN_PRE_STEP", "32"))) if (rec.codebook.stage1_dims and rec.codebook.stage1_energy < 1.0 and nw > 127 * 256 and os.environ.get("ISMHIP_KNN_PREPASS", "1") != "0") else 1.0
                    kname, peak, mult = {"f16": ("k_knn_l2_ring" if os.environ.get("ISMHIP_KNN_RING32") == "1" else "k_knn_l2_ring16", PEAK_F16_MFMA_TFLOPS, 1.0),
                                         "bf16x3": ("k_knn_l2_mfma16<bf16x3>", PEAK_F16_MFMA_TFLOPS, 3.0),
                                         "f32": ("k_knn_l2_mfma", PEAK_FP32_MFMA_TFLOPS, 1.0)}.get(knn_mode, ("k_knn_l2_ring16", PEAK_F16_MFMA_TFLOPS, 1.0))
                    flop = 2.0 * nq_l * nw * m1 * pre if knn_mode == "f16" else flop_eff * mult
                    extra = {"stage1_dims": m1, "prepass_share": round(pre - 1.0, 4),
                             "effective": {"flop_per_launch": flop_eff, "achieved": round(flop_eff / (ms_knn * 1e-3) / 1e12, 3),
                                           "note": "2*Nq*Nc*dim / stage-1 time: what an all-dimension search would have to sustain; not a fraction of any peak"}}
                ach = flop / (ms_knn * 1e-3) / 1e12
                knn_lines.append(dict({"kernel": kname, "feature": cfg.feature, "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                                       "frac": round(ach / peak, 4), "traffic": None, "flop_per_launch": flop, "ms_per_launch": round(ms_knn, 4),
                                       "queries_per_launch": nq_l, "codebook_words": nw,
                                       "note": "candidate stage of the exact kNN: f16 MFMA scores over the leading rotated coordinates are lower bounds of "
                                               "the functor values; every returned neighbour is re-ranked with the exact f32 FLANN functor and proven, "
                                               "unproven queries are searched again on a longer rotated image or in all dimensions (knn_stage2) and, failing that, by the exact scan (DESIGN.md 4.1)"}, **extra))
            if cfg.distance == "ChiSquared" and tmm["knn_chi2"][1] > 0:
                ms = tmm["knn_chi2"][0] / tmm["knn_chi2"][1]
                hell = rec.codebook is not None and os.environ.get("ISMHIP_KNN_HELLINGER", "1") != "0"
                if hell:
                    flop = 2.0 * nq_l * nw * cfg.dim
                    ach = flop / (ms * 1e-3) / 1e12
                    knn_lines.append({"kernel": "k_knn_l2_ring16 / k_knn_l2_mfma16 on sqrt images (Hellinger candidates)", "feature": cfg.feature, "bound": "mfma",
                                      "achieved": round(ach, 3), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4),
                                      "traffic": None, "flop_per_launch": flop, "ms_per_launch": round(ms, 4), "queries_per_launch": nq_l, "codebook_words": nw,
                                      "note": "chi-square candidates: |sqrt q - sqrt c|^2 <= chi2 as a dense f16 contraction; the exact chi-square functor runs "
                                              "in the re-rank (knn_rerank) and in the list-and-evaluate stage of the unproven queries (knn_stage2), which are "
                                              "latency-bound and carry most of this config's kNN time (kernel_ms_per_step)"})
                else:
                    flop = 5.0 * nq_l * nw * cfg.dim
                    ach = flop / (ms * 1e-3) / 1e12
                    knn_lines.append({"kernel": "k_knn_chi2", "feature": cfg.feature, "bound": "valu", "achieved": round(ach, 3), "peak": PEAK_FP32_VALU_TFLOPS,
                                      "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_VALU_TFLOPS, 4), "traffic": None,
                                      "flop_per_launch": flop, "ms_per_launch": round(ms, 4), "queries_per_launch": nq_l, "codebook_words": nw,
                                      "note": "5 flop per (query, word, dim) element: sub, add, mul, rcp, fma (SURVEY 8d)"})
            tname = {"SHOT": "shot352", "CSHOT": "cshot1344", "FPFH": "fpfh33"}[cfg.feature]
            if tmm[tname][1] > 0:
                ms_d = tmm[tname][0] / tmm[tname][1]
                if cfg.feature == "FPFH":
                    # sum_{p in U} M_p*24 + |U|*132 + sum_k M_k*136 + K*132. U (surface points inside some keypoint ball) and M_p
                    # are not returned by the call: U is taken as ALL points and M_p as the keypoints' mean neighbour count (both upper
                    # estimates at radius 0.3 on unit-size objects with 2048 keypoints, where nearly every point is in U)
                    mbar = m_sum / max(1, nkp)
                    bytes_d = (npts * mbar * 24.0 + npts * 132.0 + m_sum * 136.0 + nkp * 132.0) / len(chunks)
                    kern = "k_spfh + k_fpfh_sum + k_fpfh_mark"
                else:
                    per_nb, per_kp = (24.0, 12 + 36 + 352 * 4) if cfg.feature == "SHOT" else (28.0, 12 + 36 + 4 + 1344 * 4)
                    bytes_d = (m_sum * per_nb + nkp * per_kp) / len(chunks)
                    kern = "k_shot<%s>" % ("true" if cfg.feature == "CSHOT" else "false")
                gbs = bytes_d / (ms_d * 1e-3) / 1e9
                desc_lines.append({"kernel": kern, "feature": cfg.feature, "bound": "hbm", "model": "gather: every neighbour visit counted as one record read (SURVEY 8d); the "
                                   "clouds are cache-resident, so this is L2/LDS-side throughput -- `traffic` (config 1: committed PMC pass) is what crosses the XCD L2s",
                                   "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                   "traffic": None, "bytes_per_launch": bytes_d, "ms_per_launch": round(ms_d, 4),
                                   "mean_neighbours": round(m_sum / max(1, nkp), 1),
                                   "lrf_plus_descriptor_ms_per_512_objects": round((tmm["lrf"][0] + tmm[tname][0]) / args.steps / max(1, hi - lo) * 512, 3)})
        # the dominant kernel of the step is the `roofline`; every other model / stage follows in `roofline_more`
        knn_lines.sort(key=lambda r: -r["ms_per_launch"]); desc_lines.sort(key=lambda r: -r["ms_per_launch"])
        if knn_lines and (not desc_lines or knn_lines[0]["ms_per_launch"] >= desc_lines[0]["ms_per_launch"] or args.config == 1):
            rooflines["roofline"] = knn_lines[0]
        elif desc_lines:
            rooflines["roofline"] = desc_lines[0]
        if desc_lines:
            rooflines["roofline_shot"] = desc_lines[0]
        rest = [r for r in knn_lines + desc_lines if r is not rooflines.get("roofline") and r is not rooflines.get("roofline_shot")]
        if rest:
            rooflines["roofline_more"] = rest
        # beyond-L2 bytes per launch come from the committed PMC passes of this very command (separate FETCH_SIZE / WRITE_SIZE
        # runs, gfx950 correction applied); they are NOT measured in the run that prints this line, and are null for other shapes
        tf, tj = latest_traffic_json(args.config)
        if tj and world == 1 and tj.get("objects") == G and tj.get("launches_per_step") == len(chunks) and not args.points and not args.keypoints:
            for key in ("roofline", "roofline_shot"):
                r = rooflines.get(key)
                if r and r["kernel"] in tj:
                    r["traffic"] = tj[r["kernel"]]["bytes_per_launch"]
                    r["traffic_source"] = f"committed PMC pass of this command ({tf}); bytes beyond the XCD L2s (Infinity Cache + HBM) per launch"

    # ---- PCIe-inclusive leg (rank-local, N = 1): pinned host inputs -> H2D on a copy stream, double-buffered; scores D2H
    e2e = None
    if world == 1 and not args.no_e2e:
        # the uploads only hide behind compute when there is more than one launch per step: this leg cuts the shard into launches
        # of at most 512 objects (two of 454 for the default split) whatever --batch says
        e_bounds = plan_shard(G, rank, world, min(args.batch, 512))[3]
        if len(e_bounds) - 1 == len(chunks_h):
            e_chunks_h, e_ids = chunks_h, chunk_ids
        else:
            e_chunks_h, e_ids = [], []
            for j in range(len(e_bounds) - 1):
                ids = list(range(e_bounds[j], e_bounds[j + 1]))
                if ids:
                    e_chunks_h.append(test.batch(ids)); e_ids.append(torch.as_tensor(ids, device=dev))
        stager = pipeline.HostStager(e_chunks_h, dev)
        score_h = torch.empty((G, C), dtype=torch.float32).pin_memory()

        def step_e2e():
            n = 0
            stager.begin()
            for j, ids in enumerate(e_ids):
                b = stager.get(j)                      # waits (on the compute stream) for chunk j's upload, starts chunk j+1's
                score, outs = classify(b)
                score_h[n:n + len(ids)].copy_(score, non_blocking=True)
                stager.release(j)
                n += len(ids)
            torch.cuda.synchronize()                   # last result D2H done
        step_e2e()
        t0 = time.perf_counter()
        n_e2e = max(2, min(args.steps, 4))
        for _ in range(n_e2e):
            step_e2e()
        dte = time.perf_counter() - t0
        e2e = {"value": round(G * n_e2e / dte, 3), "unit": "objects/s", "ms_per_step": round(dte / n_e2e * 1e3, 3), "steps": n_e2e,
               "h2d_bytes_per_step": stager.bytes_per_pass, "launches_per_step": len(e_ids), "note": "first H2D -> last class-score D2H (SURVEY §8d); uploads double-buffered on a copy stream"}

    # ---- CPU baseline: the oracle on a bounded sample (rank 0, N = 1, headline config only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_objects > 0:
        ora = ge.load_oracle()
        n_thr = min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)     # the GPU box's CPU share for one GPU
        ora.set_num_threads(n_thr)
        n_cpu = min(G, args.cpu_objects if args.config == 1 else max(4, args.cpu_objects // (1 if args.config == 2 else 3)))   # FPFH / CSHOT are slower per object
        nb = test.batch(list(range(n_cpu)))
        xyz, nrm, kp = nb["xyz"], nb["normals"], nb["kp"]
        t0 = time.perf_counter()
        stages = []
        for cfg, cb in zip(cfgs, cbs):                         # the oracle's version of Recognizer.detect, model by model
            lrf = ora.shot_lrf(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nb["kp_off"], kp[:, 0], kp[:, 1], kp[:, 2], cfg.lrf_radius)
            if cfg.feature == "SHOT":
                desc, _ = ora.shot352(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["kp_off"],
                                      kp[:, 0], kp[:, 1], kp[:, 2], lrf, cfg.radius)
            elif cfg.feature == "CSHOT":
                desc, _ = ora.cshot1344(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["rgba"], nb["kp_off"],
                                        kp[:, 0], kp[:, 1], kp[:, 2], nb["kp_rgba"], lrf, cfg.radius)
            else:
                desc, _ = ora.fpfh33(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["kp_off"],
                                     kp[:, 0], kp[:, 1], kp[:, 2], cfg.radius)
            ok = ~np.isnan(desc).any(1) & ~np.isnan(lrf).any(1)
            idx, dd = ora.knn(cfg.metric, cb["words"], desc[ok], 1)
            votes = ora.cast_votes(cb, cfg.weight_flags, lrf[ok], kp[ok, 0], kp[ok, 1], kp[ok, 2], idx, dd)
            keep_off = np.concatenate([[0], np.cumsum([ok[nb["kp_off"][o]:nb["kp_off"][o + 1]].sum() for o in range(n_cpu)])])
            ora.find_maxima(keep_off.astype(np.uint32), votes, C, cfg.bandwidth, max_maxima=cfg.max_maxima)
            stages.append(f"{cfg.feature}-{cfg.dim} + exact linear-search {cfg.distance} kNN over {int(cb['words'].shape[0])} words")
        t_cpu = time.perf_counter() - t0
        cpu = {"value": round(n_cpu / t_cpu, 4), "unit": "objects/s", "cores": ora.get_num_threads(), "kind": "port",
               "sample": f"{n_cpu} objects of the same workload (oracle: SHOT LRF + " + " and ".join(stages) +
                         f" + votes + mean-shift), {t_cpu:.1f} s, OpenMP over keypoints/queries as the reference; the reference's default "
                         "kNN is an approximate 4-tree FLANN forest, so this is the exact-match (FLANNExactMatch) CPU cost"}

    n_objects = G * args.steps
    if rank == 0:
        cfg0 = cfgs[0]
        line = {
            "metric": "objects/sec classified (ModelNet10-like, SHOT-352)" if args.config == 1 else f"objects/sec classified (BASELINE configs[{args.config}])",
            "value": round(n_objects / dt, 3), "unit": "objects/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "results are the exact f32 values of the reference's functors; the squared-L2 kNN candidate filter runs on f16 MFMA",
            "config": {"workload": cdef["name"], "objects_per_step_all_gpus": G, "objects_per_step_this_rank": hi - lo,
                       "launches_per_step_per_gpu": len(chunks), "points_per_object": n_points, "keypoints_per_object": n_kp or "VoxelGrid leaf",
                       "classes": C, "codebook_words": n_words if len(n_words) > 1 else n_words[0], "descriptor_dim": [c.dim for c in cfgs] if len(cfgs) > 1 else cfg0.dim,
                       "parallelism": f"fixed {G}-object split sharded over {world} GPU(s) (contiguous blocks), codebook replicated, 1 RCCL all-gather/step",
                       "collective_world_size": dist.get_world_size() if distributed else 1},
            "roofline": rooflines.get("roofline"), "roofline_shot": rooflines.get("roofline_shot"),
            "value_end_to_end": e2e, "cpu_baseline": cpu,
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in tm.items() if v[1] > 0},
            "knn_exact_fallback_last_launch": knn_fb, "accuracy_last_step": round(correct / max(1, total), 4), "train_seconds": round(t_train, 1),
        }
        if args.share_gpu:
            line["functional_test_only"] = "all ranks share cuda:0 over gloo: the value is NOT a scaling measurement"
        if args.emit_labels:
            line["labels_last_step"] = [int(x) for x in best.tolist()]
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
