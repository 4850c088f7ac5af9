"""Per-kernel micro-benchmarks on the GPU box (development tool; numbers quoted in DESIGN.md come from bench.py).
usage: python tools/microbench.py [knn] [desc] [--nq N --words N --dim D --reps R]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="*", default=["knn", "desc"])
ap.add_argument("--nq", type=int, default=32768)
ap.add_argument("--words", type=int, default=102400)
ap.add_argument("--dim", type=int, default=352)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--objects", type=int, default=32)
ap.add_argument("--metric", type=int, default=0)
args = ap.parse_args()
import torch
pkg = ge.load_package()
capi, pipeline, synthetic = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0")
ctx = capi.Ctx(0)
ctx.timers_enable(True)

if "knn" in args.what:
    g = torch.Generator(device="cpu").manual_seed(1)
    words = torch.rand((args.words, args.dim), generator=g); words /= words.norm(dim=1, keepdim=True)
    q = torch.rand((args.nq, args.dim), generator=g); q /= q.norm(dim=1, keepdim=True)
    n = args.words
    cb = capi.Codebook(ctx, words.numpy(), np.arange(n + 1, dtype=np.uint32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32),
                       np.zeros(n, np.uint32), 1, np.ones(1, np.float32))
    qd = q.to(dev)
    capi.knn(ctx, cb, args.metric, qd, 1); ctx.sync(); ctx.timers_reset()
    for _ in range(args.reps):
        capi.knn(ctx, cb, args.metric, qd, 1)
    ctx.sync()
    name = "knn_l2_mfma" if args.metric == 0 else "knn_chi2"
    ms, cnt = ctx.timer(name); ms_all, _ = ctx.timer("knn"); ms_fb, _ = ctx.timer("knn_fallback")
    print(f"knn_fallback {ms_fb/cnt:.3f} ms/launch; flagged queries {int(ctx.timer('knn_flagged_queries')[0])} slot items {int(ctx.timer('knn_flagged_items')[0])} (last launch)")
    flop = 2.0 * args.nq * n * args.dim
    print(f"knn metric={args.metric} nq={args.nq} words={n} dim={args.dim}: {name} {ms/cnt:.3f} ms/launch = {flop/(ms/cnt*1e-3)/1e12:.1f} TFLOP/s (2NqNcD); whole call {ms_all/cnt:.3f} ms")

if "desc" in args.what:
    ds = synthetic.Dataset(10, 908, split=1)
    b = pipeline.DeviceBatch(ds.batch(range(args.objects)), dev)
    rec = pipeline.Recognizer(ctx, pipeline.IsmConfig())
    f = rec.compute_features(b, want_counts=True); ctx.sync(); ctx.timers_reset()
    for _ in range(args.reps):
        f = rec.compute_features(b, want_counts=True)
    ctx.sync()
    m = int(f["counts"].to(torch.int64).sum().item()); nkp = int(b.kp_off[-1])
    for name in ("grid", "lrf", "shot352"):
        ms, cnt = ctx.timer(name)
        print(f"{name}: {ms/max(cnt,1):.3f} ms per {args.objects} objects")
    ms, cnt = ctx.timer("shot352")
    by = m * 24.0 + nkp * 1456
    print(f"shot352 gather-model {by/(ms/cnt*1e-3)/1e9:.1f} GB/s, mean neighbours {m/nkp:.1f}")
