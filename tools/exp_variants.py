"""A/B timing of library variants selected by environment variables, on the bench workload (development tool).
Data and the codebook are built once; every variant gets a fresh ismhip ctx (the library reads its env switches at ctx creation).
usage: python tools/exp_variants.py [--objects 256] [--reps 3] "NAME=VAL NAME2=VAL" "..." ...   ("" = defaults)"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="*", default=[""])
ap.add_argument("--objects", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--train-per-class", type=int, default=10)
args = ap.parse_args()
import torch
pkg = ge.load_package()
capi, pipeline, synthetic = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0")
C = 10
t0 = time.time()
train = synthetic.Dataset(C, args.train_per_class * C, split=0)
order = sorted(range(train.n_objects), key=lambda i: (train.label(i), i))
test = synthetic.Dataset(C, 908, split=1)
nb = test.batch(range(args.objects))
ctx0 = capi.Ctx(0)
rec0 = pipeline.Recognizer(ctx0, pipeline.IsmConfig(n_classes=C, max_maxima=16))
cb = rec0.train([pipeline.DeviceBatch(train.batch(order[s:s + 32]), dev) for s in range(0, len(order), 32)])
b = pipeline.DeviceBatch(nb, dev)
print(f"setup {time.time() - t0:.1f} s, codebook {cb['words'].shape}", flush=True)
ref = None
names = ["grid", "lrf", "shot352", "knn", "knn_l2_mfma", "knn_stage2", "knn_fallback", "cast_votes", "maxima"]
for v in args.variants:
    keys = []
    for kv in v.split():
        k, val = kv.split("=", 1)
        os.environ[k] = val; keys.append(k)
    ctx = capi.Ctx(0)
    rec = pipeline.Recognizer(ctx, pipeline.IsmConfig(n_classes=C, max_maxima=16))
    rec.load_codebook(cb)
    out = rec.detect(b); ctx.sync()
    ctx.timers_enable(True); ctx.timers_reset()
    t1 = time.perf_counter()
    for _ in range(args.reps):
        out = rec.detect(b)
    ctx.sync()
    wall = (time.perf_counter() - t1) / args.reps * 1e3
    tm = {n: ctx.timer(n) for n in names}
    fb = int(ctx.timer("knn_flagged_queries")[0])
    cls = out["cls"][:, 0].cpu().numpy(); score = out["class_score"].cpu().numpy()
    if ref is None:
        ref = (cls, score)
    same = bool((cls == ref[0]).all()) and float(np.abs(score - ref[1]).max()) < 1e-5
    print(f"[{v or 'default'}] wall {wall:.2f} ms/{args.objects} obj | " + " ".join(f"{n} {tm[n][0] / max(1, tm[n][1]):.3f}" for n in names) +
          f" | stage-2 queries {int(ctx.timer('knn_stage2_queries')[0])} fallback queries {fb} | same result {same}", flush=True)
    ctx.timers_enable(False)
    L = capi.lib()
    if hasattr(L, "ismhip_debug_knn_counters"):                  # library built with -DISM_KNN_DBG_VARIANTS
        import ctypes
        c = (ctypes.c_ulonglong * 256)()
        L.ismhip_debug_knn_counters(c, 1)
        if c[0]:
            h = [c[8 + t] for t in range(248)]
            tot = max(1, sum(h))
            print("    flagged scores by tile index (share of all): " + " ".join(f"{t}:{sum(h[t:t2]) / tot:.3f}" for t, t2 in [(0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 248)]), flush=True)
            print(f"    per wave-tile: any-hit {c[1] / c[0]:.3f}, flagged columns {c[2] / c[0]:.3f}, flagged groups {c[3] / c[0]:.3f}, flagged scores {c[4] / c[0]:.3f}, inserting lanes {c[5] / c[0]:.3f} (wave-tiles {c[0]})", flush=True)
    rec.codebook.close(); ctx.close()
    for k in keys:
        del os.environ[k]
