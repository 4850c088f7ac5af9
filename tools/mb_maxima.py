import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
capi, pipeline, syn = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0"); ctx = capi.Ctx(0)
cfg = pipeline.IsmConfig(n_classes=10)
rec = pipeline.Recognizer(ctx, cfg)
train = syn.Dataset(10, 20, split=0)
order = sorted(range(20), key=lambda i: (train.label(i), i))
rec.train([pipeline.DeviceBatch(train.batch(order), dev)])
test = syn.Dataset(10, 908, split=1)
b = pipeline.DeviceBatch(test.batch(range(32)), dev)
out = rec.detect(b, keep_intermediates=True)
votes, so = out["votes"], out["slot_off"]
ctx.timers_enable(True)
for mi, thr, sup in [(1000, 1e-3, 0), (50, 1e-3, 0), (5, 1e-3, 0), (1000, 1e-1, 0), (1000, 1e-3, 1), (1000, 1e-3, 2)]:
    ctx.sync(); ctx.timers_reset()
    for _ in range(3):
        o = capi.find_maxima(ctx, so, votes, 10, cfg.bandwidth, thr, mi, suppression=sup, max_maxima=16)
    ctx.sync()
    ms, n = ctx.timer("maxima")
    print(f"max_iter={mi} thr={thr} suppression={sup}: {ms/n:.3f} ms; n_max[0]={int(o['n'][0])}")
cls = votes["cls"].cpu().numpy()
print("votes per class (object 0):", np.bincount(cls[so[0]:so[1]][cls[so[0]:so[1]] >= 0], minlength=10))
