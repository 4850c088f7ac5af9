"""times the three FPFH kernels apart (rocprofv3 kernel stats are the source of truth; this prints the library timer) -- development tool"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.load_package(); capi, pipeline, synthetic = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0")
ds = synthetic.Dataset(40, 128, split=1, n_points=16384, n_keypoints=2048)
b = pipeline.DeviceBatch(ds.batch(range(128)), dev)
for env, xcd in (("0", "1"), ("0", "0"), ("1", "1")):
    os.environ["ISMHIP_FPFH_DBG"] = env
    os.environ["ISMHIP_XCD_MAP"] = xcd
    ctx = capi.Ctx(0)
    rec = pipeline.Recognizer(ctx, pipeline.IsmConfig(feature="FPFH", radius=0.3, n_classes=40))
    rec.compute_features(b); ctx.sync()
    ctx.timers_enable(True); ctx.timers_reset()
    for _ in range(3):
        f = rec.compute_features(b, want_counts=True)
    ctx.sync()
    print("dbg", env, "xcd_map", xcd, "fpfh33 ms per 128 objects", ctx.timer("fpfh33")[0] / 3, "lrf", ctx.timer("lrf")[0] / 3, "mean neighbours", float(f["counts"].float().mean()))
