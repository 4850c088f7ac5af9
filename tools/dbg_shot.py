"""debug: SHOT neighbour counts / descriptors vs the oracle on the cfg0 objects, for the sweep variants"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.load_package(); ora = ge.load_oracle()
capi, pipeline, syn = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0")
test = syn.Dataset(5, 5, split=1, n_points=8192, leaf=50.0, scale=350.0)
nb = test.batch(range(5))
xyz, nrm, kp = nb["xyz"], nb["normals"], nb["kp"]
lrf_o = ora.shot_lrf(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nb["kp_off"], kp[:, 0], kp[:, 1], kp[:, 2], 50.0)
want, wcnt = ora.shot352(nb["pt_off"], xyz[:, 0], xyz[:, 1], xyz[:, 2], nrm[:, 0], nrm[:, 1], nrm[:, 2], nb["kp_off"], kp[:, 0], kp[:, 1], kp[:, 2], lrf_o, 60.0)
for var in ("0", "2"):
    os.environ["ISMHIP_SHOT_VAR"] = var
    ctx = capi.Ctx(0)
    b = pipeline.DeviceBatch(nb, dev)
    cloud = capi.Cloud(ctx, b.pt_off, b.x, b.y, b.z, b.nx, b.ny, b.nz, 25.0)
    lrf = capi.shot_lrf(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, 50.0)
    d, cnt = capi.shot352(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, torch.as_tensor(lrf_o).to(dev), 60.0, want_counts=True)
    ctx.sync()
    cnt = cnt.cpu().numpy().astype(np.int64); d = d.cpu().numpy()
    bad = np.nonzero(cnt != wcnt)[0]
    print("VAR", var, "lrf nan mismatch", int((np.isnan(lrf.cpu().numpy()[:, 0]) != np.isnan(lrf_o[:, 0])).sum()), "count mismatches", len(bad), [(int(k), int(cnt[k]), int(wcnt[k])) for k in bad[:20]])
    nan_g, nan_w = np.isnan(d).any(1), np.isnan(want).any(1)
    print("   nan rows gpu", int(nan_g.sum()), "oracle", int(nan_w.sum()), "differ at", np.nonzero(nan_g != nan_w)[0][:20].tolist())
    m = ~nan_g & ~nan_w
    err = np.abs(d[m] - want[m]).max(1)
    print("   max desc err", float(err.max()), "rows > 1e-4:", int((err > 1e-4).sum()))
    ctx.close()
