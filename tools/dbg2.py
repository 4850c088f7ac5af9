import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); ora = ge.load_oracle()
capi, pipeline, syn = pkg.capi, pkg.pipeline, pkg.synthetic
dev = torch.device("cuda:0"); ctx = capi.Ctx(0)
for name, cfg, ds in [("cfg0", pipeline.IsmConfig(feature="SHOT", radius=60.0, lrf_radius=50.0, distance="ChiSquared", bandwidth=50.0, n_classes=5),
                       syn.Dataset(5, 5, split=1, n_points=8192, leaf=50.0, scale=350.0)),
                      ("cfg2", pipeline.IsmConfig(feature="SHOT", n_classes=4), syn.Dataset(4, 3, split=1, n_points=16384, n_keypoints=2048))]:
    nb = ds.batch(range(3))
    b = pipeline.DeviceBatch(nb, dev)
    cell = min(cfg.radius, cfg.lrf_radius) * 0.5
    cloud = capi.Cloud(ctx, b.pt_off, b.x, b.y, b.z, b.nx, b.ny, b.nz, cell)
    lrf = capi.shot_lrf(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, cfg.lrf_radius)
    desc, cnt = capi.shot352(ctx, cloud, b.kp_off, b.kx, b.ky, b.kz, lrf, cfg.radius, want_counts=True)
    xyz, nrm, kp = nb["xyz"], nb["normals"], nb["kp"]
    wl = ora.shot_lrf(nb["pt_off"], xyz[:,0], xyz[:,1], xyz[:,2], nb["kp_off"], kp[:,0], kp[:,1], kp[:,2], cfg.lrf_radius)
    gl = lrf.cpu().numpy()
    dl = np.abs(gl - wl).max(1)
    print(name, "kp", len(kp), "lrf max diff", np.nanmax(dl), "n lrf diff>1e-4:", int((dl > 1e-4).sum()), "nan mismatch", int((np.isnan(gl[:,0]) != np.isnan(wl[:,0])).sum()))
    wd, wc = ora.shot352(nb["pt_off"], xyz[:,0], xyz[:,1], xyz[:,2], nrm[:,0], nrm[:,1], nrm[:,2], nb["kp_off"], kp[:,0], kp[:,1], kp[:,2], gl, cfg.radius)
    gd = desc.cpu().numpy()
    print("   count mismatch:", int((cnt.cpu().numpy().astype(np.uint32) != wc).sum()), "desc (same lrf) max diff", np.nanmax(np.abs(gd - wd)))
    bad = np.nonzero(dl > 1e-4)[0][:5]
    for i in bad:
        print("   kp", i, "gpu", gl[i].round(4), "ora", wl[i].round(4))
    if name == "cfg2":
        diff = np.abs(gd - wd)
        rows = np.nonzero(np.nanmax(diff, axis=1) > 1e-4)[0]
        print("   rows over tol:", len(rows), rows[:10], "objects:", np.unique(np.searchsorted(nb["kp_off"], rows, side="right") - 1))
        r = rows[0]
        cols = np.nonzero(diff[r] > 1e-5)[0]
        print("   row", r, "cnt", wc[r], "cols", cols[:12], "gpu", gd[r, cols[:12]].round(5), "ora", wd[r, cols[:12]].round(5))
        print("   sum gpu/ora of unnormalised proxy:", gd[r].sum(), wd[r].sum(), "kp", kp[r])
        # nearest neighbour distance of this keypoint
        o = np.searchsorted(nb["kp_off"], r, side="right") - 1
        P = xyz[nb["pt_off"][o]:nb["pt_off"][o+1]]
        d = np.linalg.norm(P - kp[r], axis=1)
        print("   min dist", d.min(), "n within r", (d < cfg.radius).sum())
        ratio = gd[r] / np.where(wd[r] > 0, wd[r], np.nan)
        med = np.nanmedian(ratio)
        odd = np.nonzero(np.abs(ratio - med) > 1e-5)[0]
        print("   median ratio", med, "odd cols", odd, "gpu", gd[r, odd], "ora", wd[r, odd], "ratios", ratio[odd])
        nz_g = np.nonzero((gd[r] > 0) != (wd[r] > 0))[0]
        print("   support mismatch cols", nz_g, gd[r, nz_g], wd[r, nz_g])
        # the coincident point: its normal and lrf
        j = np.argmin(d); print("   coincident normal", nrm[nb["pt_off"][o] + j], "lrf z", gl[r, 6:9], "d2", ((P[j]-kp[r])**2).sum())
