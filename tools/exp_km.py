import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as ge, torch
pkg = ge.load_package()
syn = pkg.synthetic
dev = torch.device("cuda:0")
ctx = pkg.capi.Ctx(0)
train = syn.Dataset(3, 6, split=0, n_points=8192, n_keypoints=384)
test = syn.Dataset(3, 6, split=1, n_points=8192, n_keypoints=384)
order = sorted(range(6), key=lambda i: (train.label(i), i))
tb = pkg.pipeline.DeviceBatch(train.batch(order), dev)
nb = test.batch(range(6))
for cc, k, cw in [(150, 3, True), (150, 3, False), (600, 2, False), (600, 2, True), (1000, 2, False), (300, 2, False), (300, 4, False)]:
    cfg = pkg.pipeline.IsmConfig(feature="SHOT", n_classes=3, k=k, clustering="KMeansCount", cluster_count=cc, kmeans_iterations=25, kmeans_seed=4,
                                 use_class_weight=cw, use_vote_weight=True, use_matching_weight=True, max_maxima=8)
    rec = pkg.pipeline.Recognizer(ctx, cfg)
    cb = rec.train([tb])
    got = rec.detect(pkg.pipeline.DeviceBatch(nb, dev))
    print(cc, k, cw, got["cls"][:, 0].cpu().numpy(), nb["labels"], rec.kmeans_iterations, flush=True)
