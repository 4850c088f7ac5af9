import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ora = ge.load_oracle()
from test_gpu_parity import _steep_spectrum_data, _bare_cb, T
dev = torch.device('cuda:0')
ctx = pkg.capi.Ctx(0)
rng = np.random.default_rng(99)
words, q = _steep_spectrum_data(rng, 8192 + 100, 5000, 352)
words[4000:4003] = words[17]
q[:8] = words[:8]; q[8] = words[17]
q[9] *= 40.0
q[10] *= 1e-6
cb = _bare_cb(pkg, ctx, words)
print('stage1 dims', cb.stage1_dims, cb.stage1_energy)
ctx.timers_enable(True)
for k in (1, 2):
    idx, dist = pkg.capi.knn(ctx, cb, 0, T(q, dev), k)
    gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
    n2 = int(ctx.timer("knn_stage2_queries")[0])
    widx, wdist = ora.knn(0, words, q, k)
    bad = np.nonzero((gi != widx).any(1) | (gd != wdist).any(1))[0]
    print('k', k, 'stage2', n2, 'bad', len(bad), bad[:20])
    for b in bad[:10]:
        print('  q', b, 'got', gi[b], gd[b], 'want', widx[b], wdist[b], '|q|', np.linalg.norm(q[b]))
