"""CPU model of the kNN stage-1 proof on oracle SHOT-352 descriptors of the bench's own generator. numpy + the oracle; no GPU.

  python tools/stage1_proof_model.py [train_objects=100] [queries=300]

Prints (a) how many codebook rows have a 128-coordinate partial distance below the best full distance, (b) the share of queries whose
proof fails for several slot structures -- a slot keeps its T best partial scores, its bound is the best score it dropped, a query is proven
when the lower bound of every slot's bound clears the best full distance among the kept rows (margin: 2e-3 on the distance, the size of
the rotation + f16 rounding terms of the real bound) -- and (c) the same for three row orders of the codebook: object-major, class-major (what
the bench trains: the reference iterates a std::map by class) and a fixed pseudo-random permutation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
ora = ge.load_oracle()
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def descriptors(ds, ids):
    out = []
    for i in ids:
        o = ds.get(i)
        x, y, z = (o["xyz"][:, j].copy() for j in range(3))
        nx, ny, nz = (o["normals"][:, j].copy() for j in range(3))
        kx, ky, kz = (o["kp"][:, j].copy() for j in range(3))
        po, ko = [0, len(x)], [0, len(kx)]
        lrf = ora.shot_lrf(po, x, y, z, ko, kx, ky, kz, 0.3)
        d = ora.shot352(po, x, y, z, nx, ny, nz, ko, kx, ky, kz, lrf, 0.4)
        d = d[0] if isinstance(d, tuple) else d
        out.append(d[~np.isnan(d).any(1)])
    return np.concatenate(out)


t0 = time.time()
W = descriptors(pkg.synthetic.Dataset(10, n_train, split=0, n_points=16384, n_keypoints=1024), range(n_train)).astype(np.float64)
Q = descriptors(pkg.synthetic.Dataset(10, 4, split=1, n_points=16384, n_keypoints=1024), range(1)).astype(np.float64)[:n_q]
print(f"{len(W)} words, {len(Q)} queries ({time.time() - t0:.0f} s of oracle)", flush=True)
w, V = np.linalg.eigh(W.T @ W)
V, w = V[:, ::-1], w[::-1]
M = 128
print("second moment in the leading 128 / 160 / 256 coordinates:", [round(float(w[:m].sum() / w.sum()), 4) for m in (128, 160, 256)])


def sq(A, B):
    return (A * A).sum(1)[:, None] + (B * B).sum(1)[None] - 2 * A @ B.T


D_full, D_part = sq(Q, W), sq((Q @ V)[:, :M], (W @ V)[:, :M])
best = D_full.min(1)
n_below = (D_part < best[:, None]).sum(1)
print(f"(a) rows with partial distance below the best full distance: mean {n_below.mean():.2f}, median {np.median(n_below):.0f}, "
      f"90th percentile {np.percentile(n_below, 90):.0f}, 99th {np.percentile(n_below, 99):.0f}")
n_words = len(W)


def fail_rate(dpart, dfull, n_splits, lane_slots, T, margin=2e-3):
    rows = np.arange(n_words)
    slot = (rows * n_splits // n_words) * lane_slots + (rows // 4) % lane_slots        # contiguous splits, a modulo pattern of lane slots
    ns = n_splits * lane_slots
    order = np.argsort(slot, kind="stable")
    edge = np.searchsorted(slot[order], np.arange(ns + 1))
    fails = 0
    for qi in range(len(dpart)):
        kept_best, bound = np.inf, np.inf
        for s in range(ns):
            idx = order[edge[s]:edge[s + 1]]
            dp = dpart[qi][idx]
            o = np.argpartition(dp, T)[:T + 1]
            o = o[np.argsort(dp[o])]
            kept_best = min(kept_best, dfull[qi][idx[o[:T]]].min())
            bound = min(bound, dp[o[T]])
        fails += max(np.sqrt(max(bound, 0.0)) - margin, 0.0) ** 2 < kept_best
    return 100.0 * fails / len(dpart)


print("(b) proof failures [%] at T = 2 / 3 / 4 (object-major rows):")
for ns, ls in ((2, 8), (2, 4), (2, 2), (2, 1), (4, 8)):
    print(f"    {ns} splits x {ls} lane slots:", [float(round(fail_rate(D_part, D_full, ns, ls, T), 1)) for T in (2, 3, 4)], flush=True)
obj = np.arange(n_words) // 1024
orders = {"object-major": np.arange(n_words), "class-major": np.argsort((obj % 10) * 100000 + obj, kind="stable"),
          "permuted": np.random.default_rng(1).permutation(n_words)}
print("(c) 2 splits x 8 lane slots, T = 2 / 3 / 4, by row order of the codebook:")
for name, o in orders.items():
    print(f"    {name}:", [float(round(fail_rate(D_part[:, o], D_full[:, o], 2, 8, T), 1)) for T in (2, 3, 4)], flush=True)
