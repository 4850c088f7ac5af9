#!/bin/bash
# A/B sweep of the stage-1 truncation (ISMHIP_KNN_PCA_M) on the headline bench; prints value + kNN parts per setting
for m in "$@"; do
  ISMHIP_KNN_PCA_M=$m timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-e2e --cpu-objects 0 > gpurun_out/sw_$m.json 2> gpurun_out/sw_$m.err || { echo "m=$m failed"; tail -3 gpurun_out/sw_$m.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/sw_$m.json")); k=d["kernel_ms_per_step"]
print("m=$m", d["value"], {x: k.get(x) for x in ("knn","knn_rotate","knn_l2_mfma","knn_rerank","knn_stage2","knn_fallback")}, d["knn_exact_fallback_last_launch"])
PY
done
