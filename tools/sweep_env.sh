#!/bin/bash
# A/B sweep over environment settings on the headline bench: each argument is a quoted "VAR=val VAR=val" set
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-e2e --cpu-objects 0 > gpurun_out/sw_$i.json 2> gpurun_out/sw_$i.err || { echo "[$e] failed"; tail -3 gpurun_out/sw_$i.err; continue; }
  python - <<PY
import json; d=json.load(open("gpurun_out/sw_$i.json")); k=d["kernel_ms_per_step"]
print("[$e]", d["value"], {x: k.get(x) for x in ("knn","knn_rotate","knn_l2_mfma","knn_rerank","knn_stage2")}, d["knn_exact_fallback_last_launch"]["stage2_queries"])
PY
done
