bash tools/sweep_env.sh "ISMHIP_XCD_MAP=1" "ISMHIP_XCD_MAP=0" "ISMHIP_XCD_MAP=1" "ISMHIP_XCD_MAP=0"
for i in 1 2 3 4; do python - <<PY
import json; d=json.load(open("gpurun_out/sw_$i.json")); k=d["kernel_ms_per_step"]; print($i, d["ms_per_step"], {x: k.get(x) for x in ("grid","lrf","shot352","maxima")})
PY
done
