set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lrf or shot or fpfh or normals or cfg" > gpurun_out/t_sel.log 2>&1 || { tail -30 gpurun_out/t_sel.log; exit 1; }
tail -2 gpurun_out/t_sel.log
timeout -k 10 300 python bench.py --cpu-objects 0 --no-e2e --steps 6 --warmup 2 > gpurun_out/b_q.json 2> gpurun_out/b_q.err || { tail gpurun_out/b_q.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b_q.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
timeout -k 10 300 python tools/fpfh_time.py > gpurun_out/fpfh_time.log 2>&1; grep dbg gpurun_out/fpfh_time.log
