set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -30 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
timeout -k 10 300 python bench.py --cpu-objects 0 --no-e2e --steps 6 --warmup 2 > gpurun_out/b_q.json 2> gpurun_out/b_q.err || { tail gpurun_out/b_q.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b_q.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])"
