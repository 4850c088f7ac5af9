set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -30 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/b_default.json 2> gpurun_out/b_default.err || { tail gpurun_out/b_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'], d['cpu_baseline']['value'])"
