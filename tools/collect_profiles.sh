#!/bin/bash
# Collects the judged profiles of the default bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats        -> gpurun_out/prof_final/
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE     -> gpurun_out/pmc_fetch_final/, pmc_write_final/   (separate passes, as the guide prescribes)
#   3. two SQ counter passes                   -> gpurun_out/pmc_sq_final{A,B}/
# Afterwards (locally): python tools/make_profiles.py roundN   copies / condenses them into profiles/roundN_*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
set -e
B="--cpu-objects 0 --no-e2e"
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch_final gpurun_out/pmc_write_final gpurun_out/pmc_sq_finalA gpurun_out/pmc_sq_finalB
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py $B > gpurun_out/prof_final.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_final -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_fetch_final.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_final -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_write_final.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_finalA -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_sq_finalA.log 2>&1
echo "sq A pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_sq_finalB -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_sq_finalB.log 2>&1
echo "sq B pass done"
# keep what travels back small: the per-dispatch CSVs of the counter passes are condensed on the box
python3 tools/make_profiles.py --condense gpurun_out
ls gpurun_out/prof_final | head
