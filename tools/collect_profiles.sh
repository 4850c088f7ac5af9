#!/bin/bash
# Collects the judged profiles of a bench command on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh            headline config: stats + FETCH/WRITE + two SQ counter passes
#   bash tools/collect_profiles.sh 3          --config 3:      stats + FETCH/WRITE passes      (-> gpurun_out/*_cfg3)
#   1. rocprofv3 --kernel-trace --stats        -> gpurun_out/prof_final<sfx>/
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE     -> gpurun_out/pmc_fetch_final<sfx>/, pmc_write_final<sfx>/   (separate passes, as the guide prescribes)
#   3. two SQ counter passes (headline only)   -> gpurun_out/pmc_sq_final{A,B}/
# Afterwards (locally): python tools/make_profiles.py roundN [config]   copies / condenses them into profiles/roundN_[cfgN_]*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
set -e
CFG=${1:-1}
SFX=""; CARG=""
if [ "$CFG" != "1" ]; then SFX="_cfg$CFG"; CARG="--config $CFG"; fi
B="--cpu-objects 0 --no-e2e $CARG"
rm -rf gpurun_out/prof_final$SFX gpurun_out/pmc_fetch_final$SFX gpurun_out/pmc_write_final$SFX
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final$SFX -o bench -- python3 bench.py $B --steps 3 --warmup 1 > gpurun_out/prof_final$SFX.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_final$SFX -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_fetch_final$SFX.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_final$SFX -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_write_final$SFX.log 2>&1
echo "write pass done"
if [ "$CFG" = "1" ]; then
  rm -rf gpurun_out/pmc_sq_finalA gpurun_out/pmc_sq_finalB
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_finalA -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_sq_finalA.log 2>&1
  echo "sq A pass done"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_sq_finalB -- python3 bench.py $B --steps 2 --warmup 1 > gpurun_out/pmc_sq_finalB.log 2>&1
  echo "sq B pass done"
fi
# keep what travels back small: the per-dispatch CSVs of the counter passes are condensed on the box and the raw ones removed
python3 tools/make_profiles.py --condense gpurun_out $CFG
find gpurun_out/pmc_fetch_final$SFX gpurun_out/pmc_write_final$SFX gpurun_out/pmc_sq_finalA gpurun_out/pmc_sq_finalB -name "*counter_collection.csv" -delete 2>/dev/null || true
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null || true
ls gpurun_out/prof_final$SFX | head
