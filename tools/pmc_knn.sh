#!/bin/bash
# FETCH_SIZE of the ring kernel for a few (T, splits) settings on real descriptors (development tool; run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for v in ${PMC_KNN_VARIANTS:-"" "ISMHIP_KNN_T=2" "ISMHIP_KNN_T=2_ISMHIP_KNN_SPLITS=4"}; do
  v=${v//_ISMHIP/ ISMHIP}
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_knn_$i -- python3 tools/exp_variants.py --objects 256 --reps 1 "$v" > gpurun_out/pmc_knn_$i.log 2>&1
  echo "== variant [$v]"; tail -1 gpurun_out/pmc_knn_$i.log
  python3 tools/pmc_summary.py gpurun_out/pmc_knn_$i k_knn_l2_ring16
done
