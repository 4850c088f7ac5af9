cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for x in 1 0; do
  export ISMHIP_XCD_MAP=$x
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/xcd$x -o run -- python3 bench.py --config 4 --objects 823 --steps 2 --warmup 0 --cpu-objects 0 --no-e2e > gpurun_out/xcd$x.json 2> gpurun_out/xcd$x.err || { tail -5 gpurun_out/xcd$x.err; exit 1; }
  f=$(find gpurun_out/xcd$x -name "*kernel_stats.csv" | head -1)
  echo "XCD_MAP=$x"; grep -E "k_spfh|k_fpfh_sum|k_fpfh_mark" $f | cut -c1-200
  find gpurun_out/xcd$x -name "*kernel_trace.csv" -delete
done
