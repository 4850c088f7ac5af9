cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for x in 1 0; do
  export ISMHIP_XCD_MAP=$x
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xcd$x -o run -- python3 bench.py --config 4 --objects 823 --steps 2 --warmup 0 --cpu-objects 0 --no-e2e > gpurun_out/xcd$x.json 2> gpurun_out/xcd$x.err || { tail -5 gpurun_out/xcd$x.err; exit 1; }
  f=$(find gpurun_out/xcd$x -name "*kernel_stats.csv" | head -1)
  echo "XCD_MAP=$x $f"
  if [ -n "$f" ]; then grep -E "k_spfh|k_fpfh_sum|k_fpfh_mark|k_shot|k_lrf_cov" "$f" | cut -c1-60,100-220; fi
  find gpurun_out/xcd$x -name "*kernel_trace.csv" -delete
done
