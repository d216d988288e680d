mkdir -p gpurun_out/r06/tl; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in ${WORKLOADS:-cfg1 cfg2 cfg4 base_stock real_data_driven cfg5 cfg3_shard8 gnn_many_warehouses}; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06/tl/$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r06/tl/$w.log 2>&1
  find $R/gpurun_out/r06/tl/$w -name "*agent_info.csv" -delete; gzip -f $(find $R/gpurun_out/r06/tl/$w -name "*kernel_trace.csv")
  tail -1 $R/gpurun_out/r06/tl/$w.log | cut -c1-120
done
