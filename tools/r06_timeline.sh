mkdir -p gpurun_out/r06/tl; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in cfg3 gnn cfg3_yaml; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06/tl/$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r06/tl/$w.log 2>&1
done
cd $R
for w in cfg3 gnn cfg3_yaml; do find gpurun_out/r06/tl/$w -name "*agent_info.csv" -delete; gzip -f $(find gpurun_out/r06/tl/$w -name "*kernel_trace.csv"); done
ls -la gpurun_out/r06/tl/*/*/
