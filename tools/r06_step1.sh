#!/bin/bash
# round 6, step 1: the period backward kernel - parity first, then stamps and the A/B on the two GNN workloads
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu \
  -k "period_backward or (gnn_fused_rollout_matches_reference and period) or train_loop_checkpoint or upstream_zero_lead" \
  > $O/step1_pytest.log 2>&1
echo "pytest rc $?" >> $O/step1_pytest.log
tail -15 $O/step1_pytest.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sampler_vs_oracle.py -x -q -m gpu -k "sampler" > $O/step1_pytest_sampler.log 2>&1
echo "pytest sampler rc $?"; tail -4 $O/step1_pytest_sampler.log
timeout 300 python bench.py --workload base_stock_1m --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_base_stock_1m.json 2> $O/bench_base_stock_1m.err
echo "bench base_stock_1m rc $?"; python tools/show_bench.py $O/bench_base_stock_1m.json | head
for w in gnn gnn_many_warehouses; do
  timeout 300 python tools/gnn_period_bwd_probe.py --workload $w --periods 6 --out $O/gnn_period_bwd_stamps_$w.json > $O/probe_$w.log 2>&1
  echo "probe $w rc $?"; tail -3 $O/probe_$w.log
  for b in on off; do
    timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --gnn-bwd $b > $O/bench_${w}_bwd_$b.json 2> $O/bench_${w}_bwd_$b.err
    echo "bench $w bwd=$b rc $?"; python tools/show_bench.py $O/bench_${w}_bwd_$b.json 2>/dev/null | head -12
  done
done
