#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_full_size.py -q -m gpu --timeout 900 -k "gnn" > $O/step7_pytest.log 2>&1
echo "pytest rc $?" >> $O/step7_pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/step7_pytest.log | tail -12
for w in gnn gnn_many_warehouses; do
  timeout 300 python tools/gnn_period_bwd_probe.py --workload $w --periods 6 --out $O/gnn_period_bwd_stamps3_$w.json > $O/probe3_$w.log 2>&1
  echo "probe $w rc $?"; grep -A 12 stage_us_slowest $O/probe3_$w.log | head -14
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench7_${w}.json 2> $O/bench7_${w}.err
  echo "bench $w rc $?"; python tools/show_bench.py $O/bench7_${w}.json 2>/dev/null | head -8
done
