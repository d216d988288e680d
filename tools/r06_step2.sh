#!/bin/bash
# round 6, step 2: period backward with DMA staging + prefetched slab values; forward with spilled edge tiles
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_route_table.py -x -q -m gpu \
  -k "gnn or route_table" > $O/step2_pytest.log 2>&1
echo "pytest rc $?" >> $O/step2_pytest.log
tail -8 $O/step2_pytest.log
for w in gnn gnn_many_warehouses; do
  timeout 300 python tools/gnn_period_bwd_probe.py --workload $w --periods 6 --out $O/gnn_period_bwd_stamps2_$w.json > $O/probe2_$w.log 2>&1
  echo "probe $w rc $?"; grep -A 12 stage_us_slowest $O/probe2_$w.log | head -14
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench2_${w}.json 2> $O/bench2_${w}.err
  echo "bench $w rc $?"; python tools/show_bench.py $O/bench2_${w}.json 2>/dev/null | head -12
done
timeout 300 python bench.py --workload gnn --steps 5 --warmup 2 --no-cpu-baseline --gnn-bwd off > $O/bench2_gnn_bwd_off.json 2> $O/bench2_gnn_bwd_off.err
python tools/show_bench.py $O/bench2_gnn_bwd_off.json | head -3
timeout 300 python bench.py --workload gnn --eval --steps 5 --warmup 2 --no-cpu-baseline > $O/bench2_gnn_eval.json 2>/dev/null; python tools/show_bench.py $O/bench2_gnn_eval.json | head -3
timeout 300 python bench.py --workload gnn_many_warehouses --eval --steps 5 --warmup 2 --no-cpu-baseline > $O/bench2_gnn_mw_eval.json 2>/dev/null; python tools/show_bench.py $O/bench2_gnn_mw_eval.json | head -4
