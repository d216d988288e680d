#!/bin/bash
# round 6 evidence, GNN workloads (bench.py lets the engine decide between eager launches and HIP-graph replay, as the Trainer does)
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
python -c "from neural_inventory_control_amd import _lib, build; print('library id', _lib.lib().nic_build_id().decode(), '= source id', build.source_id())" > $O/r06_collection_gnn_manifest.txt 2>&1
date -u >> $O/r06_collection_gnn_manifest.txt
timeout 600 python -m pytest tests/test_gpu_rollout.py -q -m gpu --timeout 600 -k "gnn" > $O/collect_gnn_pytest.log 2>&1
grep -E "^FAILED|passed|failed" $O/collect_gnn_pytest.log | tail -5
timeout 1800 python tools/collect_profiles.py r06 gnn gnn_many_warehouses gnn_yaml > $O/collect_gnn.log 2>&1
echo "collect rc $?"
for w in gnn gnn_many_warehouses; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r06_bench_${w}_eager.json 2>/dev/null
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gnn-bwd off > $O/r06_bench_${w}_per_mlp_backward.json 2>/dev/null
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --graph --gnn-bwd off --gnn-period $( [ $w = gnn ] && echo on || echo off ) > $O/r06_bench_${w}_round5_route_replayed.json 2>/dev/null
  python tools/show_bench.py $O/r06_bench_${w}.json $O/r06_bench_${w}_eager.json $O/r06_bench_${w}_per_mlp_backward.json $O/r06_bench_${w}_round5_route_replayed.json | grep -v "^   "
done
for w in gnn gnn_many_warehouses; do
  timeout 900 python bench.py --workload $w --eval --periods 5000 --steps 3 --warmup 1 --no-cpu-baseline > $O/r06_bench_${w}_eval_T5000.json 2> $O/eval_$w.err
  echo "eval $w rc $?"; python tools/show_bench.py $O/r06_bench_${w}_eval_T5000.json | head -3
done
