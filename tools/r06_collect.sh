#!/bin/bash
# round 6 evidence: every workload's bench line + rocprofv3 kernel summary, HBM counter traffic of the headline and the GNN workloads,
# in-kernel stamps and SQ counters of the GNN backward kernel, the long-horizon evaluations, the scaling prediction from 1-GPU shard
# steps and the full-batch CPU baseline (>= 3 repetitions).  Everything lands in gpurun_out/r06/ and is copied to profiles/.
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
python -c "from neural_inventory_control_amd import _lib, build; print('library id', _lib.lib().nic_build_id().decode(), '= source id', build.source_id())" > $O/r06_collection_manifest.txt 2>&1
date -u >> $O/r06_collection_manifest.txt
timeout 3600 python tools/collect_profiles.py r06 > $O/collect.log 2>&1
echo "collect rc $?"
timeout 900 python tools/collect_profiles.py r06 traffic:gnn traffic:gnn_many_warehouses > $O/collect_traffic.log 2>&1
echo "traffic rc $?"
# (as run in round 6 the remaining counter passes came in a later call, with the files of the first copied to profiles/ in between:
#  bench.py reads profiles/*traffic*.json, so a line only cites a traffic file that is already there)
timeout 1500 python tools/collect_profiles.py r06 traffic:base_stock traffic:base_stock_1m traffic:echelon_stock traffic:cfg1 traffic:cfg2 \
    traffic:cfg4 traffic:cfg5 traffic:cfg3_shard8 traffic:real_data_driven >> $O/collect_traffic.log 2>&1
cp $O/r06_traffic_*.json profiles/ 2>/dev/null
for w in base_stock base_stock_1m echelon_stock cfg1 cfg2 cfg4 real_data_driven; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 > $O/r06_bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 > $O/r06_bench_cfg5.json 2>/dev/null
timeout 300 python bench.py --workload cfg3_shard8 --steps 8 --warmup 2 > $O/r06_bench_cfg3_shard8.json 2>/dev/null
for w in gnn gnn_many_warehouses; do
  timeout 300 python tools/gnn_period_bwd_probe.py --workload $w --periods 6 --out $O/r06_gnn_period_bwd_stamps_$w.json > $O/probe_final_$w.log 2>&1
  echo "probe $w rc $?"
done
timeout 600 python tools/gnn_period_pmc.py $O/r06_gnn_period_pmc.json > $O/pmc.log 2>&1
echo "pmc rc $?"
for w in gnn gnn_many_warehouses; do   # same-box A/B: eager launches with the per-kernel timer; the per-MLP backward; round 5's route replayed
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r06_bench_${w}_eager.json 2>/dev/null
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-graph --gnn-bwd off > $O/r06_bench_${w}_per_mlp_backward.json 2>/dev/null
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --graph --gnn-bwd off --gnn-period $( [ $w = gnn ] && echo on || echo off ) > $O/r06_bench_${w}_round5_route_replayed.json 2>/dev/null
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_cfg3_driver_cmd.json 2>/dev/null
for w in cfg1 cfg2 cfg4 gnn gnn_many_warehouses; do
  timeout 900 python bench.py --workload $w --eval --periods 5000 --steps 3 --warmup 1 --no-cpu-baseline > $O/r06_bench_${w}_eval_T5000.json 2> $O/eval_$w.err
  echo "eval $w rc $?"; python tools/show_bench.py $O/r06_bench_${w}_eval_T5000.json | head -3
done
WL=echelon_stock bash tools/pmc_sq_probe.sh > $O/r06_sq_echelon_chain.txt 2>&1   # (SQ counters of the closed-form chain kernel)
for w in cfg3 gnn base_stock_1m; do   # whole-step kernel timelines (launches by name, kernel vs idle time)
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 )
  python tools/step_timeline.py $O/tl_$w FusedOptimizer --steps 1 --out $O/r06_step_timeline_$w.txt > /dev/null 2>&1
  rm -rf $O/tl_$w
done
timeout 900 python tools/scaling_prediction.py --out $O/r06_scaling_prediction.json --steps 10 > $O/scaling.log 2>&1
echo "scaling rc $?"
timeout 1800 python tools/cpu_baseline_full.py $O/r06_cpu_baseline_full.json cfg3 cfg5 > $O/cpu_full.log 2>&1
echo "cpu full rc $?"; tail -2 $O/cpu_full.log | cut -c1-300
ls $O | wc -l
