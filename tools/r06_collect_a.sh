#!/bin/bash
# round 6 evidence, part A: every workload whose kernels are final (everything but the GNN ones), the long-horizon evaluations,
# the scaling prediction from 1-GPU shard steps and the full-batch CPU baseline
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
python -c "from neural_inventory_control_amd import _lib, build; print('library id', _lib.lib().nic_build_id().decode(), build.source_id())" > $O/r06_collection_a_manifest.txt 2>&1
date -u >> $O/r06_collection_a_manifest.txt
timeout 3000 python tools/collect_profiles.py r06 cfg3 cfg3_shard8 cfg3_batch1024 cfg3_yaml cfg5_yaml cfg2 cfg4 cfg5 cfg1 base_stock base_stock_1m echelon_stock real_data_driven real_data_yaml one_store_real_yaml one_store_real_transformed_nv_yaml > $O/collect_a.log 2>&1
echo "collect rc $?"
for w in cfg1 cfg2 cfg4; do
  timeout 600 python bench.py --workload $w --eval --periods 5000 --steps 5 --warmup 1 --no-cpu-baseline > $O/r06_bench_${w}_eval_T5000.json 2> $O/eval_$w.err
  echo "eval $w rc $?"; python tools/show_bench.py $O/r06_bench_${w}_eval_T5000.json | head -3
done
timeout 900 python tools/scaling_prediction.py --out $O/r06_scaling_prediction.json --steps 10 > $O/scaling.log 2>&1
echo "scaling rc $?"; grep -E '"gpus"|speedup' $O/r06_scaling_prediction.json | paste - - | head -8
timeout 1500 python tools/cpu_baseline_full.py $O/r06_cpu_baseline_full.json cfg3 cfg5 > $O/cpu_full.log 2>&1
echo "cpu full rc $?"; tail -2 $O/cpu_full.log | cut -c1-300
