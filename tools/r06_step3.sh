#!/bin/bash
# round 6, step 3: library operators, closed-form chain, replayed closed-form steps
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_library_ops.py tests/test_gpu_rollout.py tests/test_gpu_full_size.py -x -q -m gpu \
  -k "library or opcheck or compile or closed_form or graph_replay or million" > $O/step3_pytest.log 2>&1
echo "pytest rc $?" >> $O/step3_pytest.log
tail -12 $O/step3_pytest.log
for w in echelon_stock base_stock base_stock_1m; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench3_${w}.json 2> $O/bench3_${w}.err
  echo "bench $w rc $?"; python tools/show_bench.py $O/bench3_${w}.json 2>/dev/null | head -5
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $O/bench3_${w}_eager.json 2> $O/bench3_${w}_eager.err
  python tools/show_bench.py $O/bench3_${w}_eager.json 2>/dev/null | head -3
done
