#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 200 python bench.py --workload gnn_many_warehouses --steps 2 --warmup 1 --no-cpu-baseline --scenarios 2048 --periods 12 --no-dist-init > $O/bisect_fixed.json 2> $O/bisect_fixed.err
echo "gnn_many_warehouses 2048x12 rc $?"
for rep in 1 2; do
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_full_size.py tests/test_library_ops.py -q -m gpu --timeout 900 -k "gnn or library or opcheck or compile" > $O/step6_pytest_$rep.log 2>&1
echo "pytest rc $?" >> $O/step6_pytest_$rep.log
grep -E "^FAILED|^ERROR|passed|failed" $O/step6_pytest_$rep.log | tail -12
done
