#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_full_size.py tests/test_gpu_route_table.py tests/test_gpu_kernels.py -q -m gpu --timeout 600 -k "gnn or route_table or alloc" > $O/step12_pytest.log 2>&1
grep -E "^FAILED|passed|failed" $O/step12_pytest.log | tail -8
for w in gnn_many_warehouses gnn; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/step12_bench_${w}.json 2>/dev/null
  python tools/show_bench.py $O/step12_bench_${w}.json | head -6
done
