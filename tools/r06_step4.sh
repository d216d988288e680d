#!/bin/bash
# round 6, step 4: the whole GPU suite after the pruning + the library tests
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/step4_pytest_all.log 2>&1
echo "pytest rc $?" >> $O/step4_pytest_all.log
tail -15 $O/step4_pytest_all.log
timeout 300 python bench.py --workload base_stock_1m --steps 20 --warmup 3 --no-cpu-baseline > $O/bench4_base_stock_1m.json 2> $O/bench4_base_stock_1m.err
python tools/show_bench.py $O/bench4_base_stock_1m.json | head -4
