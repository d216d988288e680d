#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 200 python bench.py --workload gnn_many_warehouses --steps 2 --warmup 1 --no-cpu-baseline --scenarios 2048 --periods 12 --no-dist-init > $O/bisect_fixed.json 2> $O/bisect_fixed.err
echo "gnn_many_warehouses 2048x12 rc $?"; tail -2 $O/bisect_fixed.err | cut -c1-300
timeout 2700 python -m pytest tests/ -x -q -m gpu > $O/step5_pytest_all.log 2>&1
echo "pytest rc $?" >> $O/step5_pytest_all.log
tail -15 $O/step5_pytest_all.log
