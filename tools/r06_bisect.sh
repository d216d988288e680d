#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
for combo in "on off" "off on" "on on"; do
  set -- $combo
  timeout 200 python bench.py --workload gnn_many_warehouses --steps 2 --warmup 1 --no-cpu-baseline --scenarios 2048 --periods 12 --no-dist-init --gnn-period $1 --gnn-bwd $2 > $O/bisect_$1_$2.json 2> $O/bisect_$1_$2.err
  echo "period=$1 bwd=$2 rc $?"; tail -2 $O/bisect_$1_$2.err | cut -c1-300
done
timeout 200 python bench.py --workload gnn_many_warehouses --steps 2 --warmup 1 --no-cpu-baseline --scenarios 2048 --periods 12 --no-dist-init --no-kernel-timing > $O/bisect_notimer.json 2> $O/bisect_notimer.err
echo "no timer rc $?"; tail -2 $O/bisect_notimer.err | cut -c1-300
