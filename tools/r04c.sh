mkdir -p gpurun_out/r04c
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04c/pytest.txt
cp gpurun_out/grad_parity.json gpurun_out/r04c/ 2>/dev/null
for spec in "cfg3 65536 100" "cfg3 8192 100" "cfg3 1024 50" "cfg5 32768 70" "cfg5 4096 70" "cfg5 1024 50"; do
  set -- $spec
  python bench.py --workload $1 --scenarios $2 --periods $3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04c/${1}_${2}_${3}.json 2> gpurun_out/r04c/${1}_${2}_${3}.err
done
python bench.py --workload cfg3 --scenarios 1024 --periods 50 --steps 5 --warmup 2 --no-cpu-baseline --graph > gpurun_out/r04c/cfg3_1024_50_graph.json 2>/dev/null
python bench.py --workload cfg3 --scenarios 8192 --periods 100 --steps 5 --warmup 2 --no-cpu-baseline --graph > gpurun_out/r04c/cfg3_8192_100_graph.json 2>/dev/null
