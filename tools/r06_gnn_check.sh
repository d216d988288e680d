mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q -k "gnn or route or library or trace" > gpurun_out/r06/gnn_tests.log 2>&1; tail -3 gpurun_out/r06/gnn_tests.log
for w in gnn gnn_many_warehouses gnn_yaml; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06/g_$w.json 2> gpurun_out/r06/g_$w.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06/g_$w.json')); print('$w', d['ms_per_step'], d['roofline']['frac'], d['roofline']['mean_launch_ms'])
PY
done
