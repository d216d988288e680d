mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/full_gpu_head2.log 2>&1; tail -3 gpurun_out/r06/full_gpu_head2.log
for w in base_stock_1m base_stock echelon_stock; do
  python bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/r06/cf_$w.json 2> gpurun_out/r06/cf_$w.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06/cf_$w.json')); print('$w', d['ms_per_step'], d['roofline']['frac'], d['roofline']['mean_launch_ms'])
PY
done
