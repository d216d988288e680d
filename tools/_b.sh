timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rollout.py tests/test_gpu_full_size.py -x -q -m gpu -k "reduce or small or whole_horizon" 2>&1 | tail -2
for w in cfg1 cfg2 cfg4; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
  python -c "import json,sys; d=json.load(open('/tmp/l.json')); print(d['config']['name'], round(d['ms_per_step'],4), {k:v.get('mean_ms') for k,v in d['kernels'].items()})"
done
