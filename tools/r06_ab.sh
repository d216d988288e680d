mkdir -p gpurun_out/r06/ab
python -m pytest tests -m gpu -x -q > gpurun_out/r06/ab/full_gpu.log 2>&1; tail -2 gpurun_out/r06/ab/full_gpu.log
for w in ${WORKLOADS:-cfg3 cfg5 gnn_many_warehouses cfg3_shard8 cfg2 cfg4 cfg1 real_data_driven}; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06/ab/$w.json 2> gpurun_out/r06/ab/$w.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06/ab/$w.json')); ks=d.get('kernels') or {}
print('$w', round(d['ms_per_step'],3), ' '.join(f"{k}:{v['total_ms_per_step']:.2f}" for k,v in ks.items() if isinstance(v,dict) and v.get('total_ms_per_step',0)>0 and any(x in k for x in ('env','tail','small','horizon','thin'))))
PY
done
