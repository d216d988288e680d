#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_full_size.py -q -m gpu --timeout 600 -k "gnn" > $O/step8_pytest.log 2>&1
grep -E "^FAILED|passed|failed" $O/step8_pytest.log | tail -5
for w in gnn gnn_many_warehouses; do
  timeout 300 python tools/gnn_period_bwd_probe.py --workload $w --periods 6 --out $O/step8_stamps_$w.json > $O/step8_probe_$w.log 2>&1
  grep -A 12 stage_us_slowest $O/step8_probe_$w.log | head -14
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/step8_bench_${w}.json 2>/dev/null
  python tools/show_bench.py $O/step8_bench_${w}.json | head -5
done
timeout 600 python tools/collect_profiles.py step8 traffic:gnn > $O/step8_traffic.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/step8/step8_traffic_gnn.json'))
for k in d["kernels"]: print(k["label"], round(k["hbm_bytes_per_launch"]/1e6,1),"MB", k["FETCH_SIZE_KiB"], k["WRITE_SIZE_KiB"])
PY
