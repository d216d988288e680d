#!/bin/bash
# A/B of the GNN period kernel on the GPU box: parity tests, stamp probe (train + eval), bench on / off.   usage: tools/gnn_period_ab.sh <outdir>
out=${1:-gpurun_out/gnn_ab}
mkdir -p $out
python -m pytest tests/test_gpu_rollout.py -x -q -k "gnn" 2>&1 | tail -4
python tools/gnn_period_probe.py --out $out/probe_train.json > $out/probe_train.log 2>&1
python tools/gnn_period_probe.py --eval --out $out/probe_eval.json > $out/probe_eval.log 2>&1
for m in on off; do python bench.py --workload gnn --steps 3 --warmup 2 --gnn-period $m > $out/bench_gnn_$m.json 2> $out/bench_gnn_$m.err; done
python - <<PY
import json
for f in ("probe_train", "probe_eval"):
    try:
        j = json.load(open("$out/%s.json" % f))
        print(f, j["period_kernel_fwd_us_per_period"], "vs", j["per_mlp_launches_fwd_us_per_period"], j["stage_us_slowest_wave"])
    except Exception as e:
        print(f, "failed", e)
for m in ("on", "off"):
    try:
        j = json.loads(open("$out/bench_gnn_%s.json" % m).read().strip().splitlines()[-1]); print(m, round(j["ms_per_step"], 3))
    except Exception as e:
        print(m, "failed", e)
PY
