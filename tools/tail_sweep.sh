#!/bin/bash
# A/B of the fused per-period tail (csrc/period_tail.hip) against the separate launches over batch sizes: bench.py cfg3, T = 20.
out=${1:-gpurun_out/tail_sweep}
mkdir -p $out
for n in 1024 4096 8192 16384 32768 65536; do
  for tail in on off; do
    timeout 300 python bench.py --workload cfg3 --scenarios $n --periods 20 --steps 4 --warmup 1 --no-cpu-baseline --tail $tail > $out/cfg3_${n}_$tail.json 2> $out/cfg3_${n}_$tail.err
    python - <<PY
import json
try:
    j=json.loads(open("$out/cfg3_${n}_$tail.json").read().strip().splitlines()[-1])
    k=j["kernels"]
    pick=lambda t: ("%.1f"%(k[t]["mean_ms"]*1e3)) if t in k else "-"
    print("n=%6d tail=%-3s ms/step=%8.3f  tail_fwd=%s tail_bwd=%s | fwd_in=%s logits=%s head_env_fwd=%s | dgrad_in=%s head_env_bwd=%s thin_bwd=%s" % ($n, "$tail", j["ms_per_step"], pick("tail_fwd"), pick("tail_bwd"), pick("fwd_512x51"), pick("fwd_17x512"), pick("head_env_fwd"), pick("dgrad_512x51"), pick("head_env_bwd"), pick("bwd_thin_17x512")))
except Exception as e:
    print("n=$n tail=$tail failed", e)
PY
  done
done
