"""One-off record of the CPU baseline at the GPU run's FULL batch ("same B as the GPU run when host RAM allows"):
the oracle (PyTorch-CPU eager restatement of the reference path) timed on this host's cores on BASELINE cfg3 (65,536 scenarios x 16
stores x T = 100) and cfg5's per-GPU shard (32,768 x 64 x T = 70) - training steps, fwd + bwd.  No GPU involved.  The default
`bench.py` run keeps its bounded sample (10-30 s of CPU work); it quotes this file's figure as `cpu_baseline.full_batch_value`.

    python tools/cpu_baseline_full.py [out.json] [workload ...]

The autograd graph of a full batch is tens of GB (cfg3: ~16 KB per scenario-period): the process caps its own address space at
70 % of the host's memory (a failed allocation then raises instead of taking the box down) and a workload whose estimate does
not fit runs at the largest power-of-two fraction of its batch that does (recorded as such).
"""
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EST_BYTES_PER_SCENARIO_PERIOD = {"cfg3": 18e3, "cfg5": 26e3}   # saved activations of the 512 x 3 MLP + the env step's aten ops


def meminfo():
    out = {}
    for ln in open("/proc/meminfo"):
        k, v = ln.split(":")
        out[k] = int(v.split()[0]) * 1024
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".json") else os.path.join(ROOT, "gpurun_out", "r06_cpu_baseline_full.json")
    names = [a for a in sys.argv[1:] if not a.endswith(".json")] or ["cfg3", "cfg5"]
    mi = meminfo()
    cap = int(0.7 * mi["MemTotal"])
    resource.setrlimit(resource.RLIMIT_AS, (cap, cap))
    import bench
    from neural_inventory_control_amd import workloads
    res = {"host_threads": os.cpu_count(), "mem_total_gb": round(mi["MemTotal"] / 2**30, 1), "mem_available_gb": round(mi["MemAvailable"] / 2**30, 1),
           "address_space_cap_gb": round(cap / 2**30, 1), "workloads": {}}
    for w in names:
        setting, policy, n, T, desc = workloads.get(w)
        sample = n
        while sample > 256 and EST_BYTES_PER_SCENARIO_PERIOD.get(w, 20e3) * sample * T > 0.5 * mi["MemAvailable"]:
            sample //= 2
        t0 = time.time()
        try:
            rec = bench.cpu_baseline(w, sample, T, reps=3, strict_reps=True)   # 1 warm-up + >= 3 timed repetitions, median
            rec.update(full_batch=n, scenarios=sample, periods=T, is_full_batch=sample == n, wall_s=round(time.time() - t0, 1))
        except MemoryError as e:
            rec = {"value": None, "error": f"MemoryError at {sample} scenarios: {e}", "full_batch": n, "scenarios": sample, "periods": T}
        res["workloads"][w] = rec
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        json.dump(res, open(out_path, "w"), indent=1)
        print(w, json.dumps(rec)[:400], flush=True)


if __name__ == "__main__":
    main()
