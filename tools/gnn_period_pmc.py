"""SQ counters of the GNN period kernel (csrc/gnn_period.hip): two rocprofv3 --pmc passes (8 SQ slots each) over a short bench run,
summed per kernel.  Writes <out>.json.   python tools/gnn_period_pmc.py <out.json> [--eval]"""
import csv
import glob
import json
import os
import subprocess
import sys

os.environ.setdefault("TMPDIR", "/tmp")
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
     "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_SCA",
     "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_ACTIVE_INST_VMEM"],
]


def main():
    out = sys.argv[1]
    extra = sys.argv[2:]
    res = {"command": "bench.py --workload gnn --steps 1 --warmup 1 --periods 10 --no-dist-init --no-kernel-timing " + " ".join(extra), "kernels": {}}
    for i, counters in enumerate(PASSES):
        d = f"/tmp/gnn_pmc_{i}"
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", "bench.py",
               "--workload", "gnn", "--steps", "1", "--warmup", "1", "--periods", "10", "--no-dist-init", "--no-kernel-timing"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            res.setdefault("errors", []).append(p.stderr[-1500:])
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "gnn_period" not in k and "mlp3" not in k:
                    continue
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                e = res["kernels"].setdefault(k, {})
                c = row["Counter_Name"]
                e[c] = e.get(c, 0.0) + float(row["Counter_Value"])
                e.setdefault("_dispatches", set()).add(row["Dispatch_Id"])
    for e in res["kernels"].values():
        e["dispatches"] = len(e.pop("_dispatches"))
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res["kernels"].items():
        if "gnn_period" in k:
            print(k, json.dumps(e, indent=1))


if __name__ == "__main__":
    main()
