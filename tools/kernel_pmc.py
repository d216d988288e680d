"""SQ counters of the kernels of one bench workload: two rocprofv3 --pmc passes (8 SQ slots each) over a short bench run, summed per
kernel whose name contains one of the given substrings.  Writes <out>.json.
    python tools/kernel_pmc.py <out.json> <workload> <kernel substring>[,<substring>...] [bench.py arguments ...]"""
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
    out, workload = sys.argv[1], sys.argv[2]
    global MATCH
    MATCH = sys.argv[3].split(",")
    extra = sys.argv[4:]
    res = {"command": f"bench.py --workload {workload} --steps 2 --warmup 1 --no-dist-init --no-kernel-timing --no-cpu-baseline " + " ".join(extra), "kernels": {}}
    for i, counters in enumerate(PASSES):
        d = f"/tmp/gnn_pmc_{i}"
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", "bench.py",
               "--workload", workload, "--steps", "2", "--warmup", "1", "--no-dist-init", "--no-kernel-timing", "--no-cpu-baseline"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            res.setdefault("errors", []).append(p.stderr[-1500:])
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if not any(m in k for m in MATCH):
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
        if True:
            print(k, json.dumps(e, indent=1))


if __name__ == "__main__":
    main()
