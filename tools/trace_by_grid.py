"""Per-(kernel, grid) launch statistics from a rocprofv3 kernel trace CSV.

`rocprofv3 --stats` averages a kernel template over ALL its launches; the policy GEMMs run one template on several layer
shapes (512x512, 512x51, 17x512), so the per-shape means that bench.py's HIP-event timing reports are only comparable
after splitting the trace by grid size.  usage: python tools/trace_by_grid.py <kernel_trace.csv> <out.csv>
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def main(src, dst):
    acc = defaultdict(list)
    for r in csv.DictReader(open(src)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        n_wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
        acc[(short(r["Kernel_Name"]), n_wg, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for (name, n_wg, wg), d in rows:
            w.writerow([name, n_wg, wg, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
