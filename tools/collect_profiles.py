#!/usr/bin/env python3
"""Collects the per-round evidence under profiles/ in one go (run on the MI355X box, from the repo root):

    python3 tools/collect_profiles.py r02 [workload ...]

`python3 tools/collect_profiles.py r02 traffic:gnn traffic:cfg2 ...` instead collects the HBM counters of every kernel class
of those workloads (`<round>_traffic_<w>.json`, see traffic_pass).

For each workload: the plain bench line (`<round>_bench_<w>.json`), the `rocprofv3 --kernel-trace --stats` summary of the
same command (`<round>_bench_<w>_kernel_stats.csv`, per-(kernel, grid) split `..._kernel_by_grid.csv`) and the bench line
printed under the profiler.  For cfg3 additionally the counter traffic (`traffic:cfg3`, see below).
Everything is written to gpurun_out/<round>/ (merged back by gpurun); the caller copies what it wants judged to profiles/.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:100]


def run(cmd, out_path=None):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    if out_path:
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        open(out_path, "w").write((lines[-1] if lines else json.dumps({"error": r.stderr[-2000:]})) + "\n")
    return r


def stats_from_trace(trace_csv, stats_out, grid_out):
    by_name, by_grid = defaultdict(list), defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        n_wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
        by_name[short(r["Kernel_Name"])].append(d)
        by_grid[(short(r["Kernel_Name"]), n_wg, wg)].append(d)
    total = sum(sum(v) for v in by_name.values())
    with open(stats_out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
    with open(grid_out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for (k, n_wg, wg), v in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, n_wg, wg, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])


def note_name(k):
    """rocprofv3's spelling of a kernel (template arguments as numbers) -> the spelling the C ABI reports (nic_last_kernel),
    which is what bench.py looks up."""
    m = re.match(r"mlp3_bwd_hist_kernel<(\d+), (true|false), (\d+)>", k)
    if m:
        return "mlp3_bwd_hist_kernel<%s,%s%s>" % (m.group(1), "gather" if m.group(2) == "true" else "stored",
                                                  ",tail" if m.group(3) != "0" else "")
    shapes = {"0": "any", "1": "one_store", "2": "serial"}
    m = re.match(r"small_rollout_fwd_mfma_kernel<(\d+), (\d+)>", k)
    if m:
        return "small_rollout_fwd_mfma_kernel<%s,%s>" % (m.group(1), shapes[m.group(2)])
    m = re.match(r"small_rollout_bwd_mfma_kernel<(\d+), (true|false), (\d+)>", k)
    if m:
        return "small_rollout_bwd_mfma_kernel<%s,%s%s>" % (m.group(1), "wgrad," if m.group(2) == "true" else "", shapes[m.group(3)])
    m = re.match(r"small_rollout16_fwd_kernel<(\d+), (\d+)>", k)
    if m:
        return "small_rollout16_fwd_kernel<%s,%s>" % (m.group(1), shapes[m.group(2)])
    m = re.match(r"small_rollout16_bwd_kernel<(\d+), (\d+)>", k)
    if m:
        return "small_rollout16_bwd_kernel<%s,wgrad,%s>" % (m.group(1), shapes[m.group(2)])
    m = re.match(r"(gemm_wgrad_dma_kernel<\d+, \d+, \d+, \d+), (true|false)>", k)
    if m:
        return (m.group(1) + (",skip>" if m.group(2) == "true" else ">")).replace(" ", "")
    m = re.match(r"gnn_period_fwd_kernel<(\d+), (true|false), (\d+), (true|false)>", k)
    if m:
        return "gnn_period_fwd_kernel<%s,%s,%s%s>" % (m.group(1), m.group(2), m.group(3), ",spill" if m.group(4) == "true" else "")
    m = re.match(r"closed_form_kernel<(\d+), (\d+), (true|false), (\d+)>", k)   # (the ABI spells the store variant's WC, not a chain's 0)
    if m:
        return "closed_form_kernel<%s,%s,%s%s>" % (m.group(1), m.group(2), m.group(3), "," + m.group(4) if m.group(4) != "0" else "")
    m = re.match(r"thin_in_fwd_kernel<(\d+), (?:true|false)>", k)
    if m:
        return "thin_in_fwd_kernel<%s>" % m.group(1)
    m = re.match(r"(gemm_wx_stream_kernel<\d+, \d+), (\d)>", k)
    if m:
        return (m.group(1) + "," + {"0": "EPI_BIAS_ACT", "1": "EPI_DGRAD"}.get(m.group(2), m.group(2)) + ">").replace(" ", "")
    m = re.match(r"(gemm_wx(?:_dma)?_kernel<\d+, \d+, \d+, \d+), (\d)>", k)
    if m:
        return (m.group(1) + "," + {"0": "EPI_BIAS_ACT", "1": "EPI_DGRAD"}.get(m.group(2), m.group(2)) + ">").replace(" ", "")
    return k.replace(" ", "")


def pmc_rows(counter, out_dir, bench_args):
    """One --pmc pass; [(dispatch id, kernel, value)] in dispatch order (rows of one dispatch averaged)."""
    run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", "python3", "bench.py"]
        + bench_args)
    per = defaultdict(list)
    names = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                d = int(r["Dispatch_Id"])
                per[d].append(float(r["Counter_Value"]))
                names[d] = short(r["Kernel_Name"])
    return [(d, names[d], sum(v) / len(v)) for d, v in sorted(per.items())]


def traffic_pass(rnd, w, out):
    """HBM bytes per launch of every kernel CLASS of a workload: FETCH_SIZE / WRITE_SIZE passes joined, template by template
    and in dispatch order, to the launch sequence bench.py recorded in the same process (--launch-order-out)."""
    few = ["--periods", "8"] if w in ("cfg3", "cfg5", "gnn", "cfg3_shard8") else []
    args = ["--workload", w, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-dist-init"] + few
    order_f = os.path.join(out, f"order_{w}.json")
    fetch = pmc_rows("FETCH_SIZE", os.path.join(out, f"pmc_fetch_{w}"), args + ["--launch-order-out", order_f])
    write = pmc_rows("WRITE_SIZE", os.path.join(out, f"pmc_write_{w}"), args + ["--launch-order-out", order_f])
    doc = json.load(open(order_f))
    by_kernel = defaultdict(list)          # kernel (ABI spelling) -> class labels in launch order
    for tag, kern in doc["order"]:
        by_kernel[kern.replace(" ", "")].append(tag)
    kernels = []
    for kern, tags in sorted(by_kernel.items()):
        # (the ABI may qualify a name behind the template arguments - "gnn_period_bwd_kernel<4> (+ env / allocation adjoint)")
        same = lambda k: note_name(k) == kern or kern.startswith(note_name(k) + "(")  # noqa: E731
        f_rows = [v for _, k, v in fetch if same(k)]
        w_rows = [v for _, k, v in write if same(k)]
        if len(f_rows) < len(tags) or len(w_rows) < len(tags):
            print("traffic: no counter rows for", kern, len(f_rows), len(w_rows), len(tags), flush=True)
            continue
        f_rows, w_rows = f_rows[-len(tags):], w_rows[-len(tags):]   # launches before the timed step (set-up) come first
        acc = defaultdict(lambda: [0.0, 0.0, 0])
        for tag, fv, wv in zip(tags, f_rows, w_rows):
            a = acc[tag]
            a[0] += fv
            a[1] += wv
            a[2] += 1
        for tag, (fs, ws, cnt) in sorted(acc.items()):
            # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled; both counters in KiB
            kernels.append({"kernel": kern, "label": tag, "launches": cnt, "FETCH_SIZE_KiB": round(fs / cnt, 1),
                            "WRITE_SIZE_KiB": round(ws / cnt, 1), "hbm_bytes_per_launch": (2.0 * fs + ws) / cnt * 1024})
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py " + " ".join(args) +
                       "`; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request).  "
                       "Counter rows are joined to kernel classes (`label`) through the launch sequence the same process recorded "
                       "(--launch-order-out): templates shared by several classes are split in dispatch order.",
               "workload": w, "n_scenarios": doc["n_scenarios"], "periods": doc["periods"], "kernels": kernels},
              open(os.path.join(out, f"{rnd}_traffic_{w}.json"), "w"), indent=1)
    print("traffic", w, json.dumps(kernels)[:400], flush=True)


def main():
    rnd = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out", rnd)
    os.makedirs(out, exist_ok=True)
    if any(a.startswith("traffic:") for a in sys.argv[2:]):
        for a in sys.argv[2:]:
            traffic_pass(rnd, a.split(":", 1)[1], out)
        return
    workloads = sys.argv[2:] or ["cfg3", "cfg3_shard8", "cfg3_batch1024", "cfg3_yaml", "cfg5_yaml", "cfg2", "cfg4", "cfg5", "cfg1",
                                 "gnn", "gnn_many_warehouses", "base_stock", "base_stock_1m", "echelon_stock", "real_data_driven",
                                 "real_data_yaml", "gnn_yaml", "one_store_real_yaml", "one_store_real_transformed_nv_yaml"]
    epoch = ("cfg3_yaml", "cfg5_yaml", "real_data_yaml", "gnn_yaml", "one_store_real_yaml", "one_store_real_transformed_nv_yaml")   # a step = one batch of an epoch (8 or 4 batches): whole epochs
    out = os.path.join(ROOT, "gpurun_out", rnd)
    os.makedirs(out, exist_ok=True)
    for w in workloads:
        big = ("cfg3", "cfg5", "gnn", "gnn_many_warehouses", "cfg3_shard8")
        steps = (["--steps", "16", "--warmup", "1"] if w in epoch else
                 ["--steps", "3", "--warmup", "1"] if w in big else ["--steps", "20", "--warmup", "3"])
        run(["python3", "bench.py", "--workload", w] + steps, os.path.join(out, f"{rnd}_bench_{w}.json"))
        prof = os.path.join(out, "prof_" + w)
        psteps = (["--steps", "8", "--warmup", "1", "--no-kernel-timing"] if w in epoch else
                  ["--steps", "2", "--warmup", "1"] if w in big else ["--steps", "5", "--warmup", "2"])
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof, "--", "python3", "bench.py",
             "--workload", w, "--no-cpu-baseline", "--no-dist-init"] + psteps, os.path.join(out, f"{rnd}_bench_{w}_under_rocprof.json"))
        traces = glob.glob(os.path.join(prof, "**", "*kernel_trace.csv"), recursive=True)
        if traces:
            stats_from_trace(traces[0], os.path.join(out, f"{rnd}_bench_{w}_kernel_stats.csv"),
                             os.path.join(out, f"{rnd}_bench_{w}_kernel_by_grid.csv"))
        import shutil
        shutil.rmtree(prof, ignore_errors=True)   # (the raw traces are tens of MB; the summaries above are what is kept)
        if w in ("cfg1", "cfg2", "cfg4", "base_stock", "base_stock_1m", "echelon_stock"):
            # sub-millisecond steps: the one-rank RCCL all-reduce the default line includes (5 - 7 launches) is 10 - 30 % of them;
            # the same step without a process group, for comparison with the rounds that had none
            run(["python3", "bench.py", "--workload", w, "--no-dist-init", "--no-cpu-baseline"] + steps,
                os.path.join(out, f"{rnd}_bench_{w}_no_collective.json"))
        if w in ("base_stock", "base_stock_1m", "echelon_stock"):   # replayed by default since round 6: the eager step beside it
            run(["python3", "bench.py", "--workload", w, "--no-graph", "--no-cpu-baseline"] + steps,
                os.path.join(out, f"{rnd}_bench_{w}_eager.json"))
        print(w, open(os.path.join(out, f"{rnd}_bench_{w}.json")).read()[:300], flush=True)
    if "cfg3" in workloads:   # counter traffic of the headline workload rides along (bench.py reads it from profiles/ next time)
        traffic_pass(rnd, "cfg3", out)


if __name__ == "__main__":
    main()
