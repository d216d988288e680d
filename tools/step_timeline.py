"""Kernel timeline of the LAST training steps of a bench run, from a rocprofv3 kernel trace: per step the kernels by name (count, total
µs), the idle gaps between consecutive kernels, and the step's wall time on the device.  A step is delimited by a marker kernel
(substring of the kernel name that occurs once per step).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --workload W --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing
    python tools/step_timeline.py <dir> <marker> [--out file.txt] [--steps 2]
Round 6 use: found the 52 µs zero-fill + 172 µs copy of the demand trace inside the replayed closed-form step."""
import argparse
import collections
import csv
import glob
import gzip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("marker")
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--skip-last", type=int, default=0, help="ignore this many trailing marker launches (evaluation passes)")
    ap.add_argument("--list", action="store_true", help="every launch of the step in order instead of the by-name table")
    args = ap.parse_args()
    f = sorted(glob.glob(args.dir + "/**/*kernel_trace.csv*", recursive=True))[-1]
    rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if args.marker in r["Kernel_Name"]]
    if args.skip_last:
        idx = idx[:-args.skip_last]
    out = [f"{f}: {len(rows)} launches, {len(idx)} of the marker '{args.marker}'"]
    for a, b in list(zip(idx[:-1], idx[1:]))[-args.steps:]:
        t0 = int(rows[a]["Start_Timestamp"])
        wall = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
        by = collections.OrderedDict()
        busy = gaps = 0.0
        big_gaps = []
        prev_end = t0
        for r in rows[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
            c = by.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += (e - s) / 1e3
            busy += (e - s) / 1e3
            g = (s - prev_end) / 1e3
            if g > 0:
                gaps += g
            if g > 20:
                big_gaps.append((round((s - t0) / 1e3, 1), round(g, 1), name[:50]))
            if args.list:
                out.append(f"{(s - t0) / 1e3:10.1f} gap {g:7.1f} dur {(e - s) / 1e3:8.1f}  {name}")
            prev_end = max(prev_end, e)
        out.append(f"--- step: {b - a} launches, {wall:.1f} us on the device clock, kernels {busy:.1f} us, idle {gaps:.1f} us")
        for name, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            out.append(f"{us:10.1f} us {n:6d} x  {name}")
        out.append(f"gaps > 20 us (at, length, before): {big_gaps[:40]}")
    text = "\n".join(out)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
