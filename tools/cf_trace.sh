mkdir -p gpurun_out/r06/cf_trace; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06/cf_trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload base_stock_1m --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06/cf_trace/bench.log 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06/cf_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the closed_form launches; print everything between the last two
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('closed_form_kernel')]
print(len(rows), idx[-8:])
out = []
for a, b in zip(idx[-5:-1], idx[-4:]):
    t0 = int(rows[a]['Start_Timestamp'])
    out.append(f"--- step of {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.append(f"{(s - t0) / 1e3:8.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
        prev_end = e
open('gpurun_out/r06/cf_trace/steps.txt', 'w').write("\n".join(out))
print("\n".join(out[-60:]))
PY
