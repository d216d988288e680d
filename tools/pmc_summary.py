"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel (short names)."""
import sys, re, glob
import pandas as pd

def short(n):
    m = re.search(r"(gemm_\w+_kernel<[^>]*>|env_step_\w+<\d+>|head_env_\w+<[^>]*>|head_\w+_kernel|thin_\w+_kernel<[^>]*>|wgrad_reduce_kernel|sample_demand_kernel|axpy_kernel)", n)
    if m: return m.group(1)
    return re.sub(r"<.*", "", n.replace("void ", ""))[:60]

for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        df = pd.read_csv(f)
        df["k"] = df["Kernel_Name"].map(short)
        df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
        t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
        t["dur_us"] = df.groupby("k")["dur_us"].mean()
        t["n"] = df.groupby("k")["Dispatch_Id"].nunique()
        t = t[t.index.str.contains("gemm|env_step|head_|thin_")] if len(sys.argv) > 1 and "--all" not in sys.argv else t
        pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30); pd.set_option("display.float_format", lambda x: f"{x:,.0f}")
        print(f); print(t.to_string()); print()
