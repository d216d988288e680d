"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) on tools/gemm_probe.py into profiles/<round>_traffic.json:
HBM bytes per launch for each GEMM kernel class.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, so it is doubled; both counters are in KiB.  Usage: pmc_traffic.py <fetch_csv> <write_csv> <out_json> <B>"""
import json, re, sys
import pandas as pd

def short(n):
    m = re.search(r"(gemm_wx_dma_kernel<[^>]*>|gemm_wgrad_dma_kernel<[^>]*>)", n)
    return m.group(1) if m else None

fetch, write, out, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
res = {}
for f, name in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == name]
    df["k"] = df["Kernel_Name"].map(short)
    for k, v in df.dropna(subset=["k"]).groupby("k")["Counter_Value"].mean().items():
        res.setdefault(k, {})[name] = float(v)
entries = []
# (round 4's kernels for the 512 x 512 layer at >= 16,384 scenarios; rounds 1-3: <2, 4, 4, 2, *>)
kinds = {"gemm_wx_dma_kernel<2, 4, 2, 1, 0>": "fwd", "gemm_wx_dma_kernel<2, 4, 2, 1, 1>": "dgrad",
         "gemm_wgrad_dma_kernel<2, 4, 2, 2, false>": "wgrad",
         "gemm_wx_dma_kernel<2, 4, 4, 2, 0>": "fwd", "gemm_wx_dma_kernel<2, 4, 4, 2, 1>": "dgrad", "gemm_wgrad_dma_kernel<2, 4, 4, 2>": "wgrad"}
for k, c in res.items():
    if k in kinds and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        entries.append({"kernel": k, "kind": kinds[k], "N": 512, "K": 512, "n_scenarios": B, "FETCH_SIZE_KiB": c["FETCH_SIZE"],
                        "WRITE_SIZE_KiB": c["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                        "algorithmic_bytes": {"fwd": 4.0 * (512 * 512 + 2 * 512 * B), "dgrad": 4.0 * (512 * 512 + 3 * 512 * B),
                                              "wgrad": 4.0 * (2 * 512 * B) + 2 * 4.0 * 64 * 512 * 516}[kinds[k]]})  # wgrad: + slab read/write (64 splits)
json.dump(entries, open(out, "w"), indent=1)
print(json.dumps(entries, indent=1))
