#!/usr/bin/env python
"""Prints ms/step and the per-kernel-class table of one or more bench.py JSON lines (files)."""
import json
import sys
for f in sys.argv[1:]:
    d = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
    print(f"{f}: {d['ms_per_step']:.2f} ms/step  {d['value']:.4g} {d['unit']}")
    for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["total_ms_per_step"]):
        print(f"   {k:22s} {v['mean_ms'] * 1e3:9.1f} us x {v['launches_per_step']:6.0f} = {v['total_ms_per_step']:8.3f} ms   "
              f"{v.get('bound', '-'):5s} {v.get('frac', float('nan')):.3f}  {v['kernel']}")
