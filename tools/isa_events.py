#!/usr/bin/env python3
"""Memory-event sequence of a compiled kernel: compiles one csrc/*.hip file for gfx950 (hipcc -save-temps) and prints, per kernel
whose mangled name contains FILTER, the order of vector-memory loads (L), stores (S), LDS ops (d), `s_waitcnt vmcnt(N)` (WN), MFMAs
(M) and branches (b), plus counts of SGPR-spill traffic (v_readlane / v_writelane) and scratch.  This is how the serialized
`load, s_waitcnt vmcnt(0), use` loops and the SGPR-spill select chains of round 2 were found.

    python3 tools/isa_events.py neural_inventory_control_amd/csrc/mlp3.hip mlp3_fwd_kernelILi48 [-ffp-contract=off]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile


def main():
    src, filt, extra = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3:]
    tmp = tempfile.mkdtemp()
    obj = os.path.join(tmp, "k.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.abspath(src), "-o", obj,
                           "-save-temps=obj"] + extra, cwd=os.path.dirname(os.path.abspath(src)), stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(tmp, asm)).read()
    for name in re.findall(r"^(_Z[\w]+):", s, re.M):
        if filt not in name:
            continue
        a = s.index("\n" + name + ":") + 1
        b = s.find(".end_amdhsa_kernel", a)
        if b < 0:
            continue
        body = [l.strip() for l in s[a:b].split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        c = collections.Counter(l.split()[0] for l in body)
        seq = []
        for l in body:
            if l.startswith(("global_load", "buffer_load")):
                seq.append("L")
            elif l.startswith(("global_store", "buffer_store")):
                seq.append("S")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                seq.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
            elif l.startswith("v_mfma"):
                seq.append("M")
            elif l.startswith(("s_cbranch", "s_branch")):
                seq.append("b")
        print(f"== {name}\n   {len(body)} instructions, v_readlane {c['v_readlane_b32']}, v_writelane {c['v_writelane_b32']}, "
              f"scratch {sum(v for k, v in c.items() if k.startswith('scratch_'))}, vmcnt(0) waits {seq.count('W0')}")
        print("   " + " ".join(seq))


if __name__ == "__main__":
    main()
