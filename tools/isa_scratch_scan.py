#!/usr/bin/env python3
"""Where does a kernel touch scratch?  Compiles a HIP source for gfx950 to ISA (hipcc -S) and lists, per kernel that has a
private segment, every basic block that contains MFMA or scratch instructions, with the loop nest hipcc annotates.

    python3 tools/isa_scratch_scan.py neural_inventory_control_amd/csrc/linear_mfma.hip > profiles/rNN_gemm_scratch_scan.txt

The question it answers for the policy GEMMs: are the spills inside the k loop (every 256-MFMA tile would pay for them) or in the
prologue / epilogue / period bookkeeping around it?"""
import re
import subprocess
import sys
import tempfile


def main(src):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                           "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    lines = text.split("\n")
    meta = {}
    for m in re.finditer(r"\.set (\S+)\.(num_vgpr|private_seg_size), (\d+)", text):
        meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S+: ", l)]
    print(f"# {src}: kernels with a private segment (scratch), basic blocks holding MFMA / scratch instructions")
    for st in starts:
        name = lines[st].split(":")[0]
        info = meta.get(name, {})
        if not info.get("private_seg_size"):
            continue
        try:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            demangled = name
        print(f"\n{demangled}\n  VGPRs {info.get('num_vgpr')}, private segment {info['private_seg_size']} B")
        blocks, cur = [], ["entry", 0, 0, 0]
        i = st + 1
        while "s_endpgm" not in lines[i]:
            t = lines[i].strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append(cur)
                cur = [re.sub(r"\s+", " ", t), 0, 0, 0]
            elif t.startswith("v_mfma"):
                cur[1] += 1
            elif t.startswith("scratch_load"):
                cur[2] += 1
            elif t.startswith("scratch_store"):
                cur[3] += 1
            i += 1
        blocks.append(cur)
        in_loop_mfma = [b for b in blocks if b[1]]
        print(f"  {'block':60s} {'MFMA':>5s} {'scratch_load':>13s} {'scratch_store':>14s}")
        for b in blocks:
            if b[1] or b[2] or b[3]:
                print(f"  {b[0][:60]:60s} {b[1]:5d} {b[2]:13d} {b[3]:14d}")
        tot_l = sum(b[2] for b in blocks)
        tot_s = sum(b[3] for b in blocks)
        hot_l = sum(b[2] for b in in_loop_mfma)
        hot_s = sum(b[3] for b in in_loop_mfma)
        print(f"  total scratch loads / stores: {tot_l} / {tot_s}; inside blocks that issue MFMAs: {hot_l} / {hot_s} "
              f"(per {sum(b[1] for b in in_loop_mfma)} MFMAs)")


if __name__ == "__main__":
    main(sys.argv[1])
