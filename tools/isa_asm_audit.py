#!/usr/bin/env python
"""Audit of hand-counted LDS reads in compiled kernels (cdna_hip_programming.md §5.7 item 1).

The k loops of csrc/linear_mfma.hip read their MFMA fragments with inline-asm `ds_read_*` statements whose completion hipcc does
not track: the destination registers count as written when the statement ends, so the compiler is free to copy, spill or reuse
them before the data has landed.  This tool compiles a HIP source to gfx950 assembly and, kernel by kernel, follows every
asm-issued ds_read until an asm `s_waitcnt lgkmcnt(N)` retires it (LDS reads return in order: a wait with count N retires all
but the N youngest) and reports any instruction OUTSIDE an asm statement that reads or writes a still-pending register, plus any
pending read at a branch / end of a basic block that leaves the kernel's straight-line tile body.

    python tools/isa_asm_audit.py neural_inventory_control_amd/csrc/linear_mfma.hip [-D...]

Exit code 1 if a violation is found.  tests/test_isa_audit.py runs it on every build.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def compile_to_asm(src, extra):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.dirname(os.path.abspath(src)), "-o", out, src] + list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def audit(asm_path):
    """-> (kernels seen, asm reads followed, [violation strings])"""
    kernel, in_asm, pending, problems, n_reads, kernels = None, False, [], [], 0, 0
    for ln, raw in enumerate(open(asm_path), 1):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.startswith("_Z") and raw.rstrip().endswith(":") or (raw.startswith("_Z") and ":" in raw and raw.split(":")[0].isidentifier()):
            kernel, pending = raw.split(":")[0], []
            kernels += 1
            continue
        if kernel is None or not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith("s_endpgm"):
            if pending:
                problems.append(f"{kernel}:{ln}: {len(pending)} asm LDS read(s) still pending at s_endpgm")
            kernel = None
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        op = line.split()[0]
        if in_asm:
            if op.startswith("ds_read"):
                dst = line.split()[1].rstrip(",")
                pending.append(regs_of(dst))
                n_reads += 1
            elif op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", line)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
            continue
        if not pending:
            continue
        if op == "s_waitcnt":   # a compiler-inserted lgkmcnt wait also retires reads (it can only help)
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            continue
        touched = regs_of(line)
        busy = set().union(*pending)
        hit = touched & busy
        if hit:
            problems.append(f"{kernel}:{ln}: `{line}` touches {sorted(hit)[:4]} while an asm ds_read into it is in flight")
    return kernels, n_reads, problems


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src, extra = sys.argv[1], sys.argv[2:]
    asm = src if src.endswith(".s") else compile_to_asm(src, extra)
    kernels, n_reads, problems = audit(asm)
    print(f"{os.path.basename(src)}: {kernels} kernels, {n_reads} asm LDS reads followed, {len(problems)} violation(s)")
    for p in problems[:40]:
        print("  " + p)
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
