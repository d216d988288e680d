"""Builds libnic_hip.so (every HIP source under csrc/) for gfx950 with hipcc.  hipcc cross-compiles without a GPU.

    python -m neural_inventory_control_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libnic_hip.so")
# (source, extra flags).  The env/heads kernels are built with -ffp-contract=off so that a*b+c rounds twice like the
# reference's separate aten mul/add; the MFMA GEMMs are fma chains by construction.
SOURCES = [
    ("nic_abi.hip", []),
    ("env_step.hip", ["-ffp-contract=off"]),
    ("policy_heads.hip", ["-ffp-contract=off"]),
    ("head_env.hip", ["-ffp-contract=off"]),
    ("period_tail.hip", ["-ffp-contract=off"]),
    ("linear_mfma.hip", []),
    ("thin_layer.hip", []),
    ("sampler.hip", []),
    ("small_rollout.hip", ["-ffp-contract=off"]),
    ("small_rollout16.hip", ["-ffp-contract=off"]),
    ("small_reduce.hip", ["-ffp-contract=off"]),
    ("closed_form.hip", ["-ffp-contract=off"]),
    ("horizon_rollout.hip", ["-ffp-contract=off"]),
    ("mlp3.hip", []),
    ("gnn_alloc_env.hip", ["-ffp-contract=off"]),
]
HEADERS = ["nic_common.h", "env_step_body.h", "policy_heads_body.h", "small_rollout_body.h", "small_rollout16.h", "closed_form_body.h", "gnn_alloc_body.h", os.path.join("..", "..", "include", "nic_rollout.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isfile(c) or c == "hipcc"):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o] + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(OUT, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        import ctypes
        from ._lib import _init_torch_device_first
        _init_torch_device_first()   # (on a GPU box: torch's HIP context first, see _lib.load_library)
        ctypes.CDLL(OUT)  # fail the build on unresolved symbols
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
