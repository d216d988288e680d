"""Builds/loads tests/hostsim (the NIC_HD kernel bodies compiled for the host with g++).  Test infrastructure."""
import ctypes as C
import os
import subprocess

from neural_inventory_control_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
OUT_DIR = os.path.join(HERE, "hostsim", "_build")
# NIC_HOSTSIM_SANITIZE=1: the sanitizer tier (tests/test_sanitizer_tier.py) - same bodies built with AddressSanitizer +
# UBSan, any report aborts the process.  The interpreter must then run with libasan preloaded (the test does that).
SANITIZE = os.environ.get("NIC_HOSTSIM_SANITIZE") == "1"
OUT = os.path.join(OUT_DIR, "libhostsim_asan.so" if SANITIZE else "libhostsim.so")
FLAGS = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
         if SANITIZE else ["-O1"])
_h = None


def asan_runtime():
    """Path of gcc's libasan.so (to LD_PRELOAD into a python that dlopens the sanitized build), or None."""
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
        return p if os.path.isabs(p) and os.path.isfile(p) else None
    except Exception:
        return None


def load():
    global _h
    if _h is not None:
        return _h
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [SRC] + [os.path.join(HERE, "..", "neural_inventory_control_amd", "csrc", f)
                    for f in ("env_step_body.h", "policy_heads_body.h", "small_rollout_body.h", "closed_form_body.h")] + [os.path.join(HERE, "..", "include", "nic_rollout.h")]
    if not os.path.isfile(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++"] + FLAGS + ["-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", OUT])
    h = C.CDLL(OUT)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    IOP = C.POINTER(_lib.NicEnvStepIO)
    h.hostsim_env_step_fwd.argtypes = [IOP, vp, vp, vp, vp]
    h.hostsim_env_step_bwd.argtypes = [IOP, vp, vp, vp, _lib.NicTable2, vp, vp, vp, vp, vp, vp]
    h.hostsim_env_step_fwd_per_store.argtypes = [IOP, vp, vp, vp]
    h.hostsim_env_step_bwd_per_store.argtypes = [IOP, vp, vp, _lib.NicTable2, vp, vp, vp, vp]
    h.hostsim_head_data_driven.argtypes = [vp] * 9 + [i32] * 5
    h.hostsim_head_warehouse_fwd.argtypes = [vp, vp, vp, f32, i32, vp, vp, i32, i32, i32, i32, i32]
    h.hostsim_head_warehouse_bwd.argtypes = [vp, vp, vp, f32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32]
    h.hostsim_head_softplus_fwd.argtypes = [vp, vp, i32, i32, i32]
    h.hostsim_head_softplus_bwd.argtypes = [vp, vp, vp, i32, i32, i32]
    h.hostsim_head_serial_fwd.argtypes = [vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32]
    h.hostsim_head_serial_bwd.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]
    SRP = C.POINTER(_lib.NicSmallRolloutDesc)
    h.hostsim_small_rollout_fwd.argtypes = [SRP, vp, vp, vp, vp, vp]
    h.hostsim_small_rollout_bwd.argtypes = [SRP, vp, vp, vp, _lib.NicTable2, vp, vp]
    h.hostsim_closed_form_rollout.argtypes = [C.POINTER(_lib.NicClosedFormDesc), vp, vp, vp, vp]
    _h = h
    return h
