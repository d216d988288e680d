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
    # (no SLP packing: v_pk_*_f32 beside MFMAs costs more than the two scalar instructions it replaces, MI355X_MICROARCH "price of one filler")
    ("gnn_period.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("gnn_period_bwd.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
]
# Routes that lost their A/B stay out of the default library (include/nic_experiments.h): NIC_BUILD_EXPERIMENTS=1 adds them.
EXPERIMENTS = os.environ.get("NIC_BUILD_EXPERIMENTS", "") not in ("", "0")
if EXPERIMENTS:
    SOURCES.append((os.path.join(HERE, "..", "tools", "experiments", "wide_rollout.hip"), ["-ffp-contract=off", "-I", CSRC]))
HEADERS = ["nic_common.h", "env_step_body.h", "policy_heads_body.h", "tail_pieces.h", "small_rollout_body.h", "small_rollout16.h", "closed_form_body.h", "gnn_alloc_body.h", os.path.join("..", "..", "include", "nic_rollout.h")]
if EXPERIMENTS:
    HEADERS.append(os.path.join("..", "..", "include", "nic_experiments.h"))
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isfile(c) or c == "hipcc"):
            return c
    return "hipcc"


BASE_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"]


def _sha(parts):
    import hashlib
    h = hashlib.sha256()
    for p in parts:
        h.update(p if isinstance(p, bytes) else str(p).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def source_id(csrc=CSRC):
    """Identity of what the library is built FROM: sha256 over every HIP source with its flags and every header (names +
    contents), first 16 hex digits.  `nic_build_id()` of a library built by `build()` returns this string; `_lib.load_library`
    compares the two, so a stale libnic_hip.so next to newer sources (a checkout, an edited header) is an error at load time
    instead of a silently old kernel.  Content-based: touching a file without changing it changes nothing."""
    parts = []
    for src, extra in SOURCES:
        parts += [src, " ".join(BASE_FLAGS + extra), _read(os.path.join(csrc, src))]
    for h in HEADERS:
        parts += [os.path.basename(h), _read(os.path.join(csrc, h))]
    return _sha(parts)[:16]


def _object_key(csrc, src, extra, build_id):
    """what an object file depends on: its source, every header, its flags (the ABI unit also carries the build id)"""
    parts = [src, " ".join(BASE_FLAGS + extra), _read(os.path.join(csrc, src))]
    for h in HEADERS:
        parts += [os.path.basename(h), _read(os.path.join(csrc, h))]
    if src == "nic_abi.hip":
        parts.append(build_id)
    return _sha(parts)


def _key_of(obj):
    try:
        with open(obj + ".key") as f:
            return f.read().strip()
    except OSError:
        return None


def build(force=False, verbose=True):
    """Compiles what changed (by CONTENT: every object remembers the hash of its source + headers + flags in `<obj>.key`) and
    links libnic_hip.so with the build id of the sources inside.  Returns the library path; `build.last_action` says whether
    anything was compiled ("rebuilt <id>") or the library on disk already matched the sources ("reused <id>")."""
    hipcc = _hipcc()
    bid = source_id()
    objs, compiled = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        key = _object_key(CSRC, src, extra, bid)
        if force or not os.path.isfile(o) or _key_of(o) != key:
            cmd = [hipcc] + BASE_FLAGS + ["-c", s, "-o", o] + extra
            if src == "nic_abi.hip":
                cmd.append(f'-DNIC_BUILD_ID="{bid}"')
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(o + ".key", "w") as f:
                f.write(key)
            compiled.append(src)
        objs.append(o)
    if force or compiled or not os.path.isfile(OUT) or _key_of(OUT) != bid:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(OUT + ".key", "w") as f:
            f.write(bid)
        import ctypes
        from ._lib import _init_torch_device_first
        _init_torch_device_first()   # (on a GPU box: torch's HIP context first, see _lib.load_library)
        ctypes.CDLL(OUT)  # fail the build on unresolved symbols
        build.last_action = f"rebuilt {bid}" + (f" ({len(compiled)} of {len(SOURCES)} units compiled)" if compiled else " (relinked)")
    else:
        build.last_action = f"reused {bid}"
    return OUT


build.last_action = None


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
