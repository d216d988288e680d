"""Micro-benchmark of the policy GEMM kernels at BASELINE cfg3 shapes (HIP events on the current stream).
NIC_GEMM_VARIANT=1|2|3 selects an A/B variant of the dispatch (see dispatch_wx in csrc/linear_mfma.hip).  The product
library has no such switch: when the variable is set this tool compiles ITS OWN copy of the library with -DNIC_TUNING_BUILD
into tools/_build/ and binds the ops to that copy for the duration of the probe."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_inventory_control_amd import _lib, ops
from neural_inventory_control_amd.layout import pad_ld


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def _tuning_library():
    import subprocess
    from neural_inventory_control_amd import build as nb
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build")
    os.makedirs(out_dir, exist_ok=True)
    defines = os.environ.get("NIC_TUNING_DEFINES", "").split()   # e.g. "-DNIC_GNN_WAVES=8": one library per set of defines
    tag = "".join(c if c.isalnum() else "_" for c in "".join(defines))
    out = os.path.join(out_dir, f"libnic_hip_tuning{tag}.so")
    srcs = [os.path.join(nb.CSRC, s) for s, _ in nb.SOURCES]
    key = nb.source_id() + "-tuning" + tag   # (a copy built in the container travels to the GPU box: reused while the sources match)
    try:
        if os.path.isfile(out) and open(out + ".key").read().strip() == key:
            return out
    except OSError:
        pass
    # one object per source with the product's own per-file flags (+ -DNIC_TUNING_BUILD), eight compilers at a time, then one link
    from concurrent.futures import ThreadPoolExecutor
    objs = [os.path.join(out_dir, s_.replace(".hip", f".tuning{tag}.o")) for s_, _ in nb.SOURCES]

    def compile_one(job):
        (src, extra), obj = job
        subprocess.check_call([nb._hipcc()] + nb.BASE_FLAGS + ["-DNIC_TUNING_BUILD", "-ffp-contract=off"] + defines + list(extra) +
                              ["-c", os.path.join(nb.CSRC, src), "-o", obj])
    with ThreadPoolExecutor(8) as pool:
        list(pool.map(compile_one, zip(nb.SOURCES, objs)))
    subprocess.check_call([nb._hipcc(), f"--offload-arch={nb.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    with open(out + ".key", "w") as f:
        f.write(key)
    return out


def main():
    if os.environ.get("NIC_GEMM_VARIANT"):
        _lib._lib = _lib.load_library(_tuning_library())
    dev = "cuda"
    B = int(os.environ.get("PROBE_B", 65536))
    shapes = [(512, 512)] if os.environ.get("PROBE_MAIN_ONLY") else [(512, 512), (512, 51), (17, 512)]
    ldb = pad_ld(B)
    res = {"variant": os.environ.get("NIC_GEMM_VARIANT", "0")}
    for (N, K) in shapes:
        W = torch.randn(N, (K + 31) // 32 * 32, device=dev)[:, :K] * 0.05
        Wt = torch.randn(K, (N + 31) // 32 * 32, device=dev)[:, :N] * 0.05
        b = torch.randn(N, device=dev)
        X = torch.randn(K, ldb, device=dev)
        Y = torch.zeros(N, ldb, device=dev)
        dX = torch.zeros(K, ldb, device=dev)
        flops = 2.0 * N * K * B
        ms = timeit(lambda: ops.linear_fwd(W, b, X, Y, B, 1))
        res[f"fwd_{N}x{K}"] = dict(ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1))
        ms = timeit(lambda: ops.linear_dgrad(Wt, Y, X, dX, B, 1, False))
        res[f"dgrad_{N}x{K}"] = dict(ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1))
        splits = ops.wgrad_num_splits(N, K, B)
        slab = torch.zeros(splits, N, (K + 4) // 4 * 4, device=dev)
        ms = timeit(lambda: ops.linear_wgrad(Y, X, slab, B))
        res[f"wgrad_{N}x{K}"] = dict(ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1), splits=splits)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
