"""Times every tile shape of the LDS-DMA wx kernel (forward / dgrad) over scenario counts from the reference's shipped batch size
(1,024) to BASELINE cfg3's 65,536, next to what `pick_wx_tile` chooses by itself, and the all-period weight-gradient contraction
with the period-group split.  Calibrates `kKtileFloor` / `kFixed` of csrc/linear_mfma.hip::wx_cost.

Forcing a tile needs the tuning build (-DNIC_TUNING_BUILD reads NIC_WX_TILE at every launch); the product library has no such
switch.  Only linear_mfma.hip is recompiled, the other objects are the product's.

    python tools/gemm_tile_probe.py > gpurun_out/r04b/gemm_tile_probe.json
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import _lib, ops  # noqa: E402
from neural_inventory_control_amd import build as nb  # noqa: E402
from neural_inventory_control_amd.layout import pad_ld  # noqa: E402

# ids of csrc/linear_mfma.hip::WxTile.  (Round 4's first sweep covered seventeen tilings - profiles/r04_gemm_tile_probe*.json -
# of which these three stayed in the product; 256x256 only exists in tuning builds.)
TILES = ["128x128w8", "64x128w8", "32x128", "256x256"]
BM = {"128x128w8": 128, "64x128w8": 64, "32x128": 32, "256x256": 256}

def tuning_library():
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, "linear_mfma_tuning.o")
    out = os.path.join(out_dir, "libnic_hip_tile_probe.so")
    subprocess.check_call([nb._hipcc(), f"--offload-arch={nb.ARCH}", "-O3", "-std=c++17", "-fPIC", "-DNIC_TUNING_BUILD", "-c",
                           os.path.join(nb.CSRC, "linear_mfma.hip"), "-o", obj])
    objs = [obj if s == "linear_mfma.hip" else os.path.join(nb.CSRC, s.replace(".hip", ".o")) for s, _ in nb.SOURCES]
    subprocess.check_call([nb._hipcc(), f"--offload-arch={nb.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    return out


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    nb.build(verbose=False)
    _lib._lib = _lib.load_library(tuning_library())
    dev = "cuda"
    out = {"tiles": TILES, "wx": [], "wgrad_periods": []}
    shapes = [("fwd", 512, 512), ("dgrad", 512, 512), ("fwd", 17, 512), ("dgrad", 512, 51), ("fwd", 98, 512), ("dgrad", 512, 393)]
    sizes = (1024, 2048, 4096, 8192, 16384, 32768, 65536)
    if "--main" in sys.argv:   # the square layer at the shard and the headline size only, twice (run-to-run spread)
        shapes, sizes = shapes[:2] * 2, (8192, 65536)
    for B in (() if ("--pad" in sys.argv or "--stream" in sys.argv) else sizes):
        ldb = pad_ld(B)
        for kind, N, K in shapes:
            W = (torch.randn(N, (K + 31) // 32 * 32, device=dev) * 0.05)[:, :K]
            Wt = (torch.randn(K, (N + 31) // 32 * 32, device=dev) * 0.05)[:, :N]
            b = torch.randn(N, device=dev)
            X, Y = torch.randn(K, ldb, device=dev), torch.zeros(N, ldb, device=dev)
            dX = torch.zeros(K, ldb, device=dev)
            if kind == "fwd":
                fn = lambda: ops.linear_fwd(W, b, X, Y, B, 1)        # noqa: E731
            else:
                fn = lambda: ops.linear_dgrad(Wt, Y, X, dX, B, 1, False)  # noqa: E731
            rec = {"kind": kind, "N": N, "K": K, "B": B, "us": {}}
            M = N if kind == "fwd" else K
            for i, name in enumerate(TILES):
                bm = BM.get(name) or int(name.split("x")[0])
                if bm >= 2 * M and bm > 32 and not (bm == 64 and M > 32):   # (tiles more than twice as tall as the matrix: skipped)
                    continue
                os.environ["NIC_WX_TILE"] = str(i)
                rec["us"][name] = round(timeit(fn, iters=40 if "--main" in sys.argv else 20), 2)
            os.environ.pop("NIC_WX_TILE", None)
            rec["picked_us"] = round(timeit(fn), 2)
            rec["picked"] = (_lib.lib().nic_last_kernel() or b"").decode()
            best = min(rec["us"], key=rec["us"].get)
            rec["best"], rec["best_us"] = best, rec["us"][best]
            rec["tflops_best"] = round(2.0 * N * K * B / rec["best_us"] / 1e6, 1)
            out["wx"].append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
        del X, Y, dX
    if "--stream" in sys.argv:   # streamed small-launch kernel (NT, KS) against the LDS-DMA tilings
        out["stream"] = []
        shapes_s = [("fwd", 512, 512), ("dgrad", 512, 512), ("fwd", 17, 512), ("dgrad", 512, 51), ("fwd", 98, 512), ("fwd", 512, 66),
                    ("dgrad", 512, 18), ("fwd", 6, 512)]
        for B in (1024, 2048, 4096, 8192, 16384, 32768):
            ldb = pad_ld(B)
            for kind, N, K in shapes_s:
                M = N if kind == "fwd" else K
                if ((M + 31) // 32) * (B // 32) > 4096:
                    continue
                W = (torch.randn(N, (K + 31) // 32 * 32, device=dev) * 0.05)[:, :K]
                Wt = (torch.randn(K, (N + 31) // 32 * 32, device=dev) * 0.05)[:, :N]
                b = torch.randn(N, device=dev)
                X, Y = torch.randn(K, ldb, device=dev), torch.zeros(N, ldb, device=dev)
                dX = torch.zeros(K, ldb, device=dev)
                fn = (lambda: ops.linear_fwd(W, b, X, Y, B, 1)) if kind == "fwd" else (lambda: ops.linear_dgrad(Wt, Y, X, dX, B, 1, False))
                rec = {"kind": kind, "N": N, "K": K, "B": B, "us": {}}
                for mode in (0, 11, 12, 14, 21, 22, 24):
                    os.environ["NIC_WX_STREAM"] = str(mode)
                    rec["us"]["dma" if mode == 0 else f"nt{mode // 10}_ks{mode % 10}"] = round(timeit(fn, iters=40), 2)
                os.environ.pop("NIC_WX_STREAM", None)
                rec["picked_us"] = round(timeit(fn, iters=40), 2)
                rec["picked"] = (_lib.lib().nic_last_kernel() or b"").decode()
                out["stream"].append(rec)
                print(json.dumps(rec), file=sys.stderr, flush=True)
        print(json.dumps(out))
        return
    if "--pad" in sys.argv:   # occupancy experiment: unused dynamic LDS limits the co-resident workgroups per CU
        out["pad"] = []
        for B in (1024, 2048, 4096, 8192, 16384, 65536):
            ldb = pad_ld(B)
            W = torch.randn(512, 512, device=dev) * 0.05
            b = torch.randn(512, device=dev)
            X, Y = torch.randn(512, ldb, device=dev), torch.zeros(512, ldb, device=dev)
            for i, name in enumerate(TILES[:3]):
                for pad in (0, 16384, 32768, 49152, 65536):
                    os.environ["NIC_WX_TILE"], os.environ["NIC_WX_LDS_PAD"] = str(i), str(pad)
                    try:
                        us = round(timeit(lambda: ops.linear_fwd(W, b, X, Y, B, 1), iters=40), 2)
                    except Exception as e:   # (static + dynamic LDS beyond 160 KB: the launch is refused)
                        us = None
                    rec = {"B": B, "tile": name, "lds_pad": pad, "us": us}
                    out["pad"].append(rec)
                    print(json.dumps(rec), file=sys.stderr, flush=True)
        os.environ.pop("NIC_WX_TILE", None)
        os.environ.pop("NIC_WX_LDS_PAD", None)
        print(json.dumps(out))
        return
    # all-period weight gradients: old slot count (scenario splits only) vs the (period group x scenario split) count
    for B, T in (() if "--main" in sys.argv else ((1024, 50), (4096, 100), (8192, 100), (16384, 100))):
        ldb = pad_ld(B)
        for N, K in ((512, 512), (512, 51)):
            dZ, Xh = torch.randn(T, N, ldb, device=dev) * 0.01, torch.randn(T, K, ldb, device=dev)
            for label, slots in (("scenario_splits_only", ops.wgrad_num_splits(N, K, B)),
                                 ("period_groups", ops.wgrad_periods_num_splits(N, K, B, T))):
                slab = torch.zeros(slots, N, (K + 4) // 4 * 4, device=dev)
                us = timeit(lambda: ops.linear_wgrad_periods(dZ, Xh, slab, B), iters=5, warm=1)
                rec = {"N": N, "K": K, "B": B, "T": T, "mode": label, "slots": slots, "us": round(us, 1),
                       "tflops": round(2.0 * N * K * B * T / us / 1e6, 1), "kernel": (_lib.lib().nic_last_kernel() or b"").decode()}
                out["wgrad_periods"].append(rec)
                print(json.dumps(rec), file=sys.stderr, flush=True)
            del dZ, Xh
    print(json.dumps(out))


if __name__ == "__main__":
    main()
