"""Probe of the weight-streaming 512 x 512 layer on 32-scenario blocks (tools/experiments/wide_layer_probe.hip): correctness against
torch and microseconds per layer per block with one workgroup per CU (8,192 columns) and eight rounds (65,536 columns)."""
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "..", "_build", "wide_layer_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "wide_layer_probe.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "wide_layer_probe.hip")])
torch.cuda.init()
lib = ctypes.CDLL(so)
lib.wide_layer_probe.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_void_p]


def pack(W):
    N, K = W.shape
    return W.view(N // 32, 32, K // 8, 4, 2).permute(0, 2, 4, 1, 3).contiguous()   # [tile][group][h][r][j]


def main():
    H = 512
    torch.manual_seed(0)
    W = torch.randn(H, H, device="cuda") / H ** 0.5
    b = torch.randn(H, device="cuda") * 0.1
    Wp = pack(W)
    res = {}
    for n in (8192, 65536):
        X = torch.randn(H, n, device="cuda")
        Y = torch.empty_like(X)
        stream = torch.cuda.current_stream().cuda_stream
        for pf in (2, 4):
            # correctness: one and three chained layers
            for it in (1, 3):
                assert lib.wide_layer_probe(Wp.data_ptr(), b.data_ptr(), X.data_ptr(), Y.data_ptr(), H, n, n, it, pf, 0, None, stream) == 0
                ref = X.double()
                for _ in range(it):
                    ref = torch.nn.functional.elu(W.double() @ ref + b.double()[:, None])
                err = float((Y.double() - ref).abs().max() / ref.abs().max())
                assert err < 1e-5, (n, pf, it, err)
            for hist_on in (0, 1):
                n_it = 16
                hist = torch.empty(n_it, H, n, device="cuda") if hist_on else None
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                for rep in range(3):
                    if rep == 1:
                        ev[0].record()
                    lib.wide_layer_probe(Wp.data_ptr(), b.data_ptr(), X.data_ptr(), Y.data_ptr(), H, n, n, n_it, pf, hist_on,
                                         hist.data_ptr() if hist_on else None, stream)
                ev[1].record()
                torch.cuda.synchronize()
                ms = ev[0].elapsed_time(ev[1]) / 2
                rounds = max(1, n // 32 // 256)
                us_per_layer_block = ms * 1e3 / n_it / rounds
                tf = 2.0 * H * H * n * n_it / (ms * 1e-3) / 1e12
                res[f"n{n}_pf{pf}_hist{hist_on}"] = {"ms": round(ms, 4), "us_per_layer_per_block_round": round(us_per_layer_block, 2), "tflops": round(tf, 1)}
                del hist
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
