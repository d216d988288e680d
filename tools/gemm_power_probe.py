"""Is the 512 x 512 policy GEMM limited by the matrix pipe's issue rate or by the clock the part sustains under load?
Times the same launches on operands of different toggle activity (zeros / constant / N(0,1) / ELU outputs / tiny gradients):
the instruction stream is identical, so any difference is the power-managed clock.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_inventory_control_amd import ops
from neural_inventory_control_amd.layout import pad_ld


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev, B, N, K = "cuda", 65536, 512, 512
    ldb = pad_ld(B)
    flops = 2.0 * N * K * B
    W = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev) * 0.1
    Y = torch.zeros(N, ldb, device=dev)
    dX = torch.zeros(K, ldb, device=dev)
    H = torch.nn.functional.elu(torch.randn(K, ldb, device=dev))
    fills = {
        "zeros": torch.zeros(K, ldb, device=dev),
        "ones": torch.ones(K, ldb, device=dev),
        "normal": torch.randn(K, ldb, device=dev),
        "elu_outputs": H.clone(),
        "tiny_gradients": torch.randn(K, ldb, device=dev) * 1e-7,
    }
    out = {}
    for name, X in fills.items():
        ms = timeit(lambda: ops.linear_fwd(W, bias, X, Y, B, 1))
        out[f"fwd[{name}]"] = dict(us=round(ms * 1e3, 1), tflops=round(flops / ms / 1e9, 1))
        ms = timeit(lambda: ops.linear_dgrad(W, X, H, dX, B, 1, False))
        out[f"dgrad[{name}]"] = dict(us=round(ms * 1e3, 1), tflops=round(flops / ms / 1e9, 1))
    Wz = torch.zeros_like(W)
    ms = timeit(lambda: ops.linear_fwd(Wz, bias, fills["zeros"], Y, B, 1))
    out["fwd[zero weights, zero input]"] = dict(us=round(ms * 1e3, 1), tflops=round(flops / ms / 1e9, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
