"""Micro-benchmark of the demand sampler (csrc/sampler.hip): achieved HBM write bandwidth for the BASELINE shapes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neural_inventory_control_amd import _lib, ops  # noqa: E402
from neural_inventory_control_amd.layout import pad_ld  # noqa: E402


def timeit(fn, iters=10, warm=2):
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
    dev = "cuda"
    res = {}
    for name, S, B, T in (("cfg3", 16, 65536, 100), ("cfg5", 64, 32768, 70), ("cfg2", 1, 32768, 100)):
        out = torch.zeros(T, S, pad_ld(B), device=dev)
        mean = torch.full((S,), 5.0, device=dev)
        std = torch.full((S,), 1.5, device=dev)
        nbytes = 4.0 * S * B * T
        ms = timeit(lambda: ops.sample_demand_equicorrelated(out, T, S, B, 0, 1, mean, std, 0.5 if S > 1 else 0.0, True))
        res[f"{name}_equicorrelated_S{S}"] = dict(ms=round(ms, 4), gbs=round(nbytes / ms / 1e6, 1), kernel=_lib.lib().nic_last_kernel().decode())
        cov = 0.5 * np.outer(np.full(S, 1.5), np.full(S, 1.5))
        cov[range(S), range(S)] = 1.5 ** 2
        chol = torch.as_tensor(np.linalg.cholesky(cov).astype(np.float32)).to(dev)
        ms = timeit(lambda: ops.sample_demand(out, T, S, B, 0, 1, 0, mean, chol, True))
        res[f"{name}_cholesky_S{S}"] = dict(ms=round(ms, 4), gbs=round(nbytes / ms / 1e6, 1), kernel=_lib.lib().nic_last_kernel().decode())
        ms = timeit(lambda: ops.sample_demand(out, T, S, B, 0, 1, 1, mean, None, True))
        res[f"{name}_poisson_S{S}"] = dict(ms=round(ms, 4), gbs=round(nbytes / ms / 1e6, 1), kernel=_lib.lib().nic_last_kernel().decode())
        ms = timeit(lambda: out.fill_(1.0))
        res[f"{name}_fill_reference_S{S}"] = dict(ms=round(ms, 4), gbs=round(nbytes / ms / 1e6, 1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
