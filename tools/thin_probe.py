"""Times the fused thin-layer backward against the two GEMMs it replaces (BASELINE cfg3 logits layer: 17 x 512)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_inventory_control_amd import _lib, ops
from neural_inventory_control_amd.layout import pad_ld
from tools.gemm_probe import timeit


def main():
    dev, B, N, K = "cuda", 65536, 17, 512
    ldb = pad_ld(B)
    W = torch.randn(N, K, device=dev) * 0.05
    Wt = W.t().contiguous()
    dY = torch.randn(N, ldb, device=dev)
    H = torch.randn(K, ldb, device=dev)
    H = torch.where(H > 0, H, torch.expm1(H))
    dX = torch.zeros(K, ldb, device=dev)
    res = {}
    for splits in (ops.wgrad_num_splits(N, K, B), 64, 96, 128, 192, 384):
        slab = torch.zeros(splits, N, (K + 4) // 4 * 4, device=dev)
        ms = timeit(lambda: ops.linear_bwd_thin(W, dY, H, dX, slab, B, _lib.NIC_ACT_ELU))
        res[f"thin_splits{splits}"] = round(ms, 4)
    splits = ops.wgrad_num_splits(N, K, B)
    slab = torch.zeros(splits, N, (K + 4) // 4 * 4, device=dev)
    res["dgrad"] = round(timeit(lambda: ops.linear_dgrad(Wt, dY, H, dX, B, 1, False)), 4)
    res["wgrad"] = round(timeit(lambda: ops.linear_wgrad(dY, H, slab, B)), 4)
    res["algorithmic_MB"] = round(2 * K * B * 4 / 1e6, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
