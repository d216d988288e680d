"""In-kernel stamps of the GNN period kernel (csrc/gnn_period.hip) on the tuning build: wall-clock (100 MHz) of workgroup 0's eight
wavefronts at the stage boundaries of one forward launch, plus HIP-event times of the launch against the per-MLP launches.
    python tools/gnn_period_probe.py [--scenarios 8192] [--periods 10] [--eval]
Test infrastructure: the product library contains no stamp code (NIC_TUNING_BUILD)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

STAGES = ["weights staged", "initial_node", "initial_edge", "node_update (+ aggregation)", "edge_update + output", "allocation + env"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenarios", type=int, default=8192)
    ap.add_argument("--periods", type=int, default=10)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--workload", default="gnn")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from neural_inventory_control_amd import _lib
    from gemm_probe import _tuning_library
    lib = _lib._lib = _lib.load_library(_tuning_library())
    from bench import build_case
    from neural_inventory_control_amd.rollout import KernelTimer
    dev = torch.device("cuda", 0)
    setting, policy, sc, data, model, eng, n, T, desc = build_case(args.workload, dev, 0, 1, args.scenarios, args.periods, False)
    eng.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
    lib.nic_tuning_set_gnn_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
    res = {"workload": args.workload, "scenarios": n, "periods": T, "train": not args.eval}
    for period in (True, False):
        eng.use_period_kernel = period
        for _ in range(2):
            eng.run(data, T, 0, train=not args.eval, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        eng.timer = KernelTimer(stride=1)
        if period:
            lib.nic_tuning_set_gnn_stamps(stamps.data_ptr())
        eng.run(data, T, 0, train=not args.eval, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        torch.cuda.synchronize()
        lib.nic_tuning_set_gnn_stamps(None)
        fwd = {k: ms for k, (_, ms) in eng.timer.summary().items() if "fwd" in k}
        res["period_kernel" if period else "per_mlp_launches"] = {k: round(ms * 1e3, 2) for k, ms in fwd.items()}
        res[("period_kernel" if period else "per_mlp_launches") + "_fwd_us_per_period"] = round(sum(fwd.values()) * 1e3, 2)
        eng.timer = None
    st = stamps.cpu().view(16, 16)
    st = st[st[:, 1] != 0]   # the wavefronts the launch had
    nw = st.shape[0]
    t0 = int(st[:, 0].min())
    res["stamps_us_wave_by_point"] = [[round((int(st[w, p]) - t0) / 100.0, 2) if int(st[w, p]) else None for p in range(7)] for w in range(nw)]
    res["stage_us_slowest_wave"] = {STAGES[p - 1]: round((int(st[:, p].max()) - int(st[:, p - 1].max())) / 100.0, 2)
                                    for p in range(1, 7) if int(st[:, p].max())}
    names = ["first layer (8 or 16 steps + lead)", "ELU 1", "H1 stores", "second layer", "ELU 2 + H2 stores", "third layer", "ELU 3 + tile -> LDS + Y stores"]
    res["first_initial_edge_tile_us_by_wave"] = [
        {names[p - 9]: round((int(st[w, p]) - int(st[w, p - 1])) / 100.0, 2) for p in range(9, 16) if int(st[w, p]) and int(st[w, p - 1])}
        for w in range(nw)]
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
