"""In-kernel stamps of the GNN period BACKWARD kernel (csrc/gnn_period_bwd.hip) on the tuning build: wall-clock (100 MHz) of
workgroup 0's eight wavefronts at the stage boundaries of one backward launch, plus HIP-event times of the launch against the
per-MLP launches it replaces (five nic_mlp3_bwd_hist + three nic_segment_sum_terms + the row adds).
    python tools/gnn_period_bwd_probe.py [--workload gnn] [--scenarios 8192] [--periods 10] [--out file.json]
Test infrastructure: the product library contains no stamp code (NIC_TUNING_BUILD)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

STAGES = ["weights + tables staged", "A output MLP (+ flush)", "B edge update (+ flush)", "C1 node sums of dz1 (+ flush of the endpoint columns)",
          "C2 node update (+ flush)", "D initial edge (+ flush)", "E1 node sums of dz1, aggregation columns of the node update (+ flushes)",
          "E2 initial node (+ flush)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenarios", type=int, default=8192)
    ap.add_argument("--periods", type=int, default=10)
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
    lib.nic_tuning_set_gnn_bwd_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
    res = {"workload": args.workload, "scenarios": n, "periods": T}
    grads = {}
    for period in (True, False):
        eng.use_period_bwd = period
        for _ in range(2):
            eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        eng.timer = KernelTimer(stride=1)
        if period:
            lib.nic_tuning_set_gnn_bwd_stamps(stamps.data_ptr())
        eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        torch.cuda.synchronize()
        lib.nic_tuning_set_gnn_bwd_stamps(None)
        grads[period] = [g.clone() for _, g in eng.param_grads()]
        summ = eng.timer.summary()
        bwd = {k: ms for k, (_, ms) in summ.items() if "bwd" in k}
        name = "period_kernel" if period else "per_mlp_launches"
        res[name] = {k: round(ms * 1e3, 2) for k, ms in bwd.items()}
        res[name + "_timed_bwd_us_per_period"] = round(sum(bwd.values()) * 1e3, 2)
        res[name + "_n_sub"] = getattr(eng, "_n_sub", None) if period else None
        eng.timer = None
    res["worst_relative_gradient_difference"] = max(float((a - b).norm() / (b.norm() + 1e-30)) for a, b in zip(grads[True], grads[False]))
    st = stamps.cpu().view(16, 16)
    st = st[st[:, 1] != 0]   # the wavefronts the launch had
    t0 = int(st[:, 0].min())
    res["wavefronts"] = int(st.shape[0])
    res["stamps_us_wave_by_point"] = [[round((int(st[w, p]) - t0) / 100.0, 2) if int(st[w, p]) else None for p in range(9)]
                                      for w in range(st.shape[0])]
    res["stage_us_slowest_wave"] = {STAGES[p - 1]: round((int(st[:, p].max()) - int(st[:, p - 1].max())) / 100.0, 2)
                                    for p in range(1, 9) if int(st[:, p].max())}
    res["workgroup_us"] = round((int(st[:, 8].max()) - t0) / 100.0, 2)
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
