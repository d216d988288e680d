"""Where a period of the whole-horizon data_driven kernels (csrc/horizon_rollout.hip) goes: compiles its own copy of the library
with -DNIC_TUNING_BUILD (wall-clock stamps of workgroup 0's four wavefronts at ~10 points of every period), runs training steps of
the real_data_driven workload and prints, per direction, the mean time between consecutive stamps of every wavefront (us).

    python tools/horizon_stamp_probe.py [n_scenarios] [periods]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import _lib  # noqa: E402
from gemm_probe import _tuning_library  # noqa: E402

FWD = ["top: prefetch consumed, global burst issued", "layer 1 + barrier", "layer 2 + barrier", "layer 3 + barrier",
       "head: masked ReLU | barrier | scales | barrier", "stores: final orders + env step + barrier", "warehouses / cost partials + barrier",
       "reward"]
BWD = ["top: history -> LDS, global burst + barrier", "env adjoint (stores, warehouses) + head pieces + barrier",
       "head adjoint: per-warehouse sums | barrier | dZ3 | barrier", "dgrad 3 + barrier", "dgrad 2 + barrier", "dgrad 1 (state rows)", "barrier"]


def main():
    lib = _lib._lib = _lib.load_library(_tuning_library())
    lib.nic_tuning_set_horizon_stamps.argtypes = [ctypes.c_void_p]
    import bench
    from neural_inventory_control_amd import horizon_rollout as hz
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 95
    setting, policy, sc, data, model, eng, n, T, desc = bench.build_case("real_data_driven", torch.device("cuda"), 0, 1, n, T, False)
    eng.materialize(eng.input_rows(data, setting["observation_params"]))

    def step():
        eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    assert eng.horizon is not None
    stamps = torch.zeros(T * 4 * 16, dtype=torch.int64, device="cuda")
    for name, labels in (("horizon_fwd", FWD), ("horizon_bwd", BWD)):
        real = getattr(hz, name)

        def wrapped(*a, _real=real, **k):
            assert lib.nic_tuning_set_horizon_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
            _real(*a, **k)
            lib.nic_tuning_set_horizon_stamps(None)
        setattr(hz, name, wrapped)
        stamps.zero_()
        step()
        torch.cuda.synchronize()
        setattr(hz, name, real)
        st = stamps.view(T, 4, 16).double().cpu() / 100.0     # us
        mid = st[2:T - 2]
        print(f"--- {name}: {n} scenarios x T={T}; mean over periods 2..{T - 3}, per wavefront (us)")
        period = (st[1:, 0, 0] - st[:-1, 0, 0]).abs()
        raw = stamps.view(T, 4, 16).double().cpu()
        cyc = (raw[1:, 0, 15] - raw[:-1, 0, 15]).abs()[1:-1].mean()
        print(f"    period (wave 0, point 0 to point 0): mean {float(period[1:-1].mean()):.2f} us = {float(cyc):.0f} s_memtime ticks "
              f"({float(cyc / period[1:-1].mean()):.0f} per us)")
        for i, lab in enumerate(labels):
            d = mid[:, :, i + 1] - mid[:, :, i]
            print(f"    {lab:42s} " + "  ".join(f"{float(d[:, w].mean()):6.2f}" for w in range(4)))


if __name__ == "__main__":
    main()
