"""Where a period of the 16-scenario whole-horizon forward kernel (csrc/small_rollout16.hip) goes: tuning build with wall-clock
stamps of workgroups 0 and 1 at 10 points of every period; runs training steps of cfg2 / cfg4 and prints the mean time between
consecutive points (us).

    python tools/small_rollout_stamp_probe.py [cfg2|cfg4]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import _lib  # noqa: E402
from gemm_probe import _tuning_library  # noqa: E402

LABELS = ["layer 1 MFMAs (4 steps x 2 tiles)", "ELU x 8", "history stores (state, hidden 1)", "layer 2 MFMAs (8 x 2)", "ELU x 8",
          "history stores + further hidden layers", "output layer MFMAs (8) + logit shuffles + store", "head", "env step + reward store + next demand"]


def main():
    lib = _lib._lib = _lib.load_library(_tuning_library())
    lib.nic_tuning_set_small_rollout_stamps.argtypes = [ctypes.c_void_p]
    import bench
    from neural_inventory_control_amd import small_rollout as sr
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    setting, policy, sc, data, model, eng, n, T, desc = bench.build_case(which, torch.device("cuda"), 0, 1, None, None, False)
    eng.materialize(eng.input_rows(data, setting["observation_params"]))

    def step():
        eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stamps = torch.zeros(T * 2 * 16, dtype=torch.int64, device="cuda")
    real = sr.small_rollout_fwd

    def wrapped(*a, **k):
        assert lib.nic_tuning_set_small_rollout_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        real(*a, **k)
        lib.nic_tuning_set_small_rollout_stamps(None)
    sr.small_rollout_fwd = wrapped
    step()
    torch.cuda.synchronize()
    sr.small_rollout_fwd = real
    st = stamps.view(T, 2, 16).double().cpu() / 100.0
    mid = st[2:T - 2]
    period = (st[1:, :, 0] - st[:-1, :, 0])[1:-1].mean(dim=0)
    print(f"--- small_rollout16 forward, {which}: {n} scenarios x T={T}; per period, workgroups 0 / 1 (us): {float(period[0]):.2f} / {float(period[1]):.2f}")
    for i, lab in enumerate(LABELS):
        dlt = mid[:, :, i + 1] - mid[:, :, i]
        print(f"    {lab:52s} {float(dlt[:, 0].mean()):6.2f} {float(dlt[:, 1].mean()):6.2f}")


if __name__ == "__main__":
    main()
