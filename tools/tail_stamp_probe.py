"""Where a launch of the fused per-period tail kernels (csrc/period_tail.hip) goes: compiles its own copy of the library with
-DNIC_TUNING_BUILD (wall-clock stamps of workgroup 0's four wavefronts at the stage boundaries), runs training steps of BASELINE
cfg3's setting at a given batch and prints, per direction, the mean time between consecutive stamps of every wavefront (us).

    python tools/tail_stamp_probe.py [n_scenarios] [periods]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import _lib  # noqa: E402
from gemm_probe import _tuning_library  # noqa: E402

FWD = ["A: logits contraction (K split)", "publish + collect + tiles -> LDS", "barrier", "B: head (softmax, orders)", "env: stores",
       "env: warehouses + reward", "orders / next state -> HBM", "C: next period's first layer"]
BWD = ["A': first layer's input gradient (K split)", "collect + tiles -> LDS", "barrier", "B': env adjoint", "head adjoint",
       "state gradient -> HBM", "barrier (group)", "C': logits layer backward", "slab update"]


def main():
    lib = _lib._lib = _lib.load_library(_tuning_library())
    lib.nic_tuning_set_tail_stamps.argtypes = [ctypes.c_void_p]
    import bench
    from neural_inventory_control_amd import ops
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    setting, policy, sc, data, model, eng, n, T, desc = bench.build_case("cfg3", torch.device("cuda"), 0, 1, n, T, False)
    eng.materialize(eng.input_rows(data, setting["observation_params"]))

    def step():
        eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    assert eng._use_tail()
    stamps = torch.zeros(2, T, 4, 16, dtype=torch.int64, device="cuda")
    count = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops.period_tail_fwd, ops.period_tail_bwd

    def wrap(real, which, d):
        def wrapped(*a, **k):
            i = count[which] % T
            count[which] += 1
            assert lib.nic_tuning_set_tail_stamps(ctypes.c_void_p(stamps[d, i].data_ptr())) == 0
            real(*a, **k)
            torch.cuda.synchronize()
            lib.nic_tuning_set_tail_stamps(None)
        return wrapped
    ops.period_tail_fwd, ops.period_tail_bwd = wrap(real_f, "fwd", 0), wrap(real_b, "bwd", 1)
    step()
    torch.cuda.synchronize()
    ops.period_tail_fwd, ops.period_tail_bwd = real_f, real_b
    st = stamps.double().cpu() / 100.0   # us
    for d, (name, labels) in enumerate((("tail_fwd", FWD), ("tail_bwd", BWD))):
        mid = st[d, 1:T - 1]   # (forward: periods with stage C; backward launch i = period T - 1 - i: those with stage A')
        total = (mid[:, :, len(labels)] - mid[:, :, 0])
        print(f"--- {name}: {n} scenarios; workgroup 0, mean over {T - 2} launches, per wavefront (us); entry -> last stamp: "
              + "  ".join(f"{float(total[:, w].mean()):6.2f}" for w in range(4)))
        for i, lab in enumerate(labels):
            dd = mid[:, :, i + 1] - mid[:, :, i]
            print(f"    {lab:46s} " + "  ".join(f"{float(dd[:, w].mean()):6.2f}" for w in range(4)))


if __name__ == "__main__":
    main()
