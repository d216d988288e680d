"""The whole per-period training step — forward rollout and the analytic backward sweep in `FusedRollout`'s order — re-played
on the CPU with the HOST build of the kernel bodies (env step + heads, tests/hostsim) and torch matmuls in place of the MFMA
GEMMs (tests/host_rollout.py), against the reference's golden vectors on the FULL batch of every MLP case, no scenario
excluded, no widened band: costs to 1e-6, d(mean_loss)/d(theta) per parameter tensor to 5e-6 relative L2.

This pins the adjoint of the kernel composition (tie rules, zero-order filter, which buffer each kernel accumulates into,
sweep order) independently of GEMM summation order; the same harness runs on the device with the HIP env / head kernels in
tests/test_gpu_rollout.py::test_hybrid_host_sweep_on_device."""
import pytest
import torch

import host_rollout as hr
import kernel_checks as kc
from golden_io import Golden, case_names

MLP_CASES = [n for n in case_names() if n.endswith("vanilla")]
HEAD = {"vanilla_one_store": "softplus", "vanilla_warehouse": "warehouse", "vanilla_serial": "serial",
        "vanilla_transshipment": "warehouse"}


def golden_layers(g):
    idx = sorted({int(k.split(".")[2]) for k in g.params})
    return [(g.params[f"net.master.{i}.weight"], g.params[f"net.master.{i}.bias"]) for i in idx]


def run_case(be, name):
    g = Golden(name)
    c = g.fresh_config()
    out = hr.run(be, c["problem_params"], g.data, golden_layers(g), head=HEAD[c["policy"]], periods=c["periods"],
                 ignore=c["ignore"], ub=float(g.z["warehouse_upper_bound"][0]),
                 adjacency=c["problem_params"].get("warehouse_store_adjacency"),
                 transshipment=c["nn_params"].get("transshipment", False))
    ref = g.grads
    keys = sorted(ref.keys(), key=lambda s: (int(s.split(".")[2]), s.split(".")[3] != "weight"))
    worst = max(float((a.double() - ref[k].double()).norm() / (ref[k].double().norm() + 1e-30))
                for a, k in zip(out["grads"], keys))
    return g, c, out, worst


@pytest.fixture(scope="module")
def be():
    return kc.HostSimBackend()


@pytest.mark.parametrize("name", MLP_CASES)
def test_host_sweep_matches_golden_costs_and_gradients(be, name):
    g, c, out, worst = run_case(be, name)
    assert abs(float(out["total"]) - float(g.z["total"])) <= 1e-6 * abs(float(g.z["total"]))
    assert abs(float(out["reported"]) - float(g.z["reported"])) <= 1e-6 * abs(float(g.z["reported"]))
    torch.testing.assert_close(out["rewards"].cpu(), g.tensor("rewards"), rtol=2e-6, atol=1e-5)
    for k, v in g.states(c["periods"]).items():
        # the recurrence amplifies float32 round-off of the policy in the loop: slots agree to 1e-5 of the largest pipeline entry
        torch.testing.assert_close(out["final"][k].cpu(), v, rtol=1e-4, atol=1e-4 + 1e-5 * float(v.abs().max()))
    assert worst <= 5e-6, worst
