"""closed_form_body.h (whole-horizon closed-form policies with forward-mode gradients) compiled for the host, against the
reference's golden vectors of base_stock, capped_base_stock and echelon_stock."""
import pytest
import torch

import closed_form_checks as cfc
import hostsim_util
from oracle import inventory_oracle as orc


@pytest.fixture(scope="module")
def launch():
    return cfc.host_launch(hostsim_util.load())


@pytest.mark.parametrize("name", cfc.CLOSED_FORM_CASES)
def test_closed_form_body_matches_golden(launch, name):
    worst = cfc.check_against_golden(cfc.run_case(name, launch, "cpu"))
    assert worst <= 1e-5


@pytest.mark.parametrize("name", cfc.CLOSED_FORM_CASES)
def test_closed_form_profit_objective_matches_oracle(launch, name):
    """maximize_profit (environment.py:190-194): minimum(on hand, demand) with its 0.5 / 0.5 tie rule, against the oracle."""
    out = cfc.run_case(name, launch, "cpu", profit=True)
    g, c = out["golden"], out["config"]
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"], g.tensor("warehouse_upper_bound"))
    res, _, grads = orc.train_step_gradients(pol, c["periods"], c["problem_params"], g.data, c["observation_params"], c["ignore"])
    torch.testing.assert_close(out["rewards"], res.per_period, rtol=2e-6, atol=1e-5)
    for (k, got), ref in zip(out["grads"].items(), grads):
        if float(ref.abs().max()) == 0.0:
            continue
        assert float((got - ref).norm() / ref.norm()) <= 1e-5, k


@pytest.mark.parametrize("seed", range(12))
def test_closed_form_chain_on_random_serial_systems_matches_oracle(launch, seed):
    """1-3 extra echelons, random lead times / costs / demand moments / switches, n and T off the kernel's batch sizes: the host
    build of the chain body against the oracle's autograd (costs per period, gradient of the mean cost)."""
    assert cfc.check_random_serial_case(seed, launch, "cpu") <= 1e-5
