"""The per-scenario bodies of the HIP kernels (env_step_body.h, policy_heads_body.h), compiled for the HOST, against
the golden vectors (forward) and the oracle's autograd (backward).  Checks the kernels' arithmetic on the CPU
container; the identical checks run through the real HIP library in test_gpu_kernels.py on the GPU box."""
import pytest

import kernel_checks as kc
from golden_io import case_names


@pytest.fixture(scope="module")
def be():
    return kc.HostSimBackend()


@pytest.mark.parametrize("name", case_names())
def test_env_forward_matches_golden(be, name):
    kc.check_env_forward(be, name)


@pytest.mark.parametrize("name", case_names())
@pytest.mark.parametrize("profit", [False, True])
def test_env_backward_matches_oracle_autograd(be, name, profit):
    kc.check_env_backward(be, name, profit)


@pytest.mark.parametrize("S,Wn,adj", kc.WAREHOUSE_HEAD_CASES)
@pytest.mark.parametrize("trans", [False, True])
def test_warehouse_head(be, S, Wn, adj, trans):
    kc.check_warehouse_head(be, S, Wn, adj, trans)


def test_softplus_head(be):
    kc.check_softplus_head(be)


@pytest.mark.parametrize("E", [2, 1, 3])
def test_serial_head(be, E):
    kc.check_serial_head(be, E)


def test_zero_lead_orders_hand_computed_period(be):
    """Pins the "drop" semantics (what ZERO_LEAD_CASES are compared against) with numbers worked out by hand."""
    kc.check_zero_lead_micro(be)
