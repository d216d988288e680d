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


@pytest.mark.parametrize("name", [n for n in case_names() if "serial" not in n])
def test_one_store_bodies_equal_the_quad_composition(be, name):
    """The one-store bodies (env_fwd_one_store / env_bwd_one_store / env_bwd_wh_g_after, round 4: what the whole-horizon data_driven
    kernels give each (scenario, store) lane) composed into a period the way csrc/horizon_rollout.hip does it, against the quad
    composition the per-period kernels run: every output BIT FOR BIT (same arithmetic, same Sum4 order), on every fixture's states
    and actions incl. forced ties and zero orders."""
    import torch
    from golden_io import Golden
    from neural_inventory_control_amd import layout
    from neural_inventory_control_amd.layout import EnvProblem, to_soa
    g = Golden(name)
    c = g.fresh_config()
    prob = EnvProblem(c["problem_params"], g.data, "cpu")
    if prob.E:
        pytest.skip("the whole-horizon kernels take settings without extra echelons")
    B, shift = c["n"], c["observation_params"]["demand"]["period_shift"]
    gen = torch.Generator().manual_seed(11)
    P = kc.P
    for t in (0, c["periods"] // 2, c["periods"] - 1):
        st = {k: v.float().clone() for k, v in g.states(t).items()}
        act = {k: v.float().clone() for k, v in g.actions(t).items()}
        st["store_inventories"][0, :, 0] = g.data["demands"][0, :, t + shift]   # on-hand == demand tie
        act["stores"][1 % B] = 0.0                                               # zero orders (no placement, no pipeline gradient)
        s, w, e = kc._state_soa(st, prob, "cpu")
        ts, tw, te, _keep = kc._orders_tables(act, "cpu")
        dem = kc._demand_table(g.data["demands"], t + shift)   # (kept alive: the io holds raw addresses)
        io = prob.make_io(s, w, e, dem, ts, tw, te)
        outs = []
        for per_store in (False, True):
            so, wo = torch.zeros_like(s), (torch.zeros_like(w) if prob.Wn else None)
            r = torch.zeros(prob.ldb)
            if per_store:
                assert be.h.hostsim_env_step_fwd_per_store(io, P(so), P(wo), P(r)) == 0
            else:
                be.env_fwd(io, so, wo, None, r)
            outs.append((so, wo, r))
        for a, b_ in zip(*outs):
            assert a is None or torch.equal(a, b_), (t, "forward")
        gso = to_soa(torch.randn(st["store_inventories"].shape, generator=gen), prob.ldb)
        gwo = to_soa(torch.randn(st["warehouse_inventories"].shape, generator=gen), prob.ldb) if prob.Wn else None
        grs = torch.zeros(prob.ldb)
        grs[:B] = torch.randn(B, generator=gen)
        outs = []
        for per_store in (False, True):
            gsi, gwi = torch.zeros_like(s), (torch.zeros_like(w) if prob.Wn else None)
            gas = torch.zeros(prob.S, prob.nsup, prob.ldb)
            gaw = torch.zeros(prob.Wn, prob.ldb) if prob.Wn else None
            tab = layout.Table(grs, 0, 1).t2()
            if per_store:
                assert be.h.hostsim_env_step_bwd_per_store(io, P(gso), P(gwo), tab, P(gsi), P(gwi), P(gas), P(gaw)) == 0
            else:
                be.env_bwd(io, gso, gwo, None, tab, gsi, gwi, None, gas, gaw, None)
            outs.append((gsi, gwi, gas, gaw))
        for a, b_ in zip(*outs):
            assert a is None or torch.equal(a, b_), (t, "backward")


@pytest.mark.parametrize("S,Wn,Ww", [(21, 3, 3), (5, 2, 4), (8, 1, 2)])
def test_data_driven_head_bodies_against_autograd(be, S, Wn, Ww):
    """head_data_driven_fwd_one / bwd_one (DataDrivenNet.forward :474-515 + apply_proportional_allocation :111-138: ReLU, adjacency
    mask, proportional scaling by the warehouse's pipeline total) against torch autograd of the same expression."""
    import torch
    gen = torch.Generator().manual_seed(3)
    B = 24
    ld = 64
    Z = (torch.randn(Wn + S * Wn, B, generator=gen) * 2).requires_grad_(True)
    wh = (torch.rand(Wn, Ww, B, generator=gen) * (S / 3.0)).requires_grad_(True)    # some warehouses short, some not
    mask = (torch.rand(S, Wn, generator=gen) > 0.2).float()
    out = torch.relu(Z)
    a = out[Wn:].view(S, Wn, B) * mask[:, :, None]
    sc = torch.clip(wh.sum(dim=1) / (a.sum(dim=0) + 1e-10), max=1)
    so_ref, wo_ref = a * sc[None], out[:Wn]
    g_so, g_wo = torch.randn(S, Wn, B, generator=gen), torch.randn(Wn, B, generator=gen)
    ((so_ref * g_so).sum() + (wo_ref * g_wo).sum()).backward()
    pad = lambda x: torch.cat([x.detach(), torch.zeros(*x.shape[:-1], ld - B)], dim=-1).contiguous()   # noqa: E731
    Zp, whp, gsop, gwop = pad(Z), pad(wh), pad(g_so), pad(g_wo)
    so, wo, dZ, gwh = torch.zeros(S, Wn, ld), torch.zeros(Wn, ld), torch.zeros(Wn + S * Wn, ld), torch.zeros(Wn, Ww, ld)
    P = kc.P
    assert be.h.hostsim_head_data_driven(P(Zp), P(whp), P(mask), P(gsop), P(gwop), P(so), P(wo), P(dZ), P(gwh), S, Wn, Ww, B, ld) == 0
    torch.testing.assert_close(so[..., :B], so_ref.detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(wo[..., :B], wo_ref.detach(), rtol=0, atol=0)
    torch.testing.assert_close(dZ[..., :B], Z.grad, rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(gwh[..., :B], wh.grad, rtol=2e-5, atol=1e-5)
