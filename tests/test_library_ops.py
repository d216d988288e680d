"""`torch.library` registration of nic::linear / nic::env_step (neural_inventory_control_amd/library.py).
CPU: the operators exist and their fake-tensor kernels give the shapes / strides the device kernels produce.
GPU: `torch.library.opcheck` (schema, fake tensor, autograd registration, AOT dispatch), equality with the eager autograd
Functions, and a policy of HipLinear layers through torch.compile."""
import pytest
import torch

from neural_inventory_control_amd import _lib, library
from neural_inventory_control_amd.layout import pad_ld

HAS_GPU = torch.cuda.is_available()
DEV = "cuda:0"


def test_operators_are_registered_with_fake_kernels():
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("linear", "linear_backward", "env_step", "env_step_backward", "softmax_alloc", "softmax_alloc_backward",
                 "rollout_closed_form", "sample_demand"):
        assert hasattr(torch.ops.nic, name)
    with FakeTensorMode():
        x, w, b = torch.empty(100, 7), torch.empty(32, 7), torch.empty(32)
        y = torch.ops.nic.linear(x, w, b, _lib.NIC_ACT_ELU)
        assert tuple(y.shape) == (100, 32) and y.stride() == (1, pad_ld(100))
        gx, gw, gb = torch.ops.nic.linear_backward(torch.empty(100, 32), x, y, w, _lib.NIC_ACT_ELU, True)
        assert tuple(gx.shape) == (100, 7) and tuple(gw.shape) == (32, 7) and tuple(gb.shape) == (32,)
        ld = pad_ld(100)
        store, a_store, dem = torch.empty(3, 4, ld), torch.empty(100, 3, 1), torch.empty(3, ld)
        s2, w2, e2, r = torch.ops.nic.env_step(store, None, None, a_store, None, None, dem, 0)
        assert s2.shape == store.shape and w2.numel() == 0 and e2.numel() == 0 and tuple(r.shape) == (ld,)
        so, wo = torch.ops.nic.softmax_alloc(torch.empty(100, 3 * 2 + 2), torch.empty(100, 2, 3), torch.empty(2, 3, dtype=torch.int32),
                                             torch.empty(1), False, 3, 2)
        assert tuple(so.shape) == (100, 3, 2) and so.stride() == (1, 2 * ld, ld) and tuple(wo.shape) == (100, 2, 1)
        dz, gi = torch.ops.nic.softmax_alloc_backward(so, wo, torch.empty(100, 8), torch.empty(100, 2, 3),
                                                      torch.empty(2, 3, dtype=torch.int32), torch.empty(1), False, 3, 2)
        assert tuple(dz.shape) == (100, 8) and tuple(gi.shape) == (100, 2, 3)
        tot, rep, gl = torch.ops.nic.rollout_closed_form(torch.empty(2), torch.empty(9, 1, ld), torch.empty(1, 4, ld), 0, 1, 9, 0, 2, False)
        assert tot.dim() == 0 and rep.dim() == 0 and tuple(gl.shape) == (2,)
        d = torch.ops.nic.sample_demand(torch.empty(16), torch.empty(16), 0.5, 30, 1000, 0, 7, True, False)
        assert tuple(d.shape) == (30, 16, pad_ld(1000))


def _one_store_problem(B=96, S=1, Ws=3, T=5):
    from neural_inventory_control_amd.layout import EnvProblem, to_soa
    torch.manual_seed(0)
    data = {"demands": (torch.rand(B, S, T) * 6).to(DEV), "initial_inventories": (torch.rand(B, S, Ws) * 5).to(DEV),
            "underage_costs": torch.full((B, S), 9.0, device=DEV), "holding_costs": torch.full((B, S), 1.0, device=DEV),
            "lead_times": torch.full((B, S), 2.0, device=DEV)}
    pp = {"n_stores": S, "n_warehouses": 0, "n_extra_echelons": 0, "lost_demand": True, "maximize_profit": False}
    prob = EnvProblem(pp, data, torch.device(DEV))
    store = to_soa(data["initial_inventories"], prob.ldb)
    dem = torch.zeros(T, S, prob.ldb, device=DEV)
    dem[:, :, :B] = data["demands"].permute(2, 1, 0)
    return prob, store, dem, B


@pytest.mark.gpu
def test_opcheck_linear_and_env_step():
    x = torch.randn(200, 11, device=DEV, requires_grad=True)
    w = torch.randn(32, 11, device=DEV, requires_grad=True)
    b = torch.randn(32, device=DEV, requires_grad=True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    for args in ((x, w, b, _lib.NIC_ACT_ELU), (x, w, None, _lib.NIC_ACT_NONE)):
        torch.library.opcheck(torch.ops.nic.linear, args, test_utils=tests)
    prob, store, dem, B = _one_store_problem()
    h = library.register_problem(prob)
    a = (torch.rand(B, 1, 1, device=DEV) * 4).requires_grad_()
    torch.library.opcheck(torch.ops.nic.env_step, (store.clone().requires_grad_(), None, None, a, None, None, dem[0], h),
                          test_utils=tests)
    library.release_problem(h)


@pytest.mark.gpu
def test_registered_operators_equal_the_eager_functions():
    """Same kernels behind both entry points: forward values and every gradient are bit-identical."""
    from neural_inventory_control_amd.environment import _EnvStepFunction
    from neural_inventory_control_amd.layout import Table
    from neural_inventory_control_amd.neural_networks import _LinearFunction
    torch.manual_seed(1)
    x0, w0, b0 = torch.randn(300, 19, device=DEV), torch.randn(64, 19, device=DEV), torch.randn(64, device=DEV)
    res = []
    for fn in (lambda x, w, b: torch.ops.nic.linear(x, w, b, _lib.NIC_ACT_ELU), lambda x, w, b: _LinearFunction.apply(x, w, b, _lib.NIC_ACT_ELU)):
        x, w, b = (t.clone().requires_grad_() for t in (x0, w0, b0))
        y = fn(x, w, b)
        (y * y).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
    prob, store0, dem, B = _one_store_problem()
    h = library.register_problem(prob)
    a0 = torch.rand(B, 1, 1, device=DEV) * 4
    out = []
    for which in ("op", "function"):
        store, a = store0.clone().requires_grad_(), a0.clone().requires_grad_()
        if which == "op":
            s2, _, _, r = torch.ops.nic.env_step(store, None, None, a, None, None, dem[0], h)
        else:
            s2, _, _, r = _EnvStepFunction.apply(prob, Table(dem[0], prob.ldb, 1), store, None, None, a, None, None)
        (r[:B].sum() + (s2 * s2).sum()).backward()
        out.append((s2.detach().clone(), r.detach().clone(), store.grad.clone(), a.grad.clone()))
    for p, q in zip(*out):
        assert torch.equal(p, q)
    library.release_problem(h)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_policy_of_hip_linear_layers_survives_torch_compile(backend):
    """A plugin-style policy (HipLinear + fused ELU layers) compiled with torch.compile: the traced graph calls nic::linear; output
    and parameter gradients equal the eager run bit for bit."""
    from neural_inventory_control_amd.neural_networks import HipLinear
    torch.manual_seed(2)

    class Policy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2, self.l3 = HipLinear(9, 32), HipLinear(32, 32), HipLinear(32, 3)
            self.l1.fused_act = self.l2.fused_act = _lib.NIC_ACT_ELU

        def forward(self, x):
            return torch.nn.functional.softplus(self.l3(self.l2(self.l1(x))))
    model = Policy().to(DEV)
    x = torch.randn(160, 9, device=DEV)
    y = model(x)
    y.sum().backward()
    want = [y.detach().clone()] + [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    torch._dynamo.reset()
    if backend == "inductor":   # compile in this process: no worker pool (processes started behind an initialised GPU are refused on the pool's boxes)
        import torch._inductor.config as icfg
        icfg.compile_threads = 1
    compiled = torch.compile(model, backend=backend, fullgraph=True)
    y2 = compiled(x)
    y2.sum().backward()
    got = [y2.detach().clone()] + [p.grad.clone() for p in model.parameters()]
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=0, atol=0) if backend == "aot_eager" else torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_opcheck_softmax_alloc_closed_form_and_sampler():
    """Round 6: the remaining operators SURVEY 8(b) names - opcheck (schema, fake tensor, autograd registration, AOT dispatch) and
    equality with the engine's own entry points."""
    from neural_inventory_control_amd import closed_form, workloads
    from neural_inventory_control_amd.layout import EnvProblem
    from neural_inventory_control_amd.neural_networks import _WarehouseHead
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    torch.manual_seed(3)
    B, S, Wn, Ww = 200, 5, 2, 3
    z0, wh0 = torch.randn(B, S * Wn + Wn, device=DEV), torch.rand(B, Wn, Ww, device=DEV) * 30
    adj = torch.tensor([[1, 1, 0, 1, 1], [1, 0, 1, 1, 1]], dtype=torch.int32, device=DEV)
    z, wh = z0.clone().requires_grad_(), wh0.clone().requires_grad_()
    ub = torch.tensor([90.0], device=DEV)
    torch.library.opcheck(torch.ops.nic.softmax_alloc, (z, wh, adj, ub, False, S, Wn), test_utils=tests)
    res = []
    for fn in (lambda a, b: torch.ops.nic.softmax_alloc(a, b, adj, ub, False, S, Wn),
               lambda a, b: _WarehouseHead.apply(a, b, adj, 90.0, False, S, Wn)):
        a, b = z0.clone().requires_grad_(), wh0.clone().requires_grad_()
        so, wo = fn(a, b)
        ((so * so).sum() + (wo * 3).sum()).backward()
        res.append((so.detach().clone(), wo.detach().clone(), a.grad.clone(), b.grad.clone()))
    for p, q in zip(*res):
        assert torch.equal(p, q)
    # closed-form rollout: the operator against the engine (same launch): totals and the level gradient
    from collections import defaultdict
    from neural_inventory_control_amd.closed_form import ClosedFormRollout
    from neural_inventory_control_amd.data_handling import Scenario
    from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
    for wl in ("base_stock", "echelon_stock"):
        setting, policy, _, _, _ = workloads.get(wl)
        obs = defaultdict(lambda: None, setting["observation_params"])
        n, T = 500, 12
        sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n, obs,
                      dict(setting["seeds"]), sampler="hip", device=DEV)
        data = {k: v.to(DEV) for k, v in sc.get_data().items()}
        torch.manual_seed(5)
        model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
        eng = ClosedFormRollout(model, setting["problem_params"], DEV)
        total, reported = eng.run(data, T, 3, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        total.backward()
        want = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        prob = EnvProblem(setting["problem_params"], data, torch.device(DEV))
        h = library.register_problem(prob)
        state0 = closed_form.pack_state0(data, prob)
        levels = model.closed_form_levels()
        pid = closed_form.POLICY_ID[policy["name"]]
        t2, r2, gl = torch.ops.nic.rollout_closed_form(levels, sc.demands_soa, state0, h, pid, T, 0, 3, False)
        t2.backward()
        assert float(t2) == float(total) and float(r2) == float(reported)
        for a, b in zip([p.grad for p in model.parameters()], want):
            assert torch.equal(a, b)
        lv = levels.detach().clone().requires_grad_()
        torch.library.opcheck(torch.ops.nic.rollout_closed_form, (lv, sc.demands_soa, state0, h, pid, T, 0, 3, False), test_utils=tests)
        library.release_problem(h)
    # sampler: the operator reproduces the Scenario's own trace
    setting, policy, _, _, _ = workloads.get("cfg3")
    obs = defaultdict(lambda: None, setting["observation_params"])
    sc = Scenario(9, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], 777, obs,
                  dict(setting["seeds"]), sampler="hip", device=DEV)
    pp, dp, seed = sc._device_sampler_args
    import numpy as np
    mean = torch.as_tensor(np.asarray(dp["mean"], dtype=np.float32)).to(DEV)
    std = torch.as_tensor(np.asarray(dp["std"], dtype=np.float32)).to(DEV)
    d = torch.ops.nic.sample_demand(mean, std, float(dp.get("correlation", 0.0) or 0.0), 9, 777, 0, int(seed), bool(dp["clip"]), False)
    assert torch.equal(d, sc.demands_soa)
    torch.library.opcheck(torch.ops.nic.sample_demand, (mean, std, 0.5, 9, 777, 0, int(seed), True, False),
                          test_utils=("test_schema", "test_faketensor", "test_aot_dispatch_dynamic"))


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["aot_eager"])
def test_compiled_rollout_through_simulator_step_equals_eager(backend):
    """A plugin policy (HipLinear layers + the registered feasibility head inside `VanillaWarehouse`) rolled out through
    `Simulator.step` for T periods under torch.compile(fullgraph=True): while the compiler traces, `step` emits `nic::env_step`
    and the policy `nic::linear` / `nic::softmax_alloc`; total cost and every parameter gradient equal the eager rollout."""
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from neural_inventory_control_amd.data_handling import Scenario
    from neural_inventory_control_amd.environment import Simulator
    from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
    setting, policy, _, _, _ = workloads.get("cfg3")
    setting["problem_params"]["n_stores"] = 4
    policy["neurons_per_hidden_layer"]["master"] = [32, 32]
    obs_p = defaultdict(lambda: None, setting["observation_params"])
    n, T = 192, 4
    sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n, obs_p,
                  dict(setting["seeds"]), sampler="hip", device=DEV)
    data = {k: v.to(DEV) for k, v in sc.get_data().items()}
    torch.manual_seed(11)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
    sim = Simulator(device=DEV)

    def rollout():
        observation, _ = sim.reset(T, setting["problem_params"], data, obs_p)
        total = torch.zeros((), device=DEV)
        for _ in range(T):
            action = model(observation)
            observation, reward, _, _, _ = sim.step(action)
            total = total + reward.sum()
        return total
    with torch.no_grad():
        rollout()      # materialises the lazy layers
    model.zero_grad()
    want = rollout()
    want.backward()
    want_g = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    torch._dynamo.reset()

    def periods(observation):
        total = torch.zeros((), device=DEV)
        for _ in range(T):
            action = model(observation)
            observation, reward, _, _, _ = sim.step(action)
            total = total + reward.sum()
        return total
    compiled = torch.compile(periods, backend=backend, fullgraph=True)
    observation, _ = sim.reset(T, setting["problem_params"], data, obs_p)
    got = compiled(observation)
    got.backward()
    assert float(got) == float(want)
    for a, b in zip([p.grad for p in model.parameters()], want_g):
        torch.testing.assert_close(a, b, rtol=0, atol=0)
