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
    for name in ("linear", "linear_backward", "env_step", "env_step_backward"):
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
