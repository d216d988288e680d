"""`torch.library` registration of the engine's two plugin-facing operators (SURVEY §8b names them):

    nic::linear(x, weight, bias?, act) -> y            the policy layers' Linear(+ELU) on csrc/linear_mfma.hip
    nic::env_step(store, wh?, ech?, a_store, a_wh?, a_ech?, demand, problem) -> (store', wh', ech', reward)
                                                       one period of Simulator.step on csrc/env_step.hip

each with a fake-tensor kernel (shapes / strides without touching the device) and an autograd formula that calls the matching
backward operator (`nic::linear_backward`, `nic::env_step_backward`), so a plugin policy built from `HipLinear` layers survives
`torch.compile` - the traced graph holds `torch.ops.nic.linear` calls - and functorch / `torch.func` callers see ordinary
operators.  Eager code keeps calling the `autograd.Function`s of neural_networks.py / environment.py (same kernels, none of the
dispatcher's per-call cost); `HipLinear.forward` switches to the registered operator only while a compiler is tracing.

`nic::env_step` takes the static problem (cost / lead-time tables, sizes) as an integer handle from `register_problem`: custom
operators carry tensors and scalars, not Python objects.
"""
from typing import Optional, Tuple

import torch

from . import _lib, ops
from .layout import Table, pad_ld, ref_view, to_soa
from .ops import EnvState

_PROBLEMS = {}


def register_problem(prob):
    """Handle of an `EnvProblem` for `nic::env_step` (kept alive until `release_problem`)."""
    h = id(prob)
    _PROBLEMS[h] = prob
    return h


def release_problem(handle):
    _PROBLEMS.pop(handle, None)


def _feature_major(x):
    # a (B, K) view of a feature-major [K][ld] buffer is taken as is - only with ld = pad_ld(B), the column stride the fake kernels
    # promise for the outputs (a wider buffer is copied: real and fake strides must agree for inductor's stride assertions)
    xt = x.t()
    if x.is_cuda and xt.stride(1) == 1 and xt.stride(0) == pad_ld(x.shape[0]) and xt.data_ptr() % 16 == 0 and x.dtype == torch.float32:
        return xt
    return to_soa(x.float())


# ---- nic::linear ---------------------------------------------------------------------------------------------------------------

@torch.library.custom_op("nic::linear", mutates_args=())
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """y (B, N) = act(x (B, K) @ weight^T + bias); the result is the (B, N) view of a feature-major [N][ld] buffer (strides (1, ld)),
    which the next layer reads without a copy."""
    if not x.is_cuda:
        raise _lib.NicUnavailableError("nic::linear needs device tensors (no CPU fallback)")
    B = x.shape[0]
    X = _feature_major(x)
    N = weight.shape[0]
    Y = torch.empty(N, X.stride(0), device=x.device, dtype=torch.float32)
    if X.stride(0) > B:
        Y[:, B:].zero_()
    ops.linear_fwd(weight.detach().contiguous(), None if bias is None else bias.detach(), X, Y, B, act)
    return ref_view(Y, B)


@linear.register_fake
def _(x, weight, bias, act):
    B, N = x.shape[0], weight.shape[0]
    return x.new_empty_strided((B, N), (1, pad_ld(B)), dtype=torch.float32)


@torch.library.custom_op("nic::linear_backward", mutates_args=())
def linear_backward(gy: torch.Tensor, x: torch.Tensor, y: torch.Tensor, weight: torch.Tensor, act: int,
                    has_bias: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(gx (B, K), gw (N, K), gb (N)) of nic::linear from the output gradient, the input and the OUTPUT (ELU' = y > 0 ? 1 : y + 1)."""
    B = x.shape[0]
    N, K = weight.shape
    X = _feature_major(x)
    ld = X.stride(0)
    if act == _lib.NIC_ACT_ELU:
        gy = torch.where(y > 0, gy, gy * (y + 1))
    dZ = torch.zeros(N, ld, device=gy.device, dtype=torch.float32)
    dZ[:, :B] = gy.t()
    lds = (K + 4) // 4 * 4
    splits = max(1, min(ops.wgrad_num_splits(N, K, B), (64 << 20) // (N * lds * 4)))
    slab = torch.zeros(splits, N, lds, device=gy.device)
    ops.linear_wgrad(dZ, X, slab, B)
    gw, gb = torch.empty(N, K, device=gy.device), torch.empty(N, device=gy.device)
    ops.wgrad_reduce(slab, gw, gb, K, 1.0)
    dX = torch.zeros(K, ld, device=gy.device)
    ops.linear_dgrad(weight.detach().t().contiguous(), dZ, None, dX, B, _lib.NIC_ACT_NONE, False)
    return ref_view(dX, B), gw, (gb if has_bias else torch.zeros(0, device=gy.device))


@linear_backward.register_fake
def _(gy, x, y, weight, act, has_bias):
    B, K = x.shape
    N = weight.shape[0]
    return (x.new_empty_strided((B, K), (1, pad_ld(B)), dtype=torch.float32), weight.new_empty((N, K), dtype=torch.float32),
            weight.new_empty((N if has_bias else 0,), dtype=torch.float32))


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act = inputs
    ctx.act, ctx.has_bias = act, bias is not None
    ctx.save_for_backward(x, output, weight)


def _linear_backward(ctx, gy):
    x, y, weight = ctx.saved_tensors
    gx, gw, gb = torch.ops.nic.linear_backward(gy, x, y, weight, ctx.act, ctx.has_bias)
    return gx, gw, (gb if ctx.has_bias else None), None


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ---- nic::env_step -------------------------------------------------------------------------------------------------------------

@torch.library.custom_op("nic::env_step", mutates_args=())
def env_step(store: torch.Tensor, wh: Optional[torch.Tensor], ech: Optional[torch.Tensor], a_store: torch.Tensor,
             a_wh: Optional[torch.Tensor], a_ech: Optional[torch.Tensor], demand: torch.Tensor,
             problem: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """One period (environment.py:110-299).  store / wh / ech: scenario-minor pipelines [loc][slots][ldb]; a_*: the reference's
    (B, loc, suppliers) action tensors; demand [S][ldb]: this period's row of the demand trace; problem: `register_problem` handle.
    Returns the next pipelines (an empty tensor where the setting has none) and the period cost [ldb]."""
    prob = _PROBLEMS[problem]
    t_store = Table.from_orders(a_store)
    t_wh = Table.from_orders(a_wh[:, :, 0]) if a_wh is not None else None
    t_ech = Table.from_orders(a_ech[:, :, 0]) if a_ech is not None else None
    out, reward = ops.env_step_fwd(prob, EnvState(store, wh, ech), Table(demand, prob.ldb, 1), t_store, t_wh, t_ech)
    return (out.store, out.wh if out.wh is not None else store.new_empty(0), out.ech if out.ech is not None else store.new_empty(0),
            reward)


@env_step.register_fake
def _(store, wh, ech, a_store, a_wh, a_ech, demand, problem):
    return (torch.empty_like(store), torch.empty_like(wh) if wh is not None else store.new_empty(0),
            torch.empty_like(ech) if ech is not None else store.new_empty(0), store.new_empty(store.shape[-1]))


@torch.library.custom_op("nic::env_step_backward", mutates_args=())
def env_step_backward(g_store: Optional[torch.Tensor], g_wh: Optional[torch.Tensor], g_ech: Optional[torch.Tensor],
                      g_reward: Optional[torch.Tensor], store: torch.Tensor, wh: Optional[torch.Tensor],
                      ech: Optional[torch.Tensor], a_store: torch.Tensor, a_wh: Optional[torch.Tensor],
                      a_ech: Optional[torch.Tensor], demand: torch.Tensor,
                      problem: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Adjoint of nic::env_step: gradients of (store, wh, ech, a_store, a_wh, a_ech); empty tensors where an input is absent."""
    prob = _PROBLEMS[problem]
    B = prob.B

    def dense(g):
        return None if g is None else g.contiguous()
    if g_reward is None:
        g_reward = torch.zeros(prob.ldb, device=store.device)
    g_in, (g_as, g_aw, g_ae) = ops.env_step_bwd(
        prob, EnvState(store, wh, ech), Table(demand, prob.ldb, 1), Table.from_orders(a_store),
        Table.from_orders(a_wh[:, :, 0]) if a_wh is not None else None,
        Table.from_orders(a_ech[:, :, 0]) if a_ech is not None else None,
        EnvState(dense(g_store), dense(g_wh), dense(g_ech)), Table(g_reward.contiguous(), 0, 1))
    e = lambda: store.new_empty(0)  # noqa: E731  (a fresh one each: operator outputs may not alias each other)
    return (g_in.store, g_in.wh if wh is not None else e(), g_in.ech if ech is not None else e(), ref_view(g_as, B),
            ref_view(g_aw, B).unsqueeze(2) if a_wh is not None else e(), ref_view(g_ae, B).unsqueeze(2) if a_ech is not None else e())


@env_step_backward.register_fake
def _(g_store, g_wh, g_ech, g_reward, store, wh, ech, a_store, a_wh, a_ech, demand, problem):
    e = lambda: store.new_empty(0)  # noqa: E731
    ld = store.shape[-1]

    def like_action(a):
        return a.new_empty_strided(a.shape, (1, ld * a.shape[2], ld) if a.dim() == 3 else (1, ld), dtype=torch.float32)
    return (torch.empty_like(store), torch.empty_like(wh) if wh is not None else e(), torch.empty_like(ech) if ech is not None else e(),
            like_action(a_store), a_wh.new_empty_strided(a_wh.shape, (1, ld, 1), dtype=torch.float32) if a_wh is not None else e(),
            a_ech.new_empty_strided(a_ech.shape, (1, ld, 1), dtype=torch.float32) if a_ech is not None else e())


def _env_setup(ctx, inputs, output):
    store, wh, ech, a_store, a_wh, a_ech, demand, problem = inputs
    ctx.problem = problem
    ctx.has = (wh is not None, ech is not None, a_wh is not None, a_ech is not None)
    ctx.save_for_backward(*[x for x in (store, wh, ech, a_store, a_wh, a_ech, demand) if x is not None])


def _env_backward(ctx, g_store, g_wh, g_ech, g_reward):
    has_wh, has_ech, has_awh, has_aech = ctx.has
    saved = list(ctx.saved_tensors)
    store = saved.pop(0)
    wh = saved.pop(0) if has_wh else None
    ech = saved.pop(0) if has_ech else None
    a_store = saved.pop(0)
    a_wh = saved.pop(0) if has_awh else None
    a_ech = saved.pop(0) if has_aech else None
    demand = saved.pop(0)
    gs, gw, ge, gas, gaw, gae = torch.ops.nic.env_step_backward(
        g_store, g_wh if has_wh else None, g_ech if has_ech else None, g_reward, store, wh, ech, a_store, a_wh, a_ech, demand,
        ctx.problem)
    return (gs, gw if has_wh else None, ge if has_ech else None, gas, gaw if has_awh else None, gae if has_aech else None,
            None, None)


env_step.register_autograd(_env_backward, setup_context=_env_setup)
