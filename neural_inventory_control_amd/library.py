"""`torch.library` registration of the engine's plugin-facing operators (SURVEY §8b names them):

    nic::linear(x, weight, bias?, act) -> y            the policy layers' Linear(+ELU) on csrc/linear_mfma.hip (= "mlp_layer_fwd";
                                                       nic::linear_backward = "mlp_layer_bwd")
    nic::env_step(store, wh?, ech?, a_store, a_wh?, a_ech?, demand, problem) -> (store', wh', ech', reward)
                                                       one period of Simulator.step on csrc/env_step.hip
    nic::softmax_alloc(z, wh_inv, adjacency, ub, transshipment, S, Wn) -> (store orders, warehouse orders)
                                                       the warehouse policies' feasibility head on csrc/policy_heads.hip
    nic::rollout_closed_form(levels, demand, state0, problem, policy, T, t0, ignore, round) -> (total, reported, d total / d levels)
                                                       whole horizon of a closed-form policy on csrc/closed_form.hip
    nic::sample_demand(mean, std?, rho, T, n, offset, seed, clip, poisson) -> demand [T][S][ldb]
                                                       the batched demand sampler on csrc/sampler.hip (not differentiable)

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


# ---- nic::softmax_alloc: the warehouse policies' feasibility head --------------------------------------------------------------

@torch.library.custom_op("nic::softmax_alloc", mutates_args=())
def softmax_alloc(z: torch.Tensor, wh_inv: torch.Tensor, adjacency: torch.Tensor, upper_bound: torch.Tensor, transshipment: bool,
                  n_stores: int, n_warehouses: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`apply_softmax_feasibility_function` + the warehouse's own sigmoid order (neural_networks.py:140-166, :403-426) on
    csrc/policy_heads.hip: z (B, S * Wn + Wn) logits, wh_inv (B, Wn, Ww) warehouse pipelines, adjacency [Wn][S] int32,
    upper_bound = the policy's `warehouse_upper_bound` tensor (a launch argument: read on the host once per tensor version, as the
    eager head does - a TENSOR here because a traced graph cannot hold a value read from the device) ->
    (store orders (B, S, Wn), warehouse orders (B, Wn, 1)), both views of scenario-minor buffers."""
    if not z.is_cuda:
        raise _lib.NicUnavailableError("nic::softmax_alloc needs device tensors (no CPU fallback)")
    from .neural_networks import _scalar
    upper_bound = _scalar(upper_bound)
    B, S, Wn = z.shape[0], n_stores, n_warehouses
    Z = _feature_major(z)
    ld = Z.stride(0)
    Wh = to_soa(wh_inv, ld)
    so, wo = torch.zeros(S, Wn, ld, device=z.device), torch.zeros(Wn, ld, device=z.device)
    ops.head_warehouse_fwd(Z, Wh, adjacency, upper_bound, transshipment, so, wo, S, Wn, wh_inv.shape[2], B)
    return ref_view(so, B), ref_view(wo, B).unsqueeze(2)


@softmax_alloc.register_fake
def _(z, wh_inv, adjacency, upper_bound, transshipment, n_stores, n_warehouses):
    B, ld = z.shape[0], pad_ld(z.shape[0])
    return (z.new_empty_strided((B, n_stores, n_warehouses), (1, n_warehouses * ld, ld), dtype=torch.float32),
            z.new_empty_strided((B, n_warehouses, 1), (1, ld, 1), dtype=torch.float32))


@torch.library.custom_op("nic::softmax_alloc_backward", mutates_args=())
def softmax_alloc_backward(g_store: torch.Tensor, g_wh: torch.Tensor, z: torch.Tensor, wh_inv: torch.Tensor, adjacency: torch.Tensor,
                           upper_bound: torch.Tensor, transshipment: bool, n_stores: int,
                           n_warehouses: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d z (B, S * Wn + Wn), d wh_inv (B, Wn, Ww)) of nic::softmax_alloc."""
    from .neural_networks import _scalar
    upper_bound = _scalar(upper_bound)
    B, S, Wn, Ww = z.shape[0], n_stores, n_warehouses, wh_inv.shape[2]
    Z = _feature_major(z)
    ld = Z.stride(0)
    Wh = to_soa(wh_inv, ld)
    Gs, Gw = to_soa(g_store, ld), to_soa(g_wh[:, :, 0], ld)
    dZ, g_inv = torch.zeros(S * Wn + Wn, ld, device=z.device), torch.zeros(Wn, Ww, ld, device=z.device)
    ops.head_warehouse_bwd(Z, Wh, adjacency, upper_bound, transshipment, Gs, Gw, dZ, g_inv, S, Wn, Ww, B)
    return ref_view(dZ, B), ref_view(g_inv, B)


@softmax_alloc_backward.register_fake
def _(g_store, g_wh, z, wh_inv, adjacency, upper_bound, transshipment, n_stores, n_warehouses):
    B, ld = z.shape[0], pad_ld(z.shape[0])
    Wn, Ww = wh_inv.shape[1], wh_inv.shape[2]
    return (z.new_empty_strided(z.shape, (1, ld), dtype=torch.float32),
            z.new_empty_strided((B, Wn, Ww), (1, Ww * ld, ld), dtype=torch.float32))


def _alloc_setup(ctx, inputs, output):
    z, wh_inv, adjacency, ub, trans, S, Wn = inputs
    ctx.meta = (trans, S, Wn)
    ctx.save_for_backward(z, wh_inv, adjacency, ub)


def _alloc_backward(ctx, g_so, g_wo):
    z, wh_inv, adjacency, ub = ctx.saved_tensors
    trans, S, Wn = ctx.meta
    dz, g_inv = torch.ops.nic.softmax_alloc_backward(g_so, g_wo, z, wh_inv, adjacency, ub, trans, S, Wn)
    return dz, g_inv, None, None, None, None, None


softmax_alloc.register_autograd(_alloc_backward, setup_context=_alloc_setup)


# ---- nic::rollout_closed_form: whole horizon of a closed-form policy, with the forward-mode level gradient -----------------------

@torch.library.custom_op("nic::rollout_closed_form", mutates_args=())
def rollout_closed_form(levels: torch.Tensor, demand: torch.Tensor, state0: torch.Tensor, problem: int, policy: int, periods: int,
                        first_period: int, ignore_periods: int,
                        round_orders: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """`Trainer.simulate_batch` (trainer.py:181-216) for base_stock / capped_base_stock / echelon_stock (policy = _lib.NIC_CF_*;
    neural_networks.py:216-311) as ONE launch of csrc/closed_form.hip: levels [n_levels] (what the policy's tiny net outputs),
    demand [T_total][S][ldb], state0 [S][F][ldb] (closed_form.pack_state0), problem = `register_problem` handle ->
    (total cost, cost from period `ignore_periods` on, d total / d levels)."""
    from . import closed_form
    prob = _PROBLEMS[problem]
    name = {v: k for k, v in closed_form.POLICY_ID.items()}[policy]
    lv = levels.detach().float().contiguous()
    desc = closed_form.make_desc(prob, name, periods, first_period, ignore_periods, lv, demand, state0, round_orders)
    ng = 0 if round_orders else lv.numel()   # (rounded orders have no gradient)
    partial = torch.empty(_lib.lib().nic_closed_form_num_partials(prob.B, prob.S), ng + 2, device=levels.device)
    _lib.check(_lib.lib().nic_closed_form_rollout_sums(desc, None, None, None, _lib.ptr(partial), ng + 2, int(ng > 0), 1,
                                                       _lib.current_stream()))
    sums = partial.sum(dim=0)
    return sums[ng].clone(), sums[ng + 1].clone(), (sums[:ng].clone() if ng else torch.zeros_like(lv))


@rollout_closed_form.register_fake
def _(levels, demand, state0, problem, policy, periods, first_period, ignore_periods, round_orders):
    return (levels.new_empty((), dtype=torch.float32), levels.new_empty((), dtype=torch.float32),
            levels.new_empty(levels.shape, dtype=torch.float32))


def _cf_setup(ctx, inputs, output):
    ctx.save_for_backward(output[2])


def _cf_backward(ctx, g_total, g_reported, g_glevels):
    (g_levels,) = ctx.saved_tensors
    return (g_total * g_levels,) + (None,) * 8


rollout_closed_form.register_autograd(_cf_backward, setup_context=_cf_setup)


# ---- nic::sample_demand: the batched demand sampler ------------------------------------------------------------------------------

@torch.library.custom_op("nic::sample_demand", mutates_args=())
def sample_demand(mean: torch.Tensor, std: Optional[torch.Tensor], correlation: float, periods: int, n_scenarios: int,
                  scenario_offset: int, seed: int, clip: bool, poisson: bool) -> torch.Tensor:
    """`Scenario.generate_normal_demand` / `generate_poisson_demand` (data_handling.py:178-211) on csrc/sampler.hip: mean [S]
    (std [S] for normal demand, pairwise correlation in [0, 1]) -> demand trace [T][S][ldb], Philox keyed by the GLOBAL scenario
    index (scenario_offset + column) so shards reproduce the single-process trace.  Not differentiable."""
    if not mean.is_cuda:
        raise _lib.NicUnavailableError("nic::sample_demand needs device tensors (no CPU fallback)")
    S = mean.numel()
    out = torch.zeros(periods, S, pad_ld(n_scenarios), device=mean.device)
    if poisson:
        ops.sample_demand(out, periods, S, n_scenarios, scenario_offset, seed, 1, mean.float().contiguous(), None, clip)
    else:
        ops.sample_demand_equicorrelated(out, periods, S, n_scenarios, scenario_offset, seed, mean.float().contiguous(),
                                         std.float().contiguous(), correlation if S > 1 else 0.0, clip)
    return out


@sample_demand.register_fake
def _(mean, std, correlation, periods, n_scenarios, scenario_offset, seed, clip, poisson):
    return mean.new_empty((periods, mean.numel(), pad_ld(n_scenarios)), dtype=torch.float32)
