"""Whole-horizon rollout of the small (32-wide, one-store chain) policies through `nic_small_rollout_fwd/bwd`
(csrc/small_rollout.hip): ONE kernel runs all T periods of `Trainer.simulate_batch` (trainer.py:190-213), one lane per
scenario with pipelines and activations in registers; the backward sweep is a second kernel plus one weight-gradient GEMM
per layer contracting over (period x scenario).  Used by `FusedRollout` when `SmallRollout.supports(...)`.

Descriptor building is pointer plumbing and device-agnostic (the CPU test build of the kernel bodies uses it too).
"""
import torch

from . import _lib, ops
from ._lib import NicSmallRolloutDesc, NicTable2
from .layout import EnvProblem, Table

HEAD_SOFTPLUS, HEAD_SERIAL = 0, 1
H = _lib.NIC_SR_HIDDEN


def packed_weight_count(F, n_hidden, n_out):
    return (H * F + H) + (n_hidden - 1) * (H * H + H) + (n_out * H + n_out)


def pack_weights(linears, out=None):
    """[W1 (32 x F), b1] [W_l (32 x 32), b_l]... [Wout (n_out x 32), bout] as one flat float32 tensor."""
    parts = []
    for m in linears:
        parts += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
    if out is None:
        return torch.cat(parts).contiguous()
    if out.numel() != sum(p.numel() for p in parts):   # (torch.cat would silently re-allocate `out`: kernels hold its address)
        raise ValueError("pack_weights: the output buffer does not have packed_weight_count elements")
    torch.cat(parts, out=out)   # (one launch)
    return out


def layer_slices(F, n_hidden, n_out):
    """[(weight offset, rows, cols, bias offset)] per layer in the packed buffer."""
    out, off, k = [], 0, F
    for i in range(n_hidden + 1):
        n = H if i < n_hidden else n_out
        out.append((off, n, k, off + n * k))
        off += n * k + n
        k = n
    return out


class SmallRolloutPlan:
    """Shapes of one small-policy rollout; `supports` says whether the fused kernels apply."""

    def __init__(self, prob: EnvProblem, head, dims):
        self.prob, self.head, self.dims = prob, head, list(dims)
        self.F = prob.Ws + prob.Wn * prob.Ww + prob.E * prob.We
        self.n_hidden = len(dims) - 2
        self.n_out = dims[-1]

    @staticmethod
    def supports(prob: EnvProblem, head, dims):
        F = prob.Ws + prob.Wn * prob.Ww + prob.E * prob.We
        n_hidden = len(dims) - 2
        return (head in ("softplus", "serial") and prob.S == 1 and prob.Wn <= 1 and prob.E <= 3
                and (prob.Wn == 1 or prob.E == 0) and F <= _lib.NIC_SR_MAX_INPUTS and dims[0] == F
                and 1 <= n_hidden <= 3 and all(w == H for w in dims[1:-1]) and dims[-1] <= _lib.NIC_SR_MAX_OUTPUTS
                and (head != "softplus" or dims[-1] == 1) and (head != "serial" or dims[-1] == prob.E + 2))

    def desc(self, T, t0, weights, demand_soa, state0, upper_bound, round_orders=False, prob=None, lane_scenarios=0):
        """demand_soa: [T_total][1][ldb]; state0: [F][ldb]; prob: the CURRENT batch's EnvProblem (same shapes as the
        plan's; its cost / lead-time tables are the ones the kernels read)."""
        p = prob if prob is not None else self.prob
        d = NicSmallRolloutDesc()
        d.n_scenarios, d.ldb, d.T, d.t0 = p.B, p.ldb, T, t0
        d.F, d.n_hidden, d.n_out = self.F, self.n_hidden, self.n_out
        d.head = HEAD_SERIAL if self.head == "serial" else HEAD_SOFTPLUS
        d.Ws, d.Wn, d.Ww, d.E, d.We = p.Ws, p.Wn, p.Ww, p.E, p.We
        d.lost_demand, d.maximize_profit = int(p.lost_demand), int(p.maximize_profit)
        d.detach_input = int(self.head == "serial")
        d.round_orders = int(bool(round_orders))
        d.upper_bound = float(upper_bound)
        d.lane_scenarios = int(lane_scenarios)   # 0: the library picks 16 or 32 scenarios per wavefront from the batch size
        d.weights, d.demand, d.state0 = weights.data_ptr(), demand_soa.data_ptr(), state0.data_ptr()
        d.underage, d.holding = p.underage.t2(), p.holding.t2()
        lead = p.lead  # (s, w, b) table with one store and one supplier column
        d.lead = NicTable2(_lib.ptr(lead.tensor), lead.loc_stride, lead.scn_stride)
        d.wh_holding, d.wh_lead, d.wh_edge = p.wh_holding.t2(), p.wh_lead.t2(), p.wh_edge.t2()
        d.ech_holding, d.ech_lead = p.ech_holding.t2(), p.ech_lead.t2()
        self._keep = (weights, demand_soa, state0, p)
        return d


def small_rollout_fwd(desc, rewards, state_final, states_hist, hidden_hist, logits_hist):
    ops._dev(rewards)
    _lib.check(_lib.lib().nic_small_rollout_fwd(desc, _lib.ptr(rewards), _lib.ptr(state_final), _lib.ptr(states_hist),
                                                _lib.ptr(hidden_hist), _lib.ptr(logits_hist), _lib.current_stream()))


def small_rollout_bwd(desc, states_hist, hidden_hist, logits_hist, g_reward: Table, dz_hidden, dz_out):
    ops._dev(dz_out)
    _lib.check(_lib.lib().nic_small_rollout_bwd(desc, _lib.ptr(states_hist), _lib.ptr(hidden_hist), _lib.ptr(logits_hist),
                                                g_reward.t2(), _lib.ptr(dz_hidden), _lib.ptr(dz_out),
                                                _lib.current_stream()))


def small_rollout_bwd_wgrad_slots(n_scenarios):
    return _lib.lib().nic_small_rollout_bwd_wgrad_slots(int(n_scenarios))


def small_rollout_bwd_wgrad(desc, states_hist, hidden_hist, logits_hist, g_reward: Table, slab):
    """Backward sweep with in-kernel weight gradients: slab [slots][>= packed_weight_count] receives one partial gradient per
    wavefront in the packed-weight layout (sum over dim 0 = d total / d packed weights)."""
    ops._dev(slab)
    _lib.check(_lib.lib().nic_small_rollout_bwd_wgrad(desc, _lib.ptr(states_hist), _lib.ptr(hidden_hist), _lib.ptr(logits_hist),
                                                      g_reward.t2(), _lib.ptr(slab), slab.stride(0), _lib.current_stream()))


def small_rollout_reduce_scratch(n_rows, P, n_reward_elems):
    return int(_lib.lib().nic_small_rollout_reduce_scratch(int(n_rows), int(P), int(n_reward_elems)))


def small_rollout_reduce(slab, n_rows, grad, rewards, ignore_periods, totals, scratch):
    """grad <- sum of the first n_rows rows of slab; totals <- [sum of rewards [T][ldb], sum of its periods >= ignore_periods]:
    two launches with a fixed summation order (csrc/small_reduce.hip).  Either pair may be None."""
    ops._dev(scratch)
    P = grad.numel() if grad is not None else 0
    n_el = rewards.numel() if rewards is not None else 0
    assert scratch.numel() >= small_rollout_reduce_scratch(n_rows if slab is not None else 0, P, n_el)
    assert rewards is None or rewards.is_contiguous()
    _lib.check(_lib.lib().nic_small_rollout_reduce(_lib.ptr(slab), int(n_rows), slab.stride(0) if slab is not None else 0, P,
                                                   _lib.ptr(grad), _lib.ptr(rewards), n_el,
                                                   int(ignore_periods) * rewards.shape[-1] if rewards is not None else 0,
                                                   _lib.ptr(totals), _lib.ptr(scratch), _lib.current_stream()))
