"""Whole-horizon rollout of the closed-form policies — `base_stock`, `capped_base_stock`, `echelon_stock`
(neural_networks.py:216-229, 296-311, 231-294 of the reference) — through `nic_closed_form_rollout`
(csrc/closed_form.hip): ONE kernel runs all T periods of `Trainer.simulate_batch` (trainer.py:190-213) AND produces the
gradient, by carrying the derivative of every pipeline slot with respect to the policy's 1 .. E+2 "levels" in registers
(forward mode; no activation history, no backward kernel).  The levels are what the reference's tiny `net` outputs after
its activation; that part stays in torch, so the returned total is an ordinary differentiable tensor:
`(total / n).backward()` reaches `net.master.0.bias` exactly as upstream.

Descriptor building is pointer plumbing and device-agnostic (the CPU test build of the kernel body uses it too).
"""
import torch

from . import _lib
from ._lib import NicClosedFormDesc, NicTable2
from .layout import EnvProblem, ProblemCache, demand_trace_soa, pad_ld

POLICY_ID = {"base_stock": _lib.NIC_CF_BASE_STOCK, "capped_base_stock": _lib.NIC_CF_CAPPED,
             "echelon_stock": _lib.NIC_CF_ECHELON}


def supports_shapes(name, prob: EnvProblem):
    if name not in POLICY_ID or max(prob.Ws, prob.Ww, prob.We) > _lib.NIC_MAX_SLOTS:
        return False
    if name == "echelon_stock":
        return prob.S == 1 and prob.Wn == 1 and 1 <= prob.E <= 3
    return prob.Wn == 0 and prob.E == 0


def make_desc(prob: EnvProblem, name, T, t0, ignore_periods, levels, demand_soa, state0, round_orders=False):
    """levels: device float32 [n_levels]; demand_soa: [T_total][S][ldb]; state0: [S][F][ldb]."""
    d = NicClosedFormDesc()
    d.n_scenarios, d.ldb, d.T, d.t0, d.ignore_periods = prob.B, prob.ldb, T, t0, ignore_periods
    d.policy, d.n_levels = POLICY_ID[name], int(levels.numel())
    d.S, d.Ws, d.Wn, d.Ww, d.E, d.We = prob.S, prob.Ws, prob.Wn, prob.Ww, prob.E, prob.We
    d.lost_demand, d.maximize_profit = int(prob.lost_demand), int(prob.maximize_profit)
    d.round_orders = int(bool(round_orders))
    d.levels, d.demand, d.state0 = levels.data_ptr(), demand_soa.data_ptr(), state0.data_ptr()
    d.underage, d.holding = prob.underage.t2(), prob.holding.t2()
    lead = prob.lead  # (s, supplier, b) table with a single supplier column
    d.lead = NicTable2(_lib.ptr(lead.tensor), lead.loc_stride, lead.scn_stride)
    d.wh_holding, d.wh_lead, d.wh_edge = prob.wh_holding.t2(), prob.wh_lead.t2(), prob.wh_edge.t2()
    d.ech_holding, d.ech_lead = prob.ech_holding.t2(), prob.ech_lead.t2()
    d._keep = (levels, demand_soa, state0, prob)
    return d


def pack_state0(data, prob: EnvProblem, out=None):
    """Initial pipelines -> [S][F][ldb] (store slots, then the warehouse's, then the echelons': the reference's cat order)."""
    F = prob.Ws + prob.Wn * prob.Ww + prob.E * prob.We
    dev = data["initial_inventories"].device
    if out is None:
        out = torch.zeros(prob.S, F, prob.ldb, device=dev)
    B = prob.B
    out[:, :prob.Ws, :B] = data["initial_inventories"].permute(1, 2, 0)
    if prob.Wn:
        out[0, prob.Ws:prob.Ws + prob.Ww, :B] = data["initial_warehouse_inventories"][:, 0].t()
    if prob.E:
        o = prob.Ws + prob.Ww
        out[0, o:o + prob.E * prob.We, :B] = data["initial_echelon_inventories"].reshape(B, -1).t()
    return out


class _ClosedFormTotal(torch.autograd.Function):
    """(levels) -> (total, reported): the kernel launch; backward scales the forward-mode level gradients."""

    @staticmethod
    def forward(ctx, levels, eng, desc_args, want_grad):
        total, reported, g_levels = eng._launch(levels.detach().float().contiguous(), desc_args, want_grad)
        ctx.g_levels = g_levels
        ctx.mark_non_differentiable(reported)
        return total, reported

    @staticmethod
    def backward(ctx, g_total, _g_reported):
        if ctx.g_levels is None:
            return None, None, None, None
        return g_total * ctx.g_levels, None, None, None


class ClosedFormRollout:
    """Engine behind `Trainer.simulate_batch` for the closed-form policies (one per policy and train / eval mode)."""

    @staticmethod
    def supports(model):
        name = getattr(model, "nn_args", {}).get("name") if hasattr(model, "nn_args") else None
        return name in POLICY_ID and type(model).__name__ in ("BaseStock", "CappedBaseStock", "EchelonStock") \
            and hasattr(model, "closed_form_levels")

    def __init__(self, model, problem_params, device):
        _lib.require_device()
        if not self.supports(model):
            raise ValueError("ClosedFormRollout handles base_stock / capped_base_stock / echelon_stock")
        self.model, self.problem_params, self.device = model, problem_params, torch.device(device)
        self.name = model.nn_args["name"]
        self.keep_rewards = False   # per-period rewards [T][S][ldb] (inspection / tests); the trainer only needs the totals
        self.keep_chain_totals = False   # per-chain totals [2][S][ldb] (inspection); the step's two sums come from per-wavefront partials
        self.timer = None
        self._probs = ProblemCache()
        self._key = None

    def shapes_ok(self, data):
        return supports_shapes(self.name, self._probs.get(self.problem_params, data, self.device))

    def _setup(self, prob, T):
        key = (prob.B, T, prob.S, prob.Ws, prob.Wn, prob.Ww, prob.E, prob.We, self.keep_rewards, self.keep_chain_totals)
        if key == self._key:
            return
        dev, ld = self.device, prob.ldb
        F = prob.Ws + prob.Wn * prob.Ww + prob.E * prob.We
        self.state0 = torch.zeros(prob.S, F, ld, device=dev)
        self.state_final = torch.zeros(prob.S, F, ld, device=dev)
        self.totals = torch.zeros(2, prob.S, ld, device=dev) if self.keep_chain_totals else None
        self.rewards = torch.zeros(T, prob.S, ld, device=dev) if self.keep_rewards else None
        self.n_partials = _lib.lib().nic_closed_form_num_partials(prob.B, prob.S)
        self._key = key

    def run(self, data, periods, ignore_periods=0, train=True, observation_params=None, demand_soa=None,
            discrete_allocation=False):
        """Returns (total, reported) = `simulate_batch`'s return values (trainer.py:216); with `train` and autograd
        recording, `total` is differentiable with respect to the policy's parameters."""
        if discrete_allocation and train:
            raise ValueError("discrete_allocation is an evaluation-time option (rounded orders have zero gradient)")
        prob = self._probs.get(self.problem_params, data, self.device)
        if not supports_shapes(self.name, prob):
            raise ValueError(f"{self.name}: setting outside the fused closed-form kernel (see nic_rollout.h)")
        T, B, ld = periods, prob.B, prob.ldb
        self._setup(prob, T)
        self.prob = prob
        shift = observation_params["demand"]["period_shift"] if observation_params else 0
        if demand_soa is None:
            demand_soa = demand_trace_soa(data["demands"], ld, self.device)
        if demand_soa.shape[0] < T + shift:
            raise ValueError("Current period is greater than the number of periods in the data")
        pack_state0(data, prob, self.state0)
        want_grad = bool(train) and torch.is_grad_enabled()
        levels = self.model.closed_form_levels()
        args = (prob, T, shift, ignore_periods, demand_soa, bool(discrete_allocation))
        if want_grad:
            return _ClosedFormTotal.apply(levels, self, args, True)
        total, reported, _ = self._launch(levels.detach().float().contiguous(), args, False)
        return total, reported

    def _launch(self, levels, args, want_grad):
        prob, T, shift, ignore, demand_soa, rounded = args
        desc = make_desc(prob, self.name, T, shift, ignore, levels, demand_soa, self.state0, rounded)
        # one row per wavefront of chains: [d total / d level_j ...][total, reported]; ONE small sum gives the step's numbers (the
        # reduction of per-chain totals - 2 x 10^6 floats into 2 - was the longest launch of the 10^6-chain step)
        ng = levels.numel() if want_grad else 0
        partial = torch.empty(self.n_partials, ng + 2, device=self.device)
        call = lambda: _lib.check(_lib.lib().nic_closed_form_rollout_sums(  # noqa: E731
            desc, _lib.ptr(self.rewards), _lib.ptr(self.totals), _lib.ptr(self.state_final), _lib.ptr(partial), ng + 2,
            int(want_grad), 1, _lib.current_stream()))
        if self.timer is not None:
            self.timer.call("closed_form_fwd", call)
        else:
            call()
        sums = partial.sum(dim=0)
        return sums[ng], sums[ng + 1], (sums[:ng] if want_grad else None)

    # ---- inspection helpers used by the parity tests ------------------------------------------------------------------
    def per_period_rewards(self):
        return self.rewards[:, :, :self.prob.B].sum(dim=1)

    def final_state(self):
        p, B = self.prob, self.prob.B
        out = {"store_inventories": self.state_final[:, :p.Ws, :B].permute(2, 0, 1)}
        if p.Wn:
            out["warehouse_inventories"] = self.state_final[0, p.Ws:p.Ws + p.Ww, :B].t().unsqueeze(1)
        if p.E:
            o = p.Ws + p.Ww
            out["echelon_inventories"] = self.state_final[0, o:o + p.E * p.We, :B].t().reshape(B, p.E, p.We)
        return out
