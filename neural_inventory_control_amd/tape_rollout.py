"""`TapeRollout`: whole-horizon rollout for the policies whose decisions do not need a network INSIDE the period loop.

  * The quantile policies - `transformed_nv`, `fixed_quantile`, `quantile_nv`, `returns_nv` (QuantilePolicy, neural_networks.py:
    517-631 of the reference) - order up to a level that the frozen quantile forecaster reads off the past-demand window and the
    policy's desired quantile: the level of period t depends on the demand trace and on the policy's parameters, NOT on the state.
    All T levels are therefore computed in one batched pass (the forecaster runs once over T x B x S rows instead of T times), and
    the only sequential part - order = clip(level - pipeline total, 0), the env step - runs in ONE launch of
    `nic_horizon_rollout_fwd` (head_mode 2); its adjoint, one launch of `nic_horizon_rollout_bwd`, returns d loss / d level for
    every (period, scenario, store), which autograd carries on through the interpolation into the policy's parameters.
  * `just_in_time` (neural_networks.py:634-739) reads FUTURE demand and nothing else: all T orders are a batched gather, the
    rollout is one launch of the forward kernel on an order tape (head_mode 1); not trainable.

`Trainer.simulate_batch` (trainer.py:181-216) otherwise walks these policies period by period: ~25 launches per period for the
forecaster, the interpolation and the env step.  The returned total is an ordinary differentiable tensor, so the caller's
`mean_loss.backward()` works as it does upstream.
"""
import torch

from . import _lib
from . import horizon_rollout as hz
from ._lib import NicHorizonDesc
from .layout import demand_trace_soa, ProblemCache, Table

_QUANTILE = ("TransformedNV", "QuantileNV", "ReturnsNV", "FixedQuantile")


class _TapeTotal(torch.autograd.Function):
    """(levels [T][B][S]) -> (total, reported); backward = the whole-horizon adjoint's d total / d level."""

    @staticmethod
    def forward(ctx, levels, eng, args):
        total, reported = eng._forward(levels.detach(), args, train=True)
        ctx.eng, ctx.args, ctx.dtype, ctx.run_id = eng, args, levels.dtype, eng._run_id
        ctx.mark_non_differentiable(reported)
        return total, reported

    @staticmethod
    def backward(ctx, g_total, _g_reported):
        if ctx.run_id != ctx.eng._run_id:   # the histories the adjoint reads belong to the engine, not to this graph
            raise RuntimeError("TapeRollout: backward of a rollout whose histories a later run of the same engine has overwritten "
                               "(call backward before the next simulate_batch, as Trainer.do_one_epoch does)")
        return ctx.eng._backward(g_total, ctx.args).to(ctx.dtype, copy=True), None, None


class TapeRollout:
    @staticmethod
    def supports(model):
        kind = type(model).__name__
        if kind == "JustInTime":
            return True
        return kind in _QUANTILE and "quantile_forecaster" in getattr(model, "fixed_nets", {})

    @staticmethod
    def observation_ok(model, observation_params, data=None):
        op = observation_params
        if op is None:
            return False
        if type(model).__name__ == "JustInTime":
            return True
        past = (op["demand"] or {}).get("past_periods")
        tf = op["time_features"]
        return (isinstance(past, int) and past >= 1 and isinstance(tf, (list, tuple)) and "days_from_christmas" in tf
                and (data is None or ("days_from_christmas" in data and data["days_from_christmas"].shape[1] == 1)))

    def __init__(self, model, problem_params, device):
        _lib.require_device()
        if not self.supports(model):
            raise ValueError("TapeRollout handles the quantile policies and just_in_time")
        self.model, self.problem_params, self.device = model, problem_params, torch.device(device)
        self.jit = type(model).__name__ == "JustInTime"
        self.timer = None
        self._probs = ProblemCache()
        self._key = None
        self._run_id = 0   # bumped by every forward: a backward checks that the histories are still its own

    def shapes_ok(self, data, periods=1, period_shift=0):
        p = self._probs.get(self.problem_params, data, self.device)
        if not hz.offsets_ok(p, periods, periods + period_shift):
            return False
        ok = p.E == 0 and p.S <= hz.MAX_STORES and p.S * p.Ws + p.Wn * p.Ww <= hz.MAX_STATE_ROWS and 2 <= p.Ws <= hz.MAX_SLOTS \
            and (p.Wn == 0 or 2 <= p.Ww <= hz.MAX_SLOTS) and p.S * p.nsup + p.Wn <= hz.MAX_OUT
        return ok and (self.jit or p.Wn == 0)

    def _k(self, tag, fn, *a):
        return fn(*a) if self.timer is None else self.timer.call(tag, fn, *a)

    def _setup(self, prob, T, train):
        key = (prob.B, T, bool(train), prob.S, prob.Ws, prob.Wn, prob.Ww)
        if key == self._key:
            return
        dev, ld = self.device, prob.ldb
        z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
        FD, n_ord = prob.S * prob.Ws + prob.Wn * prob.Ww, prob.S * prob.nsup + prob.Wn
        self.state0, self.state_final, self.rewards = z(FD, ld), z(FD, ld), z(T, ld)
        self.tape = z(n_ord if self.jit else prob.S, T, ld)
        if train:
            self.state_hist, self.orders_hist, self.dlevel = z(FD, T, ld), z(n_ord + prob.Wn, T, ld), z(prob.S, T, ld)
            self.g_reward = z(ld)
        if prob.Wn:
            conn = self.problem_params["warehouse_store_adjacency"]
            self.edge_mask = torch.tensor(conn, dtype=torch.float32, device=dev).t().contiguous()
        else:
            self.edge_mask = None
        self._key = key

    # ---- the tapes: every period's decision inputs in one batched pass ---------------------------------------------------
    def _levels(self, data, T, shift, observation_params):
        """[T][B][S] order-up-to levels = QuantilePolicy.forward's `levels` (neural_networks.py:560-588) for every period at once."""
        m, dem = self.model, data["demands"]
        B, S, n_t = dem.shape
        P_ = observation_params["demand"]["past_periods"]
        dpad = torch.cat([torch.zeros(B, S, P_, device=dem.device, dtype=dem.dtype), dem], dim=2)
        # window of period t = demands[t + shift - P, t + shift) with zeros left of the trace (environment.py:436-458)
        win = dpad.unfold(2, P_, 1)[:, :, shift:shift + T]                       # [B][S][T][P]
        dfc = data["days_from_christmas"]                                         # [B][1][periods of the data]
        idx = torch.clamp(torch.arange(T, device=dem.device) + shift, max=dfc.shape[2] - 1)
        xmas = dfc[:, 0, idx]                                                     # [B][T]
        x = torch.cat([win.permute(2, 0, 1, 3), xmas.t().reshape(T, B, 1, 1).expand(T, B, S, 1)], dim=3).reshape(T * B, S, P_ + 1)
        q = m.compute_desired_quantiles({"underage_costs": data["underage_costs"], "holding_costs": data["holding_costs"]})
        lead = data["lead_times"][:, :, 0]
        levels = m.fixed_nets["quantile_forecaster"].get_quantile(x, q.unsqueeze(0).expand(T, B, S).reshape(T * B, S),
                                                                  lead.unsqueeze(0).expand(T, B, S).reshape(T * B, S))
        return levels.reshape(T, B, S)

    def _jit_orders(self, data, prob, T, shift):
        """[T][B][S*nsup + Wn] orders = JustInTime.forward (neural_networks.py:634-739) for every period at once."""
        dem = data["demands"]
        B, S, n_t = dem.shape
        dev = dem.device
        cur = (torch.arange(T, device=dev) + shift).view(T, 1)                    # [T][1]

        rows = torch.arange(B, device=dev).view(1, B)

        def future(store, lt):   # demand of `store` at period t + shift + lt[b] (clipped to the trace), [T][B]
            return dem[:, store][rows, torch.clamp(cur + lt.long().view(1, B), max=n_t - 1)]
        if prob.Wn == 0:
            lt = data["lead_times"][:, :, 0]
            so = torch.stack([future(j, lt[:, j]) for j in range(S)], dim=2)      # [T][B][S]
            return torch.clip(so, min=0)
        lt, wlt = data["lead_times"], data["warehouse_lead_times"]
        adj = torch.tensor(self.problem_params["warehouse_store_adjacency"], dtype=torch.float32)
        mean_lt = lt.mean(dim=0).cpu()   # (the fastest connected warehouse per store, from the batch-mean lead times: :695-699)
        served, fastest = [], []
        for st in range(S):
            conn = adj[:, st].nonzero(as_tuple=True)[0]
            if len(conn):
                served.append(st)
                fastest.append(int(conn[torch.argmin(mean_lt[st, conn])]))
        alloc = torch.zeros(T, B, S, prob.Wn, device=dev)
        wh = torch.zeros(T, B, prob.Wn, device=dev)
        if served:
            sv, fw = torch.tensor(served, device=dev), torch.tensor(fastest, device=dev)
            lt_sel = lt[:, sv, fw].long()                                                  # [B][m] lead time from the chosen warehouse
            dem_t = dem[:, sv].unsqueeze(0).expand(T, B, len(served), n_t)                  # (a view: no copy)
            t_idx = cur.view(T, 1, 1)

            def future_all(extra):   # demand of every served store at period t + shift + extra[b, store] (clipped), [T][B][m]
                return dem_t.gather(3, torch.clamp(t_idx + extra.unsqueeze(0), max=n_t - 1).unsqueeze(3)).squeeze(3)
            alloc[:, :, sv, fw] = future_all(lt_sel)
            via_wh = future_all(wlt[:, fw].long() + lt_sel)
            for k, w in enumerate(fastest):   # (accumulated store by store, in store order, as upstream's `wh[:, w] += ...`)
                wh[:, :, w] += via_wh[:, :, k]
        return torch.cat([torch.clip(alloc, min=0).reshape(T, B, S * prob.Wn), torch.clip(wh, min=0)], dim=2)

    # ---- one batch ---------------------------------------------------------------------------------------------------
    def run(self, data, periods, ignore_periods=0, train=True, observation_params=None, discrete_allocation=False):
        """(total, reported) = `simulate_batch`'s return values; with `train`, autograd recording and a trainable policy `total` is
        differentiable with respect to the policy's parameters."""
        if discrete_allocation and train:
            raise ValueError("discrete_allocation is an evaluation-time option (rounded orders have zero gradient)")
        prob = self._probs.get(self.problem_params, data, self.device)
        T, B, ld = periods, prob.B, prob.ldb
        shift = observation_params["demand"]["period_shift"] if observation_params else 0
        d = data["demands"]
        if d.shape[2] < T + shift:
            raise ValueError("Current period is greater than the number of periods in the data")
        want_grad = bool(train) and torch.is_grad_enabled() and bool(getattr(self.model, "trainable", True)) and not self.jit
        self._setup(prob, T, want_grad)
        self.prob = prob
        demand_soa = demand_trace_soa(d, ld, self.device)
        a = prob.S * prob.Ws
        self.state0[:a].view(prob.S, prob.Ws, ld)[:, :, :B].copy_(data["initial_inventories"].permute(1, 2, 0))
        if prob.Wn:
            self.state0[a:].view(prob.Wn, prob.Ww, ld)[:, :, :B].copy_(data["initial_warehouse_inventories"].permute(1, 2, 0))
        args = (prob, T, shift, ignore_periods, demand_soa, bool(discrete_allocation))
        if self.jit:
            with torch.no_grad():
                tape = self._jit_orders(data, prob, T, shift)
            return self._forward(tape, args, train=False)
        if want_grad:
            return _TapeTotal.apply(self._levels(data, T, shift, observation_params), self, args)
        with torch.no_grad():
            levels = self._levels(data, T, shift, observation_params)
        return self._forward(levels, args, train=False)

    def _desc(self, args):
        prob, T, shift, _, demand_soa, rounded = args
        d = NicHorizonDesc()
        d.io = prob.make_io(None, None, None, Table.null(), Table.null(), None, None)
        d.T, d.t0, d.H1, d.H2 = int(T), int(shift), 0, 0
        d.n_out = prob.S * prob.nsup + prob.Wn
        d.round_orders = int(rounded)
        d.mask, d.demand = _lib.ptr(self.edge_mask), demand_soa.data_ptr()
        d.hist_stride = T * prob.ldb
        d.head_mode = 1 if self.jit else 2
        d.allow_negative = int(bool(getattr(self.model, "allow_back_orders", False)))
        d.tape = self.tape.data_ptr()
        self._keep = (prob, demand_soa)
        return d

    def _forward(self, tape_values, args, train):
        """tape_values: [T][B][rows] (levels or orders; any float dtype - the interpolation is float64 upstream, the env step's
        operands float32)."""
        prob, T, _, ignore, _, _ = args
        self._run_id += 1
        self.tape[:, :, :prob.B].copy_(tape_values.permute(2, 0, 1))
        desc = self._desc(args)
        hist = (self.state_hist, None, None, None, self.orders_hist) if train else (None,) * 5
        self._k("horizon_fwd", hz.horizon_fwd, desc, None, self.state0, self.rewards, self.state_final, *hist)
        total = self.rewards.sum()
        reported = self.rewards[ignore:].sum() if ignore else total
        return total, reported

    def _backward(self, g_total, args):
        prob, T = args[0], args[1]
        self.g_reward.zero_()
        self.g_reward[:prob.B] = g_total
        desc = self._desc(args)
        self._k("horizon_bwd", hz.horizon_bwd, desc, self.state_hist, None, None, None, self.orders_hist,
                Table(self.g_reward, 0, 1), None, None, self.dlevel)
        return self.dlevel[:, :, :prob.B].permute(1, 2, 0)

    # ---- inspection helpers used by the parity tests ------------------------------------------------------------------
    def per_period_rewards(self):
        return self.rewards[:, :self.prob.B]

    def final_state(self):
        from .layout import ref_view
        p = self.prob
        a = p.S * p.Ws
        out = {"store_inventories": ref_view(self.state_final[:a].view(p.S, p.Ws, -1), p.B)}
        if p.Wn:
            out["warehouse_inventories"] = ref_view(self.state_final[a:].view(p.Wn, p.Ww, -1), p.B)
        return out
