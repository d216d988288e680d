"""`Simulator`: the reference's differentiable inventory simulator (environment.py:7-169) behind the same Python API,
with the whole period (36-87 aten ops and 4-6 host syncs in the reference) executed by ONE HIP kernel
forward and ONE backward (csrc/env_step.hip) registered as a normal autograd function, so arbitrary user policies
(any nn.Module returning the reference's action dict) keep working and backpropagate through the horizon.

Differences that are deliberate and invisible to callers:
  * state lives in scenario-minor storage; `observation[...]` exposes the reference's (B, S, W) shapes as views;
  * the current period is a host integer — no `.item()` device syncs (environment.py:123,135) and no data-dependent
    branch (`non_zero_mask.any()`, environment.py:427): the `!= 0` filter is arithmetic inside the kernel;
  * no CPU path: `Simulator(device='cpu')` raises at reset (the CPU checker lives in oracle/, for tests only).
"""
import torch

from . import _lib, ops
from .layout import demand_trace_soa, EnvProblem, ProblemCache, Table, pad_ld, ref_view, to_soa
from .ops import EnvState


class _EnvStepFunction(torch.autograd.Function):
    """(state_t, orders_t) -> (state_{t+1}, reward_t) with the analytic backward of csrc/env_step_body.h."""

    @staticmethod
    def forward(ctx, prob, demand, store, wh, ech, a_store, a_wh, a_ech):
        st = EnvState(store, wh, ech)
        t_store = Table.from_orders(a_store)
        t_wh = Table.from_orders(a_wh[:, :, 0]) if a_wh is not None else None
        t_ech = Table.from_orders(a_ech[:, :, 0]) if a_ech is not None else None
        out, reward = ops.env_step_fwd(prob, st, demand, t_store, t_wh, t_ech)
        ctx.prob, ctx.demand = prob, demand
        ctx.has = (wh is not None, ech is not None, a_wh is not None, a_ech is not None)
        ctx.save_for_backward(*[x for x in (store, wh, ech, a_store, a_wh, a_ech) if x is not None])
        outs = (out.store, out.wh, out.ech, reward)
        ctx.mark_non_differentiable(*[])
        return outs

    @staticmethod
    def backward(ctx, g_store, g_wh, g_ech, g_reward):
        prob = ctx.prob
        has_wh, has_ech, has_awh, has_aech = ctx.has
        saved = list(ctx.saved_tensors)
        store = saved.pop(0)
        wh = saved.pop(0) if has_wh else None
        ech = saved.pop(0) if has_ech else None
        a_store = saved.pop(0)
        a_wh = saved.pop(0) if has_awh else None
        a_ech = saved.pop(0) if has_aech else None
        B = prob.B

        def dense(g):
            return None if g is None else g.contiguous()

        if g_reward is None:
            g_reward = torch.zeros(prob.ldb, device=store.device)
        g_reward = g_reward.contiguous()
        g_in, (g_as, g_aw, g_ae) = ops.env_step_bwd(
            prob, EnvState(store, wh, ech), ctx.demand, Table.from_orders(a_store),
            Table.from_orders(a_wh[:, :, 0]) if a_wh is not None else None,
            Table.from_orders(a_ech[:, :, 0]) if a_ech is not None else None,
            EnvState(dense(g_store), dense(g_wh), dense(g_ech)), Table(g_reward, 0, 1))
        return (None, None, g_in.store, g_in.wh, g_in.ech, ref_view(g_as, B),
                ref_view(g_aw, B).unsqueeze(2) if a_wh is not None else None,
                ref_view(g_ae, B).unsqueeze(2) if a_ech is not None else None)


class Simulator:
    """Drop-in for the reference's `Simulator(gym.Env)` (environment.py:7).  gymnasium is not required: the action /
    observation spaces the reference builds (environment.py:347-389) are never consumed by its own code."""

    metadata = {"render_modes": None}

    def __init__(self, device="cpu"):
        self.device = device
        self.problem_params, self.observation_params, self.maximize_profit = None, None, None
        self.batch_size, self.n_stores, self.periods, self.observation, self._internal_data = None, None, None, None, None
        self.action_space = None
        self.observation_space = None
        self._prob = None
        self._t = 0
        # what becomes of a non-zero store order on a (store, supplier) pair whose lead time is 0: "drop" (the kernel's rule,
        # default) or "upstream" (the reference's flat-index put, environment.py:415-432: the order is added to the element in
        # front of the store's pipeline - previous store / previous scenario - reproduced as a differentiable fix-up behind the
        # kernel; single process only, see gnn_rollout.GnnRollout.zero_lead_orders)
        self.zero_lead_orders = "drop"

    # ---- reset (environment.py:24-75, 301-345) ----------------------------------------------------------------
    def reset(self, periods, problem_params, data, observation_params):
        _lib.require_device()
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise _lib.NicUnavailableError(
                "Simulator(device='cpu'): this engine only runs on the GPU (no CPU fallback); pass device='cuda:0'")
        self.problem_params = problem_params
        self.observation_params = observation_params
        if self.zero_lead_orders == "upstream":
            from . import parallel
            if parallel.active() and parallel.world_size() > 1:   # (the rule adds an order to the NEIGHBOURING scenario's state)
                raise ValueError("zero_lead_orders='upstream' couples neighbouring scenarios across shard boundaries: single process only")
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
        self.batch_size, self.n_stores, self.periods = len(data["initial_inventories"]), problem_params["n_stores"], periods
        B = self.batch_size

        self._internal_data = {"demands": data["demands"], "period_shift": observation_params["demand"]["period_shift"]}
        if observation_params["time_features"] is not None:
            self._internal_data.update({k: data[k] for k in observation_params["time_features"]})
        if observation_params["sample_features"] is not None:
            self._internal_data.update({k: data[k] for k in observation_params["sample_features"]})

        prob = self._problem_for(problem_params, data, dev)
        self._prob = prob
        from . import library   # (handle of the problem for `nic::env_step`, the operator `step` emits while a compiler traces)
        library.release_problem(getattr(self, "_problem_handle", None))
        self._problem_handle = library.register_problem(prob)
        self._lead_times = data["lead_times"]
        # demand trace in [T][S][ldb]: the per-period read of the kernel is then one contiguous (S x ldb) panel instead
        # of the reference's stride-T gather (environment.py:177)
        dem = demand_trace_soa(data["demands"], prob.ldb, dev)
        self._demand_soa = dem
        self._state = EnvState(
            to_soa(data["initial_inventories"], prob.ldb),
            to_soa(data["initial_warehouse_inventories"], prob.ldb) if prob.Wn else None,
            to_soa(data["initial_echelon_inventories"], prob.ldb) if prob.E else None)
        self._t = 0

        obs = {"store_inventories": ref_view(self._state.store, B), "current_period": torch.tensor([0])}
        if observation_params["include_warehouse_inventory"]:
            obs["warehouse_lead_times"] = data["warehouse_lead_times"]
            obs["warehouse_holding_costs"] = data["warehouse_holding_costs"]
            obs["warehouse_inventories"] = ref_view(self._state.wh, B)
            if data.get("warehouse_edge_costs") is not None:
                obs["warehouse_edge_costs"] = data["warehouse_edge_costs"]
        if problem_params["n_extra_echelons"] > 0:
            obs["echelon_lead_times"] = data["echelon_lead_times"]
            obs["echelon_holding_costs"] = data["echelon_holding_costs"]
            obs["echelon_inventories"] = ref_view(self._state.ech, B)
        for k, on in observation_params["include_static_features"].items():
            if on:
                obs[k] = data[k]
        if observation_params["demand"]["past_periods"] > 0:
            obs["past_demands"] = self.update_past_demands(data, observation_params, B, self.n_stores, current_period=0)
        if observation_params["time_features"]:
            self.update_time_features(data, obs, observation_params, current_period=0)
        if observation_params["sample_features"] is not None:
            for k in observation_params["sample_features"]:
                obs[k] = data[k]
        self.observation = obs
        self.maximize_profit = problem_params["maximize_profit"]
        return self.observation, None

    def _add_zero_lead_orders(self, store, a_store):
        """Upstream's treatment of non-zero orders with lead time 0 (which the kernel dropped): each is added to the flat element in
        front of its store's pipeline.  Plain torch ops on the scenario-minor state, so autograd carries the order's gradient."""
        lt = self._lead_times
        B, S, Ws = self.batch_size, self.n_stores, store.shape[1]
        if lt.dim() == 2:
            lt = lt.unsqueeze(2)
        hit = (lt.to(a_store.device) == 0) & (a_store.detach() != 0)            # (B, S, suppliers)
        u = (a_store * hit).sum(dim=2)                                           # (B, S): what each store's misplaced orders add up to
        delta = torch.zeros_like(store)                                          # [S][Ws][ld]
        if S > 1:
            delta[:S - 1, Ws - 1, :B] = u[:, 1:].t()                             # store s >= 1 -> last slot of store s - 1
        delta[S - 1, Ws - 1, :B] = delta[S - 1, Ws - 1, :B] + torch.roll(u[:, 0], -1)   # store 0 -> previous scenario's last store
        return store + delta

    def _problem_for(self, problem_params, data, dev):
        """EnvProblem of a batch, cached per presented tensors (layout.ProblemCache: entries pin the tensors they were
        keyed on), so a rollout over a batch seen before contains no host sync at all - a requirement for capturing it
        into a HIP graph."""
        cache = self.__dict__.setdefault("_prob_cache", ProblemCache())
        return cache.get(problem_params, data, dev)

    # ---- step (environment.py:110-169) ------------------------------------------------------------------------
    def step(self, action):
        prob, obs, t = self._prob, self.observation, self._t
        idata = self._internal_data
        n_data_periods = idata["demands"].shape[2]
        if n_data_periods + 2 < t:  # environment.py:492-493
            raise ValueError("Current period is greater than the number of periods in the data")
        tt = t + idata["period_shift"]
        if not 0 <= tt < n_data_periods:
            raise IndexError(f"period {tt} outside the demand trace (0..{n_data_periods - 1})")
        if self.observation_params["demand"]["past_periods"] > 0:  # environment.py:494-501
            obs["past_demands"] = self.update_past_demands(idata, self.observation_params, self.batch_size, self.n_stores,
                                                           current_period=min(t + 1, n_data_periods))
        self.update_time_features(idata, obs, self.observation_params, current_period=t + 1)

        a_store = action["stores"]
        if a_store.dim() != 3 or a_store.shape[1] != prob.S or a_store.shape[2] != prob.nsup:
            raise ValueError(f"action['stores'] must be (B, {prob.S}, {prob.nsup}), got {tuple(a_store.shape)}")
        a_wh = action["warehouses"] if prob.Wn else None
        a_ech = action["echelons"] if prob.E else None
        for name, a in (("warehouses", a_wh), ("echelons", a_ech)):
            if a is not None and (a.dim() != 3 or a.shape[2] != 1):
                raise ValueError(f"action['{name}'] must be (B, n, 1): one outside supplier per location")
        st = self._state
        if torch.compiler.is_compiling():
            # while a compiler traces (torch.compile of a rollout loop), the period is the REGISTERED operator - same kernels,
            # fake-tensor kernel and autograd formula known to the dispatcher (library.py); eager code keeps the Function
            store, wh, ech, reward = torch.ops.nic.env_step(st.store, st.wh, st.ech, a_store, a_wh, a_ech, self._demand_soa[tt],
                                                            self._problem_handle)
            wh, ech = (wh if prob.Wn else None), (ech if prob.E else None)
        else:
            demand = Table(self._demand_soa[tt], prob.ldb, 1)
            store, wh, ech, reward = _EnvStepFunction.apply(prob, demand, st.store, st.wh, st.ech, a_store, a_wh, a_ech)
        if self.zero_lead_orders == "upstream":
            store = self._add_zero_lead_orders(store, a_store)
        self._state = EnvState(store, wh, ech)
        B = self.batch_size
        obs["store_inventories"] = ref_view(store, B)
        if prob.Wn:
            obs["warehouse_inventories"] = ref_view(wh, B)
        if prob.E:
            obs["echelon_inventories"] = ref_view(ech, B)
        self._t = t + 1
        obs["current_period"] += 1  # host tensor, like the reference's (environment.py:165,308)
        terminated = obs["current_period"] >= self.periods
        return obs, reward[:B], terminated, None, None

    # ---- real-data observation features (environment.py:436-468); plain tensor slicing, no kernels involved ----
    def update_past_demands(self, data, observation_params, batch_size, stores, current_period):
        past = observation_params["demand"]["past_periods"]
        cur = current_period + self._internal_data["period_shift"]
        dev = data["demands"].device
        if cur == 0:
            return torch.zeros(batch_size, stores, past, device=dev)
        lo = max(0, cur - past)
        window = data["demands"][:, :, lo:cur]
        missing = past - (cur - lo)
        if missing > 0:
            window = torch.cat([torch.zeros(batch_size, stores, missing, device=dev), window], dim=2)
        return window

    def update_time_features(self, data, observation, observation_params, current_period):
        if observation_params["time_features"] is not None:
            for k in observation_params["time_features"]:
                if data[k].shape[2] + 2 < current_period:
                    raise ValueError("Current period is greater than the number of periods in the data")
                observation[k] = data[k][:, :, min(current_period + observation_params["demand"]["period_shift"],
                                                   data[k].shape[2] - 1)]

    def get_current_demands(self, data, current_period):
        return data["demands"][:, :, current_period + self._internal_data["period_shift"]]
