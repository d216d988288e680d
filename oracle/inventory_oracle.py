"""CPU oracle for the differentiable inventory-rollout path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU (float32, eager, autograd), the algorithm of the
upstream reference's hot path so that the HIP kernels can be checked on the GPU box where
the reference itself cannot travel.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package
(`neural_inventory_control_amd/`) never does and has no CPU fallback.

Pinning: `tests/test_oracle_vs_reference.py` imports the real reference (this container only)
and requires BIT-EQUAL data tensors, per-period rewards, states and parameter gradients for
every BASELINE configuration at small sizes; `tests/golden/make_golden.py` dumps the reference's
outputs into `tests/golden/*.npz`, which `tests/test_oracle_golden.py` re-checks everywhere
(including the shipped-checkpoint known answer 6.854347).

Every function cites the reference file:line it follows (paths relative to /root/reference).
The code is a functional restatement (no classes mirrored, no gym), not a copy.
"""
from __future__ import annotations

import copy
import math
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# 1. Scenario generation  (data_handling.py:9-81, 125-383)
# --------------------------------------------------------------------------------------


def _flag(params: dict, key: str):
    """Missing keys read as False (data_handling.py:248 wraps the dict in a defaultdict)."""
    return params.get(key, False)


def _per_store_table(n_samples, n_stores, spec, seed, integer):
    """Cost / lead-time table for every (sample, store).  data_handling.py:239-271.

    Seeds the GLOBAL numpy legacy RNG (:244) then draws either one value per store (:256-257),
    one value per sample (:258-259), or broadcasts a configured constant / matrix (:260-271).
    """
    np.random.seed(seed)
    draw = np.random.randint if integer else np.random.uniform
    if _flag(spec, "file_location"):  # per-sample values kept on disk (:254-255); later flags still apply to `value`
        spec = dict(spec, value=torch.load(spec["file_location"], map_location="cpu")[:n_samples])
    if _flag(spec, "sample_across_stores"):
        return torch.tensor(draw(*spec["range"], n_stores)).expand(n_samples, -1)
    if _flag(spec, "vary_across_samples"):
        return torch.tensor(draw(*spec["range"], n_samples)).unsqueeze(1).expand(-1, n_stores)
    if _flag(spec, "expand"):
        v = torch.tensor(spec["value"])
        if v.dim() == 2:  # [n_stores, n_warehouses] matrix (:263-266)
            return v.unsqueeze(0).expand(n_samples, -1, -1)
        return v.expand(n_samples, n_stores)
    return torch.tensor(spec["value"])


def _sample_store_demand_moments(n_stores, demand_spec, seeds):
    """Per-store mean ~ U(mean_range).round(3), std = (mean*cv).round(3).  data_handling.py:225-237."""
    np.random.seed(seeds["mean"])
    lo, hi = demand_spec["mean_range"]
    means = np.random.uniform(lo, hi, n_stores).round(3)
    np.random.seed(seeds["coef_of_var"])
    lo, hi = demand_spec["coef_of_var_range"]
    cv = np.random.uniform(lo, hi, n_stores)
    return means, (means * cv).round(3)


def _demand_traces(n_samples, n_stores, periods, demand_spec, seed):
    """(N,S,T) demand array.  data_handling.py:178-211."""
    kind = demand_spec["distribution"]
    if kind == "real":  # :162-168: the first n_samples rows of a (products, stores, weeks) tensor file
        return torch.load(demand_spec["file_location"], map_location="cpu")[:n_samples]
    if seed is not None:
        np.random.seed(seed)
    if kind == "poisson":  # :205-211
        return np.random.poisson(demand_spec["mean"], size=(n_samples, n_stores, periods))
    if kind != "normal":
        raise NotImplementedError(f"oracle: unknown demand distribution {kind}")
    if n_stores == 1:  # :187-191
        return np.random.normal(demand_spec["mean"], demand_spec["std"], size=(n_samples, 1, periods))
    # correlated multivariate normal, cov_ij = rho*s_i*s_j (i != j), s_i^2 on the diagonal (:194-201)
    rho = demand_spec["correlation"]
    stds = demand_spec["std"]
    cov = [[rho * a * b if i != j else a * b for i, a in enumerate(stds)] for j, b in enumerate(stds)]
    draws = np.random.multivariate_normal(demand_spec["mean"], cov=cov, size=(n_samples, periods))
    return np.transpose(draws, (0, 2, 1))


def generate_scenario_data(periods, problem_params, store_params, warehouse_params, echelon_params,
                           num_samples, observation_params, seeds):
    """Restates `Scenario.__init__` + `get_data` (data_handling.py:9-81).

    MUTATES `store_params['demand']` (gains numpy mean/std, :176) and `seeds['demand']`
    (:155-158) exactly like the reference, because downstream code reads the mutated values
    (neural_networks.py:1543).  Returns the float32 batch dict (None entries dropped, :81).
    """
    S = problem_params["n_stores"]
    Wn = problem_params["n_warehouses"]
    demand_spec = store_params["demand"]

    # --- demand (data_handling.py:125-148)
    if _flag(demand_spec, "sample_across_stores"):  # :175-176
        m, s = _sample_store_demand_moments(S, demand_spec, seeds)
        demand_spec.update({"mean": m, "std": s})
    if Wn == 0 and S == 1 and demand_spec["distribution"] != "real":  # :150-160
        try:
            seeds["demand"] = seeds["demand"] + int(store_params["lead_time"]["value"]
                                                    + 10 * store_params["underage_cost"]["value"])
        except Exception as e:  # reference swallows and prints
            print(f"Error: {e}")
    demand = _demand_traces(num_samples, S, periods, demand_spec, seeds["demand"])
    if demand_spec["clip"]:
        demand = np.clip(demand, 0, None)
    demands = torch.as_tensor(demand).clone()

    # --- per-store costs and lead times (data_handling.py:20-22, 273-288)
    underage = _per_store_table(num_samples, S, store_params["underage_cost"], seeds["underage_cost"], False)
    holding = _per_store_table(num_samples, S, store_params["holding_cost"], seeds["holding_cost"], False)
    lead = _per_store_table(num_samples, S, store_params["lead_time"], seeds["lead_time"], True)
    if lead.dim() == 2:
        lead = lead.unsqueeze(2).expand(-1, -1, Wn) if Wn > 0 else lead.unsqueeze(2)
    lead = lead.to(torch.int64)

    # --- demand moments as static features (data_handling.py:373-383)
    feats = observation_params["include_static_features"]
    mean_t = std_t = None
    if feats.get("mean"):
        mean_t = torch.tensor(demand_spec["mean"]).unsqueeze(0).expand(num_samples, -1)
    if feats.get("std"):
        std_t = torch.tensor(demand_spec["std"]).unsqueeze(0).expand(num_samples, -1)

    # --- initial store pipelines (data_handling.py:290-310): global per-store demand mean (:298)
    np.random.seed(seeds["initial_inventory"])
    init_spec = store_params["initial_inventory"]
    if init_spec["sample"]:
        store_mean = demands.float().mean(dim=2).mean(dim=0)
        slots = max(init_spec["inventory_periods"], lead.max().item())
        mults = np.random.uniform(*init_spec["range_mult"], size=(num_samples, S, slots))
        init_inv = store_mean[None, :, None] * mults  # float32 tensor * float64 ndarray -> float64
    else:
        init_inv = torch.zeros(num_samples, S, init_spec["inventory_periods"])

    out = {
        "demands": demands, "underage_costs": underage, "holding_costs": holding, "lead_times": lead,
        "mean": mean_t, "std": std_t, "initial_inventories": init_inv,
    }

    # --- warehouses (data_handling.py:312-329, 343-361)
    if warehouse_params is not None:
        wl = warehouse_params["lead_time"]
        out["initial_warehouse_inventories"] = torch.zeros(num_samples, Wn, max(wl) if isinstance(wl, list) else wl)
        for key, name in (("lead_time", "warehouse_lead_times"), ("holding_cost", "warehouse_holding_costs"),
                          ("edge_cost", "warehouse_edge_costs")):
            if key == "edge_cost" and "edge_cost" not in warehouse_params:
                continue
            v = warehouse_params[key]
            if isinstance(v, list):
                if len(v) != Wn:
                    raise ValueError(f"warehouse_params['{key}'] list length {len(v)} doesn't match n_warehouses {Wn}")
                out[name] = torch.tensor(v).unsqueeze(0).expand(num_samples, -1)
            else:
                out[name] = torch.tensor([v]).expand(num_samples, Wn)

    # --- extra echelons (data_handling.py:331-341, 363-371)
    if echelon_params is not None:
        el = echelon_params["lead_time"]
        out["initial_echelon_inventories"] = torch.zeros(num_samples, len(el), max(el))
        out["echelon_holding_costs"] = torch.tensor(echelon_params["holding_cost"]).unsqueeze(0).expand(num_samples, -1)
        out["echelon_lead_times"] = torch.tensor(el).unsqueeze(0).expand(num_samples, -1)

    # --- time / sample features read from a csv (data_handling.py:36-48)
    for kind_, file_key in (("time_features", "time_features_file"), ("sample_features", "sample_features_file")):
        if observation_params.get(kind_) and observation_params.get(file_key):
            import pandas as pd
            table = pd.read_csv(observation_params[file_key])
            for k in observation_params[kind_]:
                col = torch.tensor(table[k].values)
                out[k] = (col.unsqueeze(0).unsqueeze(0).expand(num_samples, S, -1) if kind_ == "time_features"
                          else col.unsqueeze(1).expand(-1, S))

    return {k: v.float() for k, v in out.items() if v is not None}


def split_data_by_period(data, period_ranges, observation_params, problem_params=None):
    """DatasetCreator.split_by_period (data_handling.py:431-448) with the key lists of define_how_to_split_data (:84-122): per-sample
    tensors are shared, `demands` and the time features are sliced along the period axis, and ONLY the listed keys survive -
    warehouse / echelon entries are listed when the problem has such locations (:94-104), so a setting that configures
    warehouse_params with n_warehouses = 0 loses its zero-width warehouse tensors here.  period_ranges: strings like '(0, 111)'."""
    time_feats = list(observation_params.get("time_features") or [])
    sample_feats = list(observation_params.get("sample_features") or [])
    by_period = ["demands"] + time_feats
    by_sample = ["underage_costs", "holding_costs", "lead_times", "initial_inventories"]
    if problem_params is None or problem_params["n_warehouses"] > 0:
        by_sample += ["initial_warehouse_inventories", "warehouse_lead_times", "warehouse_holding_costs", "warehouse_edge_costs"]
    if problem_params is None or problem_params["n_extra_echelons"] > 0:
        by_sample += ["initial_echelon_inventories", "echelon_holding_costs", "echelon_lead_times"]
    static = observation_params.get("include_static_features") or {}
    by_sample += [k for k in ("mean", "std") if static.get(k)]
    by_sample += sample_feats
    out = []
    for rng in period_ranges:
        sl = slice(*map(int, str(rng).strip("() ").split(",")))
        this = {k: data[k].clone() for k in by_sample if k in data}
        this.update({k: data[k][:, :, sl] for k in by_period if k in data})
        out.append(this)
    return out


# --------------------------------------------------------------------------------------
# 2. One period of inventory dynamics  (environment.py:110-299, 391-434)
# --------------------------------------------------------------------------------------


def shift_pipeline_and_place(pipeline, on_hand_after, orders, lead_times, zero_lead_orders="upstream"):
    """environment.py:391-434.

    new[0] = on_hand_after + old[1]; new[k] = old[k+1]; new[W-1] = 0 (:405-412); then every order
    with value != 0 is added to slot L-1 of its location (:422-432), accumulating in ascending
    flattened (b, location, supplier) order like `Tensor.put(accumulate=True)` on CPU.  Orders
    that are exactly 0 are filtered out BEFORE the put (:426-429) so they carry no gradient on
    this path; if every order is 0 the put is skipped (:427).

    UPSTREAM DEFECT, reproduced on purpose (`zero_lead_orders="upstream"`, what the fixtures pin): a non-zero order whose lead
    time is 0 gets flat position base - 1, i.e. it is added to the element BEFORE its location's pipeline - the last slot of the
    previous location, for location 0 the previous SCENARIO's last location, for scenario 0 the last element of the whole
    batch (`put` takes index -1).  The reference's own policies never order on a (store, warehouse) pair without an edge
    (lead time 0), except the GNN on a many-warehouse graph, whose action columns are "j-th connected warehouse" rather than
    warehouse j (neural_networks.py:1423-1428; `gnn_graph`'s `misplaced`).  `zero_lead_orders="drop"` discards such orders
    instead - what the HIP env step does (an order without a lead time has no slot to arrive in, and scenarios stay
    independent); tests compare the HIP path with this mode on the one fixture where the two differ.
    """
    assert zero_lead_orders in ("upstream", "drop")
    B, N, W = pipeline.shape
    cols = [on_hand_after + pipeline[:, :, 1]]
    cols += [pipeline[:, :, k + 1] for k in range(1, W - 1)]
    cols.append(torch.zeros(orders.shape[0], orders.shape[1], dtype=pipeline.dtype))
    fresh = torch.stack(cols, dim=2)

    n_sup = orders.shape[2]
    base = (torch.arange(B) * (W * N))[:, None] + (torch.arange(N) * W).expand(B, N)  # :77-101
    flat_pos = (base.unsqueeze(2).expand(-1, -1, n_sup) + lead_times.long() - 1).flatten()
    vals = orders.flatten()
    keep = vals != 0
    if zero_lead_orders == "drop":
        keep = keep & (lead_times.long().expand(B, N, n_sup).flatten() >= 1)
    if keep.any():
        fresh = fresh.put(flat_pos[keep], vals[keep].to(fresh.dtype), accumulate=True)
    return fresh


@dataclass
class OracleEnv:
    """State of one rollout: the reference keeps these in `Simulator.observation` / `_internal_data`."""
    problem: dict
    demands: torch.Tensor
    period_shift: int
    periods: int
    obs: Dict[str, torch.Tensor] = field(default_factory=dict)
    t: int = 0
    observation_params: Optional[dict] = None
    data: Optional[dict] = None
    zero_lead_orders: str = "upstream"   # see shift_pipeline_and_place


def env_reset(periods, problem_params, data, observation_params) -> OracleEnv:
    """environment.py:24-75 + 301-345 (observation dict built by reference, no clone)."""
    env = OracleEnv(problem=problem_params, demands=data["demands"],
                    period_shift=observation_params["demand"]["period_shift"], periods=periods)
    obs = {"store_inventories": data["initial_inventories"], "current_period": torch.tensor([0])}
    if observation_params["include_warehouse_inventory"]:
        obs["warehouse_lead_times"] = data["warehouse_lead_times"]
        obs["warehouse_holding_costs"] = data["warehouse_holding_costs"]
        obs["warehouse_inventories"] = data["initial_warehouse_inventories"]
        if data.get("warehouse_edge_costs") is not None:
            obs["warehouse_edge_costs"] = data["warehouse_edge_costs"]
    if problem_params["n_extra_echelons"] > 0:
        obs["echelon_lead_times"] = data["echelon_lead_times"]
        obs["echelon_holding_costs"] = data["echelon_holding_costs"]
        obs["echelon_inventories"] = data["initial_echelon_inventories"]
    for k, on in observation_params["include_static_features"].items():
        if on:
            obs[k] = data[k]
    env.observation_params = observation_params
    env.data = data
    if observation_params["demand"]["past_periods"] > 0:  # :329-330
        obs["past_demands"] = _past_demands(env, current_period=0)
    if observation_params.get("time_features"):            # :333-334
        _update_time_features(env, obs, current_period=0)
    if observation_params.get("sample_features") is not None:
        for k in observation_params["sample_features"]:
            obs[k] = data[k]
    env.obs = obs
    return env


def _past_demands(env, current_period):
    """environment.py:436-458: the last `past_periods` demands before (current_period + period_shift), zero-filled on the left."""
    past = env.observation_params["demand"]["past_periods"]
    cur = current_period + env.period_shift
    d = env.demands
    B, S = d.shape[0], d.shape[1]
    if cur == 0:
        return torch.zeros(B, S, past, dtype=d.dtype)
    window = d[:, :, max(0, cur - past):cur]
    missing = past - (cur - max(0, cur - past))
    if missing > 0:
        window = torch.cat([torch.zeros(B, S, missing, dtype=d.dtype), window], dim=2)
    return window


def _update_time_features(env, obs, current_period):
    """environment.py:460-468."""
    for k in env.observation_params["time_features"]:
        feat = env.data[k]
        if feat.shape[2] + 2 < current_period:
            raise ValueError("Current period is greater than the number of periods in the data")
        obs[k] = feat[:, :, min(current_period + env.period_shift, feat.shape[2] - 1)]


def env_step(env: OracleEnv, action: Dict[str, torch.Tensor]) -> torch.Tensor:
    """One period; returns the per-scenario cost (B,) and rebinds the state.  environment.py:110-169."""
    obs, prob = env.obs, env.problem
    d = env.demands[:, :, env.t + env.period_shift]  # :171-177
    op = getattr(env, "observation_params", None)
    if op is not None:
        # observation features of the NEXT period are refreshed before the dynamics (:126-136, :486-501)
        if env.demands.shape[2] + 2 < env.t:
            raise ValueError("Current period is greater than the number of periods in the data")
        if op["demand"]["past_periods"] > 0:
            obs["past_demands"] = _past_demands(env, current_period=min(env.t + 1, env.demands.shape[2]))
        if op.get("time_features"):
            _update_time_features(env, obs, current_period=env.t + 1)

    # --- stores (:179-234)
    inv = obs["store_inventories"]
    on_hand = inv[:, :, 0]
    after = inv[:, :, 0] - d
    p, h = obs["underage_costs"], obs["holding_costs"]
    if prob["maximize_profit"]:
        cost = -p * torch.minimum(on_hand, d) + h * torch.clip(after, min=0)  # :191-194
    else:
        cost = p * torch.clip(-after, min=0) + h * torch.clip(after, min=0)  # :198-201
    if prob["lost_demand"]:
        after = torch.clip(after, min=0)  # :204-205
    obs["store_inventories"] = shift_pipeline_and_place(inv, after, action["stores"], obs["lead_times"], env.zero_lead_orders)
    total = cost.sum(dim=1)

    # --- warehouses (:236-270)
    if prob["n_warehouses"] > 0:
        winv = obs["warehouse_inventories"]
        shipped = action["stores"].sum(dim=1)
        w_after = winv[:, :, 0] - shipped  # may go negative: feasibility is the policy's job
        w_cost = obs["warehouse_holding_costs"] * torch.clip(w_after, min=0)
        if obs.get("warehouse_edge_costs") is not None:
            w_cost = w_cost + obs["warehouse_edge_costs"] * action["warehouses"].sum(dim=2)  # :254-259
        obs["warehouse_inventories"] = shift_pipeline_and_place(
            winv, w_after, action["warehouses"], obs["warehouse_lead_times"].unsqueeze(2))
        total = total + w_cost.sum(dim=1)

    # --- extra echelons (:272-299): echelon k ships what echelon k+1 ordered; the last one feeds the warehouses
    if prob["n_extra_echelons"] > 0:
        einv = obs["echelon_inventories"]
        downstream = action["echelons"][:, 1:, :].sum(dim=2)
        to_wh = action["warehouses"].sum(dim=(1, 2)).unsqueeze(1)
        e_after = einv[:, :, 0] - torch.concat([downstream, to_wh], dim=1)
        e_cost = obs["echelon_holding_costs"] * torch.clip(e_after, min=0)
        obs["echelon_inventories"] = shift_pipeline_and_place(
            einv, e_after, action["echelons"], obs["echelon_lead_times"].unsqueeze(2))
        total = total + e_cost.sum(dim=1)

    obs["current_period"] = obs["current_period"] + 1  # reference mutates in place (:165)
    env.t += 1
    return total


# --------------------------------------------------------------------------------------
# 3. Policies  (neural_networks.py:6-427, 1495-1574)
# --------------------------------------------------------------------------------------

_ACT = {
    "relu": F.relu, "elu": F.elu, "tanh": torch.tanh, "softplus": F.softplus, "sigmoid": torch.sigmoid,
    "softmax": lambda x: F.softmax(x, dim=1),
}


@dataclass
class OraclePolicy:
    """A policy = architecture name + ordered (weight, bias) pairs of its 'master' MLP."""
    name: str
    layers: List[Tuple[torch.Tensor, torch.Tensor]]
    inner_activation: Optional[str]
    output_activation: Optional[str]
    warehouse_upper_bound: Optional[torch.Tensor] = None
    adjacency: Optional[Sequence[Sequence[int]]] = None  # problem_params['warehouse_store_adjacency']
    transshipment: bool = False
    forecaster: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None  # frozen quantile forecaster (quantile policies)
    forecaster_lead_times: Optional[Sequence[int]] = None

    def parameters(self):
        return [t for wb in self.layers for t in wb]


def forecaster_layers(forecaster_state, dtype=torch.float32):
    """(weight, bias) pairs of the frozen FullyConnectedForecaster from its state dict (`net.<i>.weight/bias`)."""
    idx = sorted({int(k.split(".")[1]) for k in forecaster_state})
    return [(forecaster_state[f"net.{i}.weight"].detach().to(dtype), forecaster_state[f"net.{i}.bias"].detach().to(dtype))
            for i in idx]


def policy_from_state_dict(nn_params, state_dict, problem_params=None, warehouse_upper_bound=None, dtype=torch.float32,
                           forecaster_state=None):
    """Builds an OraclePolicy from reference-format keys `net.master.<i>.weight/bias`
    (layout produced by neural_networks.py:80-106).  dtype=torch.float64 (with float64 data) gives the fp64 REFEREE the
    gradient-parity tests measure both the reference's float32 arithmetic and the HIP engine against."""
    if nn_params["name"] == "gnn":
        return gnn_from_state_dict(nn_params, state_dict, problem_params)
    if nn_params["name"] == "symmetry_aware":
        return symmetry_aware_from_state_dict(nn_params, state_dict, warehouse_upper_bound, dtype)
    idx = sorted({int(k.split(".")[2]) for k in state_dict if k.startswith("net.master.")})
    layers = [(state_dict[f"net.master.{i}.weight"].detach().clone().to(dtype).requires_grad_(True),
               state_dict[f"net.master.{i}.bias"].detach().clone().to(dtype).requires_grad_(True)) for i in idx]
    if warehouse_upper_bound is not None:
        warehouse_upper_bound = warehouse_upper_bound.to(dtype)
    return OraclePolicy(
        name=nn_params["name"], layers=layers,
        inner_activation=nn_params["inner_layer_activations"]["master"],
        output_activation=nn_params["output_layer_activation"]["master"],
        warehouse_upper_bound=warehouse_upper_bound,
        adjacency=(problem_params or {}).get("warehouse_store_adjacency"),
        transshipment=nn_params.get("transshipment", False),
        forecaster=forecaster_layers(forecaster_state, dtype) if forecaster_state is not None else None,
        forecaster_lead_times=nn_params.get("forecaster_lead_times"),
    )


def default_master_output_size(problem_params):
    """neural_networks.py:1500-1517."""
    S, Wn = problem_params["n_stores"], problem_params["n_warehouses"]
    return S * Wn + Wn if Wn > 1 else S + Wn


def warehouse_upper_bound(mult, store_params):
    """neural_networks.py:1538-1546 (reads the MUTATED store_params['demand']['mean'])."""
    mean = store_params["demand"]["mean"]
    if type(mean) == float:
        mean = [mean]
    return torch.tensor([mult * sum(mean)]).float()


def init_policy(nn_params, problem_params, in_features, generator_seed, store_params=None):
    """Random-init policy with torch's default Linear init (kaiming-uniform(a=sqrt(5)), as LazyLinear/Linear do,
    neural_networks.py:88-97).  Used for synthetic benchmarks; parity tests load reference weights instead."""
    g = torch.Generator().manual_seed(generator_seed)
    hidden = list(nn_params["neurons_per_hidden_layer"]["master"])
    out = nn_params["output_sizes"]["master"]
    if out is None:
        out = default_master_output_size(problem_params)
    dims = [in_features] + hidden + [out]
    layers = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = 1.0 / math.sqrt(fan_in)
        w = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound
        b = (torch.rand(fan_out, generator=g) * 2 - 1) * bound
        layers.append((w.requires_grad_(True), b.requires_grad_(True)))
    bias0 = (nn_params.get("initial_bias") or {}).get("master")
    if bias0 is not None:  # neural_networks.py:53-58
        with torch.no_grad():
            layers[-1][1].fill_(bias0)
    ub = None
    if "warehouse_upper_bound_mult" in nn_params and store_params is not None:
        ub = warehouse_upper_bound(nn_params["warehouse_upper_bound_mult"], store_params)
    return OraclePolicy(nn_params["name"], layers, nn_params["inner_layer_activations"]["master"],
                        nn_params["output_layer_activation"]["master"], ub,
                        problem_params.get("warehouse_store_adjacency"), nn_params.get("transshipment", False))


def _zero_input(pol: OraclePolicy):
    """The constant scalar 0 the closed-form policies feed their one-layer 'net' (neural_networks.py:228)."""
    return torch.zeros(1, dtype=pol.layers[0][0].dtype)


def _mlp(pol: OraclePolicy, x):
    """Sequential(Linear, act, ..., Linear[, out_act]).  neural_networks.py:80-106."""
    n = len(pol.layers)
    for i, (w, b) in enumerate(pol.layers):
        x = F.linear(x, w, b)
        if i < n - 1:
            x = _ACT[pol.inner_activation](x)
    if pol.output_activation is not None:
        x = _ACT[pol.output_activation](x)
    return x


def _softmax_share_of_stock(logits, warehouse_pipeline, transshipment):
    """neural_networks.py:140-166: softmax over connected stores (+ a constant-1 'keep' column unless
    transshipment) times the warehouse's on-hand stock."""
    stock = warehouse_pipeline[:, :, 0].sum(dim=1)
    z = logits
    if not transshipment:
        z = torch.cat((z, torch.ones_like(z[:, 0])[:, None]), dim=1)
    sm = F.softmax(z, dim=1)
    if not transshipment:
        sm = sm[:, :-1]
    return torch.multiply(sm, stock[:, None])


# --------------------------------------------------------------------------------------
# Symmetry-aware policy (BASELINE cfg3's wording).  NOT in the reference's source: this restates SURVEY §2.2's recovery of the
# stale bytecode with the helpers that are still upstream (apply_proportional_allocation neural_networks.py:111-138,
# concatenate_signal_to_object_state_tensor :178-187).  PARITY UNPINNED: it pins the HIP path to this file only.
# --------------------------------------------------------------------------------------
SYM_MODULES = ("context", "store", "warehouse")


@dataclass
class OracleSymmetryAwarePolicy:
    name: str
    modules: Dict[str, List[Tuple[torch.Tensor, torch.Tensor]]]
    inner_activation: Dict[str, Optional[str]]
    output_activation: Dict[str, Optional[str]]
    warehouse_upper_bound: torch.Tensor
    keys: Optional[List[str]] = None

    def parameters(self):
        return [t for m in SYM_MODULES for wb in self.modules[m] for t in wb]

    def param_keys(self):
        return list(self.keys)


def symmetry_aware_from_state_dict(nn_params, state_dict, warehouse_upper_bound, dtype=torch.float32):
    modules, keys = {}, []
    for m in SYM_MODULES:
        idx = sorted({int(k.split(".")[2]) for k in state_dict if k.startswith(f"net.{m}.")})
        modules[m] = []
        for i in idx:
            w = state_dict[f"net.{m}.{i}.weight"].detach().clone().to(dtype).requires_grad_(True)
            b = state_dict[f"net.{m}.{i}.bias"].detach().clone().to(dtype).requires_grad_(True)
            modules[m].append((w, b))
            keys += [f"net.{m}.{i}.weight", f"net.{m}.{i}.bias"]
    return OracleSymmetryAwarePolicy(name="symmetry_aware", modules=modules,
                                     inner_activation=dict(nn_params["inner_layer_activations"]),
                                     output_activation=dict(nn_params["output_layer_activation"]),
                                     warehouse_upper_bound=warehouse_upper_bound.detach().clone().to(dtype), keys=keys)


def symmetry_aware_act(pol, obs):
    s_inv, w_inv = obs["store_inventories"], obs["warehouse_inventories"]
    B, S = s_inv.shape[0], s_inv.shape[1]
    context = _gnn_mlp(pol, "context", torch.cat((s_inv.flatten(start_dim=1), w_inv.flatten(start_dim=1)), dim=1))
    w_in = torch.cat((w_inv, context.unsqueeze(1).expand(-1, w_inv.size(1), -1)), dim=2)          # :178-187
    warehouse_out = _gnn_mlp(pol, "warehouse", w_in)[:, :, 0]
    store_params = torch.stack([obs["mean"], obs["std"], obs["underage_costs"], obs["lead_times"][:, :, 0]], dim=2)
    s_in = torch.cat((s_inv, store_params, context.unsqueeze(1).expand(-1, S, -1)), dim=2)
    store_out = _gnn_mlp(pol, "store", s_in)[:, :, 0]
    available = w_inv[:, :, 0].sum(dim=1)                                                             # :124-126
    scaling = torch.clip(available / (store_out.sum(dim=1) + 1e-10), max=1.0)                       # :129-135
    stores = store_out * scaling[:, None]
    return {"stores": stores.unsqueeze(2), "warehouses": (warehouse_out * pol.warehouse_upper_bound.reshape(1, -1)).unsqueeze(2)}


def policy_act(pol: OraclePolicy, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Forward of the in-scope architectures; action tensors are always 3-D."""
    name = pol.name
    if name == "gnn":
        return gnn_act(pol, obs)
    if name == "symmetry_aware":
        return symmetry_aware_act(pol, obs)
    if name == "vanilla_one_store":  # neural_networks.py:200-214
        x = obs["store_inventories"].flatten(start_dim=1)
        x = F.softplus(_mlp(pol, x) + 1)
        return {"stores": x.unsqueeze(2)}

    if name == "base_stock":  # :221-229
        pos = obs["store_inventories"].sum(dim=2)
        level = _mlp(pol, _zero_input(pol))
        return {"stores": torch.clip(level - pos, min=0).unsqueeze(2)}

    if name == "capped_base_stock":  # :301-311
        pos = obs["store_inventories"].sum(dim=2)
        out = _mlp(pol, _zero_input(pol))
        return {"stores": torch.clip(out[0] - pos, min=_zero_input(pol), max=out[1]).unsqueeze(2)}

    if name == "echelon_stock":  # :236-294
        s_inv, w_inv, e_inv = obs["store_inventories"], obs["warehouse_inventories"], obs["echelon_inventories"]
        E = e_inv.size(1)
        x = F.softplus(_mlp(pol, _zero_input(pol)) + 10.0)
        levels = torch.cumsum(x, dim=0).flip(dims=[0])
        pos = torch.concat((e_inv.sum(dim=2), w_inv.sum(dim=2), s_inv.sum(dim=2)), dim=1)
        upstream = torch.concat((1000000 * torch.ones_like(w_inv[:, :, 0]), e_inv[:, :, 0], w_inv[:, :, 0]), dim=1)
        want = torch.clip(torch.stack([levels[k] - pos[:, k:].sum(dim=1) for k in range(2 + E)], dim=1), min=0)
        alloc = torch.minimum(want, upstream)
        return {"stores": alloc[:, -1:].unsqueeze(2), "warehouses": alloc[:, -2:-1].unsqueeze(2),
                "echelons": alloc[:, :E].unsqueeze(2)}

    if name == "vanilla_serial":  # :319-355
        s_inv, w_inv, e_inv = obs["store_inventories"], obs["warehouse_inventories"], obs["echelon_inventories"]
        E = e_inv.size(1)
        feats = torch.cat([t.flatten(start_dim=1) for t in (s_inv, w_inv, e_inv)], dim=1)
        # :329 wraps the input in torch.tensor(...), which copy-constructs and DETACHES it: no gradient reaches
        # the state through the MLP input in this architecture (only through `upstream` below).  Replicated.
        x = _mlp(pol, feats.detach().clone())
        upstream = torch.concat((pol.warehouse_upper_bound.unsqueeze(1).expand(e_inv.shape[0], -1),
                                 e_inv[:, :, 0], w_inv[:, :, 0]), dim=1)
        alloc = torch.sigmoid(x) * upstream
        return {"stores": alloc[:, -1:].unsqueeze(2), "warehouses": alloc[:, -2:-1].unsqueeze(2),
                "echelons": alloc[:, :E].unsqueeze(2)}

    if name == "vanilla_warehouse":  # :369-427
        s_inv, w_inv = obs["store_inventories"], obs["warehouse_inventories"]
        S, Wn = s_inv.size(1), w_inv.size(1)
        out = _mlp(pol, torch.cat((s_inv.flatten(start_dim=1), w_inv.flatten(start_dim=1)), dim=1))
        if Wn == 1:
            adj = torch.ones(1, S)
        else:
            if pol.adjacency is None:
                raise ValueError(f"warehouse_store_adjacency matrix required for n_warehouses={Wn}")
            adj = torch.tensor(pol.adjacency, dtype=torch.float32)
        store_logits = out[:, :S * Wn].view(-1, S, Wn)
        wh_logits = out[:, S * Wn:]
        alloc = torch.zeros_like(store_logits)
        for w in range(Wn):
            conn = adj[w].nonzero(as_tuple=True)[0]
            if len(conn) > 0:
                alloc[:, conn, w] = _softmax_share_of_stock(store_logits[:, conn, w], w_inv[:, w:w + 1], pol.transshipment)
        wh = torch.sigmoid(wh_logits) * pol.warehouse_upper_bound
        return {"stores": alloc, "warehouses": wh.unsqueeze(2)}

    if name == "data_driven":  # neural_networks.py:438-512
        s_inv = obs["store_inventories"]
        Wn = obs["warehouse_inventories"].size(1) if "warehouse_inventories" in obs else 0
        S = s_inv.size(1)
        feats = [s_inv] + ([obs["warehouse_inventories"]] if Wn > 0 else [])
        feats += [obs["past_demands"], obs["underage_costs"], obs["holding_costs"], obs["days_from_christmas"], obs["lead_times"]]
        out = _mlp(pol, torch.cat([t.flatten(start_dim=1) for t in feats], dim=1))
        if Wn == 0:
            return {"stores": out.unsqueeze(2)}
        edge_mask = torch.tensor(pol.adjacency, dtype=out.dtype).transpose(0, 1)  # [S, Wn]
        wh_out, store_flat = out[:, :Wn], out[:, Wn:]
        alloc = store_flat.reshape(out.size(0), S, Wn) * edge_mask.unsqueeze(0)
        final = torch.zeros_like(alloc)
        for w in range(Wn):
            if edge_mask[:, w].sum() > 0:  # proportional allocation against the warehouse's whole pipeline (:111-138, :497-500)
                desired = alloc[:, :, w]
                avail = obs["warehouse_inventories"][:, w].sum(dim=1)
                scaling = torch.clip(avail / (desired.sum(dim=1) + 1e-10), max=1.0)
                final[:, :, w] = desired * scaling[:, None]
        return {"stores": final, "warehouses": wh_out.unsqueeze(2)}

    if name in ("transformed_nv", "quantile_nv", "returns_nv", "fixed_quantile"):  # :517-631
        lead = obs["lead_times"][:, :, 0]
        p, h = obs["underage_costs"], obs["holding_costs"]
        if name == "transformed_nv":
            q = _mlp(pol, p / (p + h))
        elif name == "fixed_quantile":
            q = _mlp(pol, _zero_input(pol)).unsqueeze(1).expand(p.shape[0], p.shape[1])
        else:
            q = p / (p + h)
        past, xmas = obs["past_demands"], obs["days_from_christmas"]
        x = torch.cat([past, xmas.unsqueeze(1).expand(past.shape[0], past.shape[1], 1)], dim=2)
        levels = forecaster_quantile(pol, x, q, lead)
        pos = obs["store_inventories"].sum(dim=2)
        if isinstance(obs.get("_probe"), list):  # tests: (level, level - position) before the clip, to find knife edges
            obs["_probe"].append((levels.detach(), (levels - pos).detach()))
        alloc = levels - pos if name == "returns_nv" else torch.clip(levels - pos, min=0)
        return {"stores": alloc.unsqueeze(2)}

    if name == "just_in_time":  # :634-739 (non-admissible: reads future demand from internal_data)
        return _just_in_time(pol, obs)

    raise KeyError(name)


FORECASTER_QS = np.arange(0.05, 1, 0.05)


def forecaster_quantile(pol, x, quantile, lead_times):
    """FullyConnectedForecaster.get_quantile (quantile_forecaster.py:62-123): the frozen net predicts, for each of 19
    probability points and each lead time, cumulative demand; the requested quantile is a linear interpolation between the two
    neighbouring points, with points 0 and 1 extrapolated linearly."""
    # the probability points are a float64 tensor upstream (built from a numpy array, quantile_forecaster.py:33), so the
    # interpolation below - and with it the orders and, from the first step on, the state - is carried in float64
    prob_points = torch.tensor([0] + list(FORECASTER_QS.round(2)) + [1])
    idx = torch.searchsorted(prob_points, quantile.detach())
    h = x
    for i, (w, b) in enumerate(pol.forecaster):
        h = F.linear(h, w, b)
        if i < len(pol.forecaster) - 1:
            h = F.elu(h)
    n_lt = len(pol.forecaster_lead_times)
    h = torch.clip(h, min=0).reshape(*h.shape[:-1], len(FORECASTER_QS), n_lt)
    dif = (lead_times - min(pol.forecaster_lead_times)).to(torch.int64)
    h = torch.gather(h, 3, dif.unsqueeze(2).expand(-1, -1, h.shape[2]).unsqueeze(3)).squeeze(3)
    h = torch.cat([(2 * h[:, :, 0] - h[:, :, 1]).unsqueeze(2), h, (2 * h[:, :, -1] - h[:, :, -2]).unsqueeze(2)], dim=2)
    prev_q = torch.gather(h, 2, (idx - 1).unsqueeze(2)).squeeze(2)
    next_q = torch.gather(h, 2, idx.unsqueeze(2)).squeeze(2)
    d_prev = quantile - prob_points[idx - 1]
    d_next = prob_points[idx] - quantile
    return prev_q + (next_q - prev_q) * d_prev / (d_prev + d_next)


def _just_in_time(pol, obs):
    cur = obs["current_period"]
    demands, shift = obs["internal_data"]["demands"], obs["internal_data"]["period_shift"]
    N, S, L = demands.shape
    Wn = obs["warehouse_inventories"].size(1) if "warehouse_inventories" in obs else 0
    rows = torch.arange(N)
    if Wn == 0:
        lt = obs["lead_times"][:, :, 0]
        fut = torch.stack([demands[:, j][rows, torch.clip(cur + shift + lt[:, j].long(), max=L - 1)] for j in range(S)], dim=1)
        return {"stores": torch.clip(fut, min=0).unsqueeze(2)}
    lt, wlt = obs["lead_times"], obs["warehouse_lead_times"]
    adj = torch.tensor(pol.adjacency, dtype=torch.float32)
    alloc = torch.zeros(N, S, Wn, dtype=demands.dtype)
    wh = torch.zeros(N, Wn, dtype=demands.dtype)
    for st in range(S):
        conn = adj[:, st].nonzero(as_tuple=True)[0]
        if len(conn) > 0:
            w = conn[torch.argmin(lt[:, st, conn].mean(dim=0))].item()  # the connected warehouse with the shortest lead time
            t_store = torch.clip(cur + lt[:, st, w].long() + shift, max=L - 1)
            alloc[:, st, w] = demands[:, st][rows, t_store]
            t_wh = torch.clip(cur + wlt[:, w].long() + lt[:, st, w].long() + shift, max=L - 1)
            wh[:, w] += demands[:, st][rows, t_wh]
    return {"stores": torch.clip(alloc, min=0), "warehouses": torch.clip(wh, min=0).unsqueeze(2)}


# --------------------------------------------------------------------------------------
# 4. Rollout, cost reduction, training step  (trainer.py:143-216, loss_functions.py:11-12)
# --------------------------------------------------------------------------------------


@dataclass
class RolloutResult:
    total: torch.Tensor            # sum_t sum_b reward (0-d, carries grad)      trainer.py:208
    reported: torch.Tensor         # same from t >= ignore_periods               trainer.py:209-210
    per_period: torch.Tensor       # (T, B) detached rewards
    final_obs: Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# GNN policy.  neural_networks.py:742-1492: message passing over the supply graph.
# Restated with the reference's node / edge ORDER and the same sequence of additions, so that results are
# bit-equal: nodes = [echelons..., warehouses..., stores...]; edges = [internal (adjacency.nonzero() order =
# source-major), supplier edges, demand edges, self-loops of nodes that supply others (absent under transshipment)].
# --------------------------------------------------------------------------------------

GNN_MODULES = ("initial_node", "initial_edge", "node_update", "edge_update", "output")


@dataclass
class OracleGNNPolicy:
    name: str
    modules: Dict[str, List[Tuple[torch.Tensor, torch.Tensor]]]
    inner_activation: Dict[str, Optional[str]]
    output_activation: Dict[str, Optional[str]]
    problem_params: Dict
    transshipment: bool = False
    keys: Optional[List[str]] = None  # state-dict keys in parameters() order
    warehouse_upper_bound: Optional[torch.Tensor] = None

    def parameters(self):
        return [t for m in GNN_MODULES for wb in self.modules[m] for t in wb]

    def param_keys(self):
        return list(self.keys)


def gnn_from_state_dict(nn_params, state_dict, problem_params):
    modules, keys = {}, []
    for m in GNN_MODULES:
        idx = sorted({int(k.split(".")[2]) for k in state_dict if k.startswith(f"net.{m}.")})
        modules[m] = []
        for i in idx:
            w = state_dict[f"net.{m}.{i}.weight"].detach().clone().float().requires_grad_(True)
            b = state_dict[f"net.{m}.{i}.bias"].detach().clone().float().requires_grad_(True)
            modules[m].append((w, b))
            keys += [f"net.{m}.{i}.weight", f"net.{m}.{i}.bias"]
    return OracleGNNPolicy(name="gnn", modules=modules, inner_activation=dict(nn_params["inner_layer_activations"]),
                           output_activation=dict(nn_params["output_layer_activation"]),
                           problem_params=dict(problem_params), transshipment=nn_params.get("transshipment", False),
                           keys=keys)


def _gnn_mlp(pol, key, x):
    layers = pol.modules[key]
    for i, (w, b) in enumerate(layers):
        x = F.linear(x, w, b)
        if i < len(layers) - 1:
            x = _ACT[pol.inner_activation[key]](x)
    if pol.output_activation[key] is not None:
        x = _ACT[pol.output_activation[key]](x)
    return x


def gnn_graph(problem_params, obs, transshipment):
    """Static structure of the supply graph (:757-944, :945-1062): edge lists, lead times, output mapping."""
    S, Wn, E = problem_params["n_stores"], problem_params["n_warehouses"], problem_params["n_extra_echelons"]
    n_nodes = E + Wn + S
    internal, lead = [], []
    if E > 0:  # serial: echelon 0 -> ... -> echelon E-1 -> warehouse -> store (one warehouse, one store)
        for i in range(E - 1):
            internal.append((i, i + 1))
            lead.append(obs["echelon_lead_times"][0, i + 1])
        internal.append((E - 1, E))
        lead.append(obs["warehouse_lead_times"][0, 0])
        internal.append((E, E + Wn))
        lead.append(obs["lead_times"][0, 0, 0])
        suppliers, demand_nodes = [0], [E + Wn]
        supplier_lead = [obs["echelon_lead_times"][0, 0]]
    else:
        if Wn == 1:
            conn = [[1] * S]
        else:
            conn = problem_params["warehouse_store_adjacency"]
        for w in range(Wn):
            for st in range(S):
                if conn[w][st]:
                    internal.append((w, Wn + st))
                    lead.append(obs["lead_times"][0][st, w])  # sample 0's lead times stand for the batch (:984)
        suppliers, demand_nodes = list(range(Wn)), list(range(Wn, Wn + S))
        supplier_lead = [obs["warehouse_lead_times"][0, w] for w in range(Wn)]
    out_deg = [0] * n_nodes
    in_deg = [0] * n_nodes
    for (a, b) in internal:
        out_deg[a] += 1
        in_deg[b] += 1
    supplying = [] if transshipment else [n for n in range(n_nodes) if out_deg[n] > 0]
    n_int, n_sup, n_dem = len(internal), len(suppliers), len(demand_nodes)
    # output mapping (:1010-1062)
    if E > 0:
        mapping = {"stores": [[n_int - 1]], "warehouses": [[n_int - 2]],
                   "echelons": [[n_int]] + [[i - 1] for i in range(1, E)]}
    else:
        stores = [[] for _ in range(S)]
        misplaced = []  # (store, action column, warehouse the edge comes from) wherever the two differ - see below
        for i, (a, b) in enumerate(internal):
            # UPSTREAM DEFECT, reproduced: the j-th CONNECTED edge of a store becomes its action column j (:1423-1428), but the
            # env step reads column j as "ordered from warehouse j" (lead time, shipment source).  For a store that is not
            # connected to every warehouse the order is therefore booked on the wrong warehouse - with lead time 0 when that
            # pair has no edge, which shift_pipeline_and_place then misplaces as well.
            if len(stores[b - Wn]) != a:
                misplaced.append((b - Wn, len(stores[b - Wn]), a))
            stores[b - Wn].append(i)
        mapping = {"stores": stores, "warehouses": [[n_int + w] for w in range(Wn)]}
    # degrees used for the normalisation (:1275-1296)
    for n in suppliers:
        in_deg[n] += 1
    for n in demand_nodes:
        out_deg[n] += 1
    for n in supplying:
        in_deg[n] += 1
        out_deg[n] += 1
    return dict(n_nodes=n_nodes, internal=internal, lead=lead, suppliers=suppliers, supplier_lead=supplier_lead,
                demand_nodes=demand_nodes, supplying=supplying, mapping=mapping, misplaced=misplaced if E == 0 else [],
                in_deg=[d if d > 0 else 1 for d in in_deg], out_deg=[d if d > 0 else 1 for d in out_deg],
                num_message_passing=(E + 1) if E > 0 else 1)


def _gnn_node_features(problem_params, obs):
    """[inventory slots padded to the longest pipeline | static features padded to the longest list] per node (:846-905)."""
    E = problem_params["n_extra_echelons"]
    feats, inv_lens = [], []
    if E > 0:
        feats.append(torch.cat([obs["echelon_inventories"], obs["echelon_holding_costs"].unsqueeze(-1)], dim=-1))
        inv_lens.append(obs["echelon_inventories"].size(-1))
    wl = [obs["warehouse_inventories"], obs["warehouse_holding_costs"].unsqueeze(-1)]
    if obs.get("warehouse_edge_costs") is not None:
        wl.append(obs["warehouse_edge_costs"].unsqueeze(-1))
    feats.append(torch.cat(wl, dim=-1))
    inv_lens.append(obs["warehouse_inventories"].size(-1))
    sl = [obs["store_inventories"], obs["holding_costs"].unsqueeze(-1), obs["underage_costs"].unsqueeze(-1),
          obs["mean"].unsqueeze(-1), obs["std"].unsqueeze(-1)]  # KeyError without mean/std, like the reference (:888)
    feats.append(torch.cat(sl, dim=-1))
    inv_lens.append(obs["store_inventories"].size(-1))
    max_inv = max(inv_lens)
    max_st = max(f.size(-1) - n for f, n in zip(feats, inv_lens))
    padded = []
    for f, n in zip(feats, inv_lens):
        inv, st = f[:, :, :n], f[:, :, n:]
        padded.append(torch.cat([F.pad(inv, (0, max_inv - n)), F.pad(st, (0, max_st - (f.size(2) - n)))], dim=2))
    return torch.cat(padded, dim=1)


def _gnn_node_inventories(problem_params, obs):
    parts = []
    if problem_params["n_extra_echelons"] > 0:
        parts.append(obs["echelon_inventories"][:, :, 0])
    if problem_params["n_warehouses"] > 0:
        parts.append(obs["warehouse_inventories"][:, :, 0])
    parts.append(obs["store_inventories"][:, :, 0])
    return torch.cat(parts, dim=1)


def _gnn_edge_endpoints(g, nodes):
    """(source features, target features) of every edge in edge order; virtual supplier / customer ends are zeros."""
    B, D = nodes.size(0), nodes.size(-1)
    src_i = [a for a, _ in g["internal"]]
    tgt_i = [b for _, b in g["internal"]]
    srcs = [nodes[:, src_i], torch.zeros(B, len(g["suppliers"]), D), nodes[:, g["demand_nodes"]]]
    tgts = [nodes[:, tgt_i], nodes[:, g["suppliers"]], torch.zeros(B, len(g["demand_nodes"]), D)]
    if g["supplying"]:
        srcs.append(nodes[:, g["supplying"]])
        tgts.append(nodes[:, g["supplying"]])
    return torch.cat(srcs, dim=1), torch.cat(tgts, dim=1)


def gnn_act(pol: OracleGNNPolicy, obs):
    """GNN.forward (:1367-1392)."""
    prob = pol.problem_params
    g = gnn_graph(prob, obs, pol.transshipment)
    B = obs["store_inventories"].size(0)
    nodes = _gnn_mlp(pol, "initial_node", _gnn_node_features(prob, obs))
    # initial edge features: [source node | target node | lead time] (:1105-1192)
    src, tgt = _gnn_edge_endpoints(g, nodes)
    lt = torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in g["lead"] + g["supplier_lead"]]
                     + [torch.tensor(0.0)] * (len(g["demand_nodes"]) + len(g["supplying"])))
    edges = _gnn_mlp(pol, "initial_edge", torch.cat([src, tgt, lt.view(1, -1, 1).expand(B, -1, -1)], dim=-1))
    n_int, n_sup, n_dem = len(g["internal"]), len(g["suppliers"]), len(g["demand_nodes"])
    in_norm = torch.sqrt(torch.tensor(g["in_deg"], dtype=torch.float32)).view(1, -1, 1)
    out_norm = torch.sqrt(torch.tensor(g["out_deg"], dtype=torch.float32)).view(1, -1, 1)
    for _ in range(g["num_message_passing"]):
        incoming = torch.zeros(B, g["n_nodes"], edges.size(-1))
        outgoing = torch.zeros(B, g["n_nodes"], edges.size(-1))
        for i, (a, b) in enumerate(g["internal"]):  # same order of += as :1230-1235
            incoming[:, b] += edges[:, i]
            outgoing[:, a] += edges[:, i]
        for i, n in enumerate(g["suppliers"]):
            incoming[:, n] += edges[:, n_int + i]
        for i, n in enumerate(g["demand_nodes"]):
            outgoing[:, n] += edges[:, n_int + n_sup + i]
        for i, n in enumerate(g["supplying"]):
            incoming[:, n] += edges[:, n_int + n_sup + n_dem + i]
            outgoing[:, n] += edges[:, n_int + n_sup + n_dem + i]
        incoming = incoming / in_norm
        outgoing = outgoing / out_norm
        nodes = nodes + _gnn_mlp(pol, "node_update", torch.cat([nodes, incoming, outgoing], dim=-1))
        src, tgt = _gnn_edge_endpoints(g, nodes)
        edges = edges + _gnn_mlp(pol, "edge_update", torch.cat([edges, src, tgt], dim=-1))
    out = _gnn_mlp(pol, "output", edges).squeeze(-1)
    # proportional allocation per supplying node over its outgoing internal edges (+ its self-loop) (:1435-1492, :111-138)
    alloc = out.clone()
    inv = _gnn_node_inventories(prob, obs)
    node_edges = {n: [] for n in range(g["n_nodes"])}
    for i, (a, _) in enumerate(g["internal"]):
        node_edges[a].append(i)
    for i, n in enumerate(g["supplying"]):
        node_edges[n].append(n_int + n_sup + n_dem + i)
    for n, idxs in node_edges.items():
        if not idxs:
            continue
        desired = torch.stack([out[:, i] for i in idxs], dim=1)
        scale = inv[:, n] / (desired.sum(dim=1) + 1e-10)
        if not pol.transshipment:
            scale = torch.clip(scale, max=1.0)
        scaled = desired * scale[:, None]
        for j, i in enumerate(idxs):
            alloc[:, i] = scaled[:, j]
    res = {}
    for kind, rows in g["mapping"].items():
        if not rows:
            continue
        n_cols = max(len(r) for r in rows)
        t = torch.zeros(B, len(rows), n_cols)
        for r, row in enumerate(rows):
            for j, e in enumerate(row):
                t[:, r, j] = alloc[:, e]
        res[kind] = t
    return res


def rollout(pol: OraclePolicy, periods, problem_params, data, observation_params,
            ignore_periods=0, discrete_allocation=False, keep_states=False, probe=None, zero_lead_orders="upstream") -> RolloutResult:
    """trainer.py:181-216.  PolicyLoss = reward.sum() (loss_functions.py:11-12).  probe: optional list that receives the
    quantile policies' (level, level - position) per period (see policy_act).  zero_lead_orders: see
    shift_pipeline_and_place ("upstream" = the reference bit for bit)."""
    env = env_reset(periods, problem_params, dict(data), observation_params)
    env.zero_lead_orders = zero_lead_orders
    total, reported = 0, 0
    per_period = []
    states = []
    for t in range(periods):
        if keep_states:
            states.append({k: v.detach().clone() for k, v in env.obs.items() if k.endswith("inventories")})
        obs_in = env.obs
        if probe is not None:
            obs_in = dict(env.obs)
            obs_in["_probe"] = probe
        if pol.name == "just_in_time":  # trainer.py:195-196: non-admissible policies read the simulator's internal data
            obs_in = dict(env.obs)
            obs_in["internal_data"] = {"demands": env.demands, "period_shift": env.period_shift}
        action = policy_act(pol, obs_in)
        if discrete_allocation:
            action = {k: v.round() for k, v in action.items()}  # trainer.py:201-202 (half-to-even)
        reward = env_step(env, action)
        r = reward.sum()
        total = total + r
        if t >= ignore_periods:
            reported = reported + r
        per_period.append(reward.detach())
    res = RolloutResult(total, reported, torch.stack(per_period), env.obs)
    if keep_states:
        res.states = states
    return res


def train_step_gradients(pol: OraclePolicy, periods, problem_params, data, observation_params, ignore_periods=0,
                         zero_lead_orders="upstream", keep_states=False):
    """Forward + backward of one batch exactly as trainer.py:160-173 (mean over B*T*S, then backward).
    Returns (rollout result, mean_loss value, list of grads aligned with pol.parameters())."""
    for p in pol.parameters():
        p.grad = None
    res = rollout(pol, periods, problem_params, data, observation_params, ignore_periods, zero_lead_orders=zero_lead_orders)
    B = len(data["demands"])
    mean_loss = res.total / (B * periods * problem_params["n_stores"])
    mean_loss.backward()
    return res, mean_loss.detach(), [p.grad.detach().clone() for p in pol.parameters()]


def epoch_losses(total, reported, n_samples, periods, ignore_periods, n_stores):
    """Normalisations returned by do_one_epoch (trainer.py:179)."""
    return total / (n_samples * periods * n_stores), reported / (n_samples * (periods - ignore_periods) * n_stores)
