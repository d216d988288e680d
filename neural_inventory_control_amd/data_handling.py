"""`Scenario`, `MyDataset`, `DatasetCreator`: the reference's scenario generator and dataset plumbing
(data_handling.py:3-458) behind the same constructor signatures and attribute names.

Two samplers produce the demand traces:
  * sampler="numpy" (default): the reference's host generators on numpy's legacy global RNG, seed for seed
    (data_handling.py:178-211).  Same YAML + seeds => bit-identical tensors to the reference (pinned by the golden
    fixtures), including its in-place mutations of `store_params['demand']` and `seeds['demand']`.
  * sampler="hip": csrc/sampler.hip — Philox4x32-10 keyed by (seed, GLOBAL scenario index, period), written straight into
    the [T][S][ldb] layout the env-step kernel reads.  Same distribution parameters; statistical (not bitwise) parity
    with numpy.

Sharded generation (one process per GPU): `Scenario(..., num_samples=n_local, scenario_offset=lo, num_total=N)`
builds rows [lo, lo + n_local) of the N-scenario job.  Demand traces are keyed by the global scenario index, the initial
store pipelines use the GLOBAL per-store demand mean (data_handling.py:298 — the one cross-scenario coupling of data
generation: an S-float SUM all-reduce over the ranks) and rows [lo, lo + n_local) of the global multiplier draw, so the
concatenated shards equal the single-process dataset.

`Alias`: `Scenarios = Scenario` (BASELINE.json's spelling).
"""
import copy

import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops
from .layout import pad_ld


class Scenario:
    def __init__(self, periods, problem_params, store_params, warehouse_params, echelon_params, num_samples,
                 observation_params, seeds=None, sampler="numpy", device=None, scenario_offset=0, num_total=None):
        self.problem_params = problem_params
        self.store_params = store_params
        self.warehouse_params = warehouse_params
        self.echelon_params = echelon_params
        self.num_samples = num_samples
        self.periods = periods
        self.observation_params = observation_params
        self.seeds = seeds
        self.sampler = sampler
        self.device = device
        self.scenario_offset = int(scenario_offset)
        self.num_total = int(num_total) if num_total is not None else self.scenario_offset + num_samples
        self.demands_soa = None  # [T][S][ldb] device tensor when sampler == "hip"

        self.demands = self.generate_demand_samples(problem_params, store_params, store_params["demand"], seeds)
        self.underage_costs = self.generate_data_for_samples_and_stores(problem_params, store_params["underage_cost"],
                                                                        seeds["underage_cost"], discrete=False)
        self.holding_costs = self.generate_data_for_samples_and_stores(problem_params, store_params["holding_cost"],
                                                                       seeds["holding_cost"], discrete=False)
        self.lead_times = self.generate_lead_times(problem_params, store_params["lead_time"], seeds["lead_time"])
        self.means, self.stds = self.generate_means_and_stds(observation_params, store_params)
        self.initial_inventories = self.generate_initial_inventories(problem_params, store_params, self.demands,
                                                                     self.lead_times, seeds["initial_inventory"])
        self.initial_warehouse_inventories = self.generate_initial_warehouse_inventory(warehouse_params)
        self.warehouse_lead_times = self.generate_warehouse_data(warehouse_params, "lead_time")
        self.warehouse_holding_costs = self.generate_warehouse_data(warehouse_params, "holding_cost")
        self.warehouse_edge_costs = (self.generate_warehouse_data(warehouse_params, "edge_cost")
                                     if warehouse_params and "edge_cost" in warehouse_params else None)
        self.initial_echelon_inventories = self.generate_initial_echelon_inventory(echelon_params)
        self.echelon_lead_times = self.generate_echelon_data(echelon_params, "lead_time")
        self.echelon_holding_costs = self.generate_echelon_data(echelon_params, "holding_cost")

        self.time_features, self.sample_features = {}, {}
        for kind, file_key, dest in (("time_features", "time_features_file", self.time_features),
                                     ("sample_features", "sample_features_file", self.sample_features)):
            if observation_params.get(kind) and observation_params.get(file_key):
                import pandas as pd
                table = pd.read_csv(observation_params[file_key])
                for k in observation_params[kind]:
                    col = torch.tensor(table[k].values)
                    if kind == "time_features":
                        dest[k] = col.unsqueeze(0).unsqueeze(0).expand(num_samples, problem_params["n_stores"], -1)
                    else:
                        if self.num_total != num_samples:  # a shard keeps its own rows of a per-sample feature column
                            col = col[self.scenario_offset: self.scenario_offset + num_samples]
                        dest[k] = col.unsqueeze(1).expand(-1, problem_params["n_stores"])
        self.split_by = self.define_how_to_split_data()

    # ---- public -------------------------------------------------------------------------------------------------
    def get_data(self):
        """float32 batch dict, None entries dropped (data_handling.py:54-81)."""
        data = {
            "demands": self.demands, "underage_costs": self.underage_costs, "holding_costs": self.holding_costs,
            "lead_times": self.lead_times, "mean": self.means, "std": self.stds,
            "initial_inventories": self.initial_inventories,
            "initial_warehouse_inventories": self.initial_warehouse_inventories,
            "warehouse_lead_times": self.warehouse_lead_times, "warehouse_holding_costs": self.warehouse_holding_costs,
            "warehouse_edge_costs": self.warehouse_edge_costs,
            "initial_echelon_inventories": self.initial_echelon_inventories,
            "echelon_holding_costs": self.echelon_holding_costs, "echelon_lead_times": self.echelon_lead_times,
        }
        data.update(self.time_features)
        data.update(self.sample_features)
        return {k: v.float() for k, v in data.items() if v is not None}

    def define_how_to_split_data(self):
        by_sample = ["underage_costs", "holding_costs", "lead_times", "initial_inventories"]
        by_period = []
        if self.problem_params["n_warehouses"] > 0:
            by_sample += ["initial_warehouse_inventories", "warehouse_lead_times", "warehouse_holding_costs",
                          "warehouse_edge_costs"]
        if self.problem_params["n_extra_echelons"] > 0:
            by_sample += ["initial_echelon_inventories", "echelon_holding_costs", "echelon_lead_times"]
        (by_period if self.store_params["demand"]["distribution"] == "real" else by_sample).append("demands")
        feats = self.observation_params["include_static_features"]
        for k in ("mean", "std"):
            if k in feats and feats[k]:
                by_sample.append(k)
        by_period += list(self.time_features.keys())
        by_sample += list(self.sample_features.keys())
        return {"sample_index": by_sample, "period": by_period}

    # ---- demand -------------------------------------------------------------------------------------------------
    def generate_demand_samples(self, problem_params, store_params, demand_params, seeds):
        self.generate_demand_parameters(problem_params, demand_params, seeds)
        self.adjust_seeds_for_consistency(problem_params, store_params, seeds)
        kind = demand_params["distribution"]
        if self.sampler == "hip" and kind in ("normal", "poisson"):
            self._device_sampler_args = (problem_params, demand_params, seeds["demand"])   # (bench.py times the same launch again)
            return self._generate_on_device(problem_params, demand_params, seeds["demand"])
        gen = {"normal": self.generate_normal_demand, "poisson": self.generate_poisson_demand,
               "real": self.read_real_demand_data}[kind]
        local, lo = self.num_samples, self.scenario_offset
        if kind != "real" and self.num_total != local:
            # a shard on the host path: numpy's stream is sequential, so the GLOBAL trace is drawn and rows [lo, lo + n) kept
            self.num_samples = self.num_total
        try:
            demand = gen(problem_params, demand_params, seeds["demand"])
        finally:
            self.num_samples = local
        if demand_params["clip"]:
            demand = np.clip(demand, 0, None)
        demand = torch.tensor(demand)
        if kind != "real" and self.num_total != local:
            self._global_store_mean = demand.float().mean(dim=2).mean(dim=0)  # the reference's formula on the global draw
            demand = demand[lo:lo + local]
        return demand

    def adjust_seeds_for_consistency(self, problem_params, store_params, seeds):
        """One-store synthetic settings shift the demand seed by lead time + 10 * underage cost (data_handling.py:155-158)."""
        if problem_params["n_warehouses"] == 0 and problem_params["n_stores"] == 1 \
                and store_params["demand"]["distribution"] != "real":
            try:
                seeds["demand"] = seeds["demand"] + int(store_params["lead_time"]["value"]
                                                        + 10 * store_params["underage_cost"]["value"])
            except Exception as e:
                print(f"Error: {e}")

    def read_real_demand_data(self, problem_params, demand_params, seed):
        lo = self.scenario_offset  # (a shard takes ITS rows of the file, not the first ones)
        return torch.load(demand_params["file_location"], map_location="cpu")[lo: lo + self.num_samples]

    def generate_demand_parameters(self, problem_params, demand_params, seeds):
        if demand_params["sample_across_stores"]:
            demand_params.update(self.sample_normal_mean_and_std(problem_params, demand_params, seeds))

    def _covariance(self, demand_params):
        stds = demand_params["std"]
        rho = demand_params["correlation"]
        return [[rho * a * b if i != j else a * b for i, a in enumerate(stds)] for j, b in enumerate(stds)]

    def generate_normal_demand(self, problem_params, demand_params, seed):
        if seed is not None:
            np.random.seed(seed)
        if problem_params["n_stores"] == 1:
            return np.random.normal(demand_params["mean"], demand_params["std"], size=(self.num_samples, 1, self.periods))
        draws = np.random.multivariate_normal(demand_params["mean"], cov=self._covariance(demand_params),
                                              size=(self.num_samples, self.periods))
        return np.transpose(draws, (0, 2, 1))

    def generate_poisson_demand(self, problem_params, demand_params, seed):
        if seed is not None:
            np.random.seed(seed)
        return np.random.poisson(demand_params["mean"], size=(self.num_samples, problem_params["n_stores"], self.periods))

    def _generate_on_device(self, problem_params, demand_params, seed, out=None, events=None):
        """HIP Philox sampler (csrc/sampler.hip); returns the reference-shaped (N, S, T) view of the SoA trace.  With `out` (a
        [T][S][ldb] buffer, bench.py's timing of the sampler launch) the trace is drawn into it and nothing else changes; `events`
        = (start, end) torch events recorded right around the launch (its operands already on the device)."""
        dev = torch.device(self.device or "cuda")
        S, N, T = problem_params["n_stores"], self.num_samples, self.periods
        mean = torch.as_tensor(np.broadcast_to(np.asarray(demand_params["mean"], dtype=np.float32), (S,)).copy()).to(dev)
        keep = out is None
        if keep:
            out = torch.zeros(T, S, pad_ld(N), device=dev)
        clip = bool(demand_params["clip"])
        mark = (lambda i: events[i].record()) if events is not None else (lambda i: None)
        if demand_params["distribution"] == "poisson":
            mark(0)
            ops.sample_demand(out, T, S, N, self.scenario_offset, int(seed), 1, mean, None, clip)
        else:
            std = torch.as_tensor(np.broadcast_to(np.asarray(demand_params["std"], dtype=np.float32), (S,)).copy()).to(dev)
            rho = float(demand_params.get("correlation", 0.0) or 0.0) if S > 1 else 0.0
            if 0.0 <= rho <= 1.0:
                # the reference's covariance (rho s_i s_j off the diagonal, data_handling.py:194-201) through its
                # one-factor form: S + 1 normals per (scenario, period), no factor matrix
                mark(0)
                ops.sample_demand_equicorrelated(out, T, S, N, self.scenario_offset, int(seed), mean, std, rho, clip)
            else:  # a negative correlation has no one-factor form: general Cholesky path
                cov = np.asarray(self._covariance(demand_params), dtype=np.float64)
                chol = torch.as_tensor(np.linalg.cholesky(cov).astype(np.float32)).contiguous().to(dev)
                mark(0)
                ops.sample_demand(out, T, S, N, self.scenario_offset, int(seed), 0, mean, chol, clip)
        mark(1)
        if keep:
            self.demands_soa = out
        return out[:, :, :N].permute(2, 1, 0)

    def sample_normal_mean_and_std(self, problem_params, demand_params, seeds):
        np.random.seed(seeds["mean"])
        means = np.random.uniform(demand_params["mean_range"][0], demand_params["mean_range"][1],
                                  problem_params["n_stores"]).round(3)
        np.random.seed(seeds["coef_of_var"])
        cv = np.random.uniform(demand_params["coef_of_var_range"][0], demand_params["coef_of_var_range"][1],
                               problem_params["n_stores"])
        return {"mean": means, "std": (means * cv).round(3)}

    # ---- static per-(sample, store) tables ----------------------------------------------------------------------
    def generate_data_for_samples_and_stores(self, problem_params, cost_params, seed, discrete=False):
        np.random.seed(seed)
        p = copy.deepcopy(cost_params)
        flag = lambda k: p.get(k, False)  # noqa: E731  (missing keys read as False, data_handling.py:248)
        draw = np.random.randint if discrete else np.random.uniform
        S = problem_params["n_stores"]
        # a shard (scenario_offset / num_total) takes rows [lo, lo + n) of what the single-process job draws or reads; the
        # GLOBAL maximum of a discrete table is kept for `generate_initial_inventories` (its slot count must not depend on
        # the shard)
        lo, n = self.scenario_offset, self.num_samples
        self._last_table_global_max = None
        if flag("file_location"):
            whole = torch.load(p["file_location"], map_location="cpu")[: self.num_total]
            self._last_table_global_max = whole.max().item() if whole.numel() else None
            p["value"] = whole[lo: lo + n]
        if flag("sample_across_stores"):
            return torch.tensor(draw(*p["range"], S)).expand(self.num_samples, -1)
        if flag("vary_across_samples"):
            whole = torch.tensor(draw(*p["range"], self.num_total))
            self._last_table_global_max = whole.max().item()
            return whole[lo: lo + n].unsqueeze(1).expand(-1, S)
        if flag("expand"):
            v = torch.tensor(p["value"])
            if v.dim() == 2:  # [n_stores, n_warehouses] lead-time matrix
                return v.unsqueeze(0).expand(self.num_samples, -1, -1)
            return v.expand(self.num_samples, S)
        return torch.tensor(p["value"])

    def generate_lead_times(self, problem_params, lead_time_params, seed):
        raw = self.generate_data_for_samples_and_stores(problem_params, lead_time_params, seed, discrete=True)
        self._lead_time_global_max = self._last_table_global_max
        if raw.dim() == 2:
            nw = problem_params.get("n_warehouses", 0)
            raw = raw.unsqueeze(2).expand(-1, -1, nw) if nw > 0 else raw.unsqueeze(2)
        return raw.to(torch.int64)

    def generate_initial_inventories(self, problem_params, store_params, demands, lead_times, seed):
        np.random.seed(seed)
        spec = store_params["initial_inventory"]
        S = problem_params["n_stores"]
        if not spec["sample"]:
            return torch.zeros(self.num_samples, S, spec["inventory_periods"])
        demand_mean = self.global_store_demand_mean(demands)
        lt_max = getattr(self, "_lead_time_global_max", None)   # per-sample lead times: the maximum over the WHOLE job
        slots = max(spec["inventory_periods"], int(lt_max) if lt_max is not None else lead_times.max().item())
        mults = self._uniform_rows(spec["range_mult"], S * slots).reshape(self.num_samples, S, slots)
        return demand_mean[None, :, None] * torch.from_numpy(mults)  # f32 x f64 -> f64, cast to f32 in get_data

    def global_store_demand_mean(self, demands):
        """Per-store mean demand over EVERY scenario of the job and every period (data_handling.py:298).
        Host sampler: the reference's float32 mean-of-means (bit-equal; a shard computed it on the global draw).
        Device sampler: the traces exist only on their shard, so the per-store float64 sums are added across ranks (one
        S-element all-reduce) and divided by the global count; a single process uses the same formula, which makes the
        result independent of the number of shards (up to float64 rounding of the partial sums, far below float32)."""
        if getattr(self, "_global_store_mean", None) is not None:
            return self._global_store_mean
        if self.demands_soa is None:
            return demands.float().mean(dim=2).mean(dim=0).cpu()
        from . import parallel
        part = self.demands_soa[:, :, :self.num_samples].double().sum(dim=(0, 2))
        if self.num_total != self.num_samples or parallel.active():
            part = parallel.all_reduce_sum(part)
        return (part / float(self.num_total * self.periods)).float().cpu()

    def _uniform_rows(self, rng, row_len):
        """Rows [scenario_offset, scenario_offset + num_samples) of the (num_total, row_len) uniform draw the single-process
        job makes at this point of numpy's global stream (the preceding rows are drawn in bounded chunks and dropped)."""
        skip = self.scenario_offset * row_len
        while skip > 0:
            k = min(skip, 1 << 22)
            np.random.uniform(rng[0], rng[1], size=k)
            skip -= k
        return np.random.uniform(rng[0], rng[1], size=(self.num_samples, row_len))

    def generate_initial_warehouse_inventory(self, warehouse_params):
        if warehouse_params is None:
            return None
        lt = warehouse_params["lead_time"]
        return torch.zeros(self.num_samples, self.problem_params["n_warehouses"], max(lt) if isinstance(lt, list) else lt)

    def generate_initial_echelon_inventory(self, echelon_params):
        if echelon_params is None:
            return None
        return torch.zeros(self.num_samples, len(echelon_params["lead_time"]), max(echelon_params["lead_time"]))

    def generate_warehouse_data(self, warehouse_params, key):
        if warehouse_params is None:
            return None
        nw = self.problem_params["n_warehouses"]
        value = warehouse_params[key]
        if isinstance(value, list):
            if len(value) != nw:
                raise ValueError(f"warehouse_params['{key}'] list length {len(value)} doesn't match n_warehouses {nw}")
            return torch.tensor(value).unsqueeze(0).expand(self.num_samples, -1)
        return torch.tensor([value]).expand(self.num_samples, nw)

    def generate_echelon_data(self, echelon_params, key):
        if echelon_params is None:
            return None
        return torch.tensor(echelon_params[key]).unsqueeze(0).expand(self.num_samples, -1)

    def generate_means_and_stds(self, observation_params, store_params):
        out = {"mean": None, "std": None}
        feats = observation_params["include_static_features"]
        for k in out:
            if k in feats and feats[k]:
                out[k] = torch.tensor(store_params["demand"][k]).unsqueeze(0).expand(self.num_samples, -1)
        return out["mean"], out["std"]


Scenarios = Scenario


class MyDataset(Dataset):
    """data_handling.py:385-395.  `tensors_on(device)` additionally makes the whole dataset device-resident so that
    `DeviceBatches` can slice batches without the per-sample __getitem__ + collate of the reference's DataLoader path
    (1.16 s per 32k-sample epoch)."""

    def __init__(self, num_samples, data):
        self.data = data
        self.num_samples = num_samples

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        return {k: v[idx] for k, v in self.data.items()}

    def tensors_on(self, device):
        return {k: v.to(device) for k, v in self.data.items()}


class DeviceBatches:
    """Iterable of device-resident batches: a drop-in for `DataLoader(dataset, batch_size, shuffle)` in
    `Trainer.do_one_epoch` (needs only `__iter__` and `.dataset`).  Shuffling draws ONE permutation per epoch from a
    generator seeded identically on every rank, and each rank takes its contiguous slice of every global batch
    (SURVEY §8e 'Train shuffle')."""

    def __init__(self, dataset, batch_size, shuffle=False, device="cuda", seed=0, rank=0, world_size=1):
        self.dataset = dataset
        self.batch_size, self.shuffle = batch_size, shuffle
        self.device = device
        self.rank, self.world_size = rank, world_size
        self._gen = torch.Generator().manual_seed(seed)
        self._data = dataset.tensors_on(device)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._gen) if self.shuffle else None
        for lo in range(0, n, self.batch_size):
            hi = min(lo + self.batch_size, n)
            per = (hi - lo + self.world_size - 1) // self.world_size
            a, b = min(lo + self.rank * per, hi), min(lo + (self.rank + 1) * per, hi)
            self.last_global_batch = hi - lo
            if order is None:
                yield {k: v[a:b] for k, v in self._data.items()}
            else:
                idx = order[a:b].to(self.device)
                yield {k: v.index_select(0, idx) for k, v in self._data.items()}


class DatasetCreator:
    """data_handling.py:398-458."""

    def create_datasets(self, scenario, split=True, by_period=False, by_sample_indexes=False, periods_for_split=None,
                        sample_index_for_split=None):
        if not split:
            return self.create_single_dataset(scenario.get_data())
        if by_period:
            return [self.create_single_dataset(d) for d in self.split_by_period(scenario, periods_for_split)]
        if by_sample_indexes:
            train, dev = self.split_by_sample_index(scenario, sample_index_for_split)
            return self.create_single_dataset(train), self.create_single_dataset(dev)
        raise NotImplementedError

    def split_by_sample_index(self, scenario, sample_index_for_split):
        """dev = first rows, train = the rest (so the train size never changes the dev set)."""
        data = scenario.get_data()
        dev = {k: v[:sample_index_for_split] for k, v in data.items()}
        train = {k: v[sample_index_for_split:] for k, v in data.items()}
        return train, dev

    def split_by_period(self, scenario, periods_for_split):
        data = scenario.get_data()
        common = {k: data[k] for k in scenario.split_by["sample_index"] if k in data}
        out = []
        for period_range in periods_for_split:
            this = copy.deepcopy(common)
            sl = slice(*map(int, period_range.strip("() ").split(",")))
            for k in scenario.split_by["period"]:
                if k in data:
                    this[k] = data[k][:, :, sl]
            out.append(this)
        return out

    def create_single_dataset(self, data):
        return MyDataset(len(data["initial_inventories"]), data)
