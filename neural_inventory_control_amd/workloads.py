"""BASELINE.json's five configurations as setting / policy dicts in the reference's YAML schema (same keys and values
as its config_files/settings/*.yml and policies_and_hyperparams/*.yml), for bench.py and examples.
Each call returns fresh deep copies because `Scenario` mutates its inputs."""
import copy

_SEEDS = {"underage_cost": 28, "holding_cost": 73, "mean": 33, "coef_of_var": 92, "lead_time": 41, "demand": 57,
          "initial_inventory": 4839}


def _const(v):
    return {"sample_across_stores": False, "vary_across_samples": False, "expand": True, "value": v}


def _per_store(lo, hi):
    return {"sample_across_stores": True, "vary_across_samples": False, "expand": False, "range": [lo, hi]}


def _obs(warehouse, moments):
    feats = {"holding_costs": True, "underage_costs": True, "lead_times": True}
    if moments:
        feats.update({"mean": True, "std": True})
    return {"include_warehouse_inventory": warehouse, "include_static_features": feats,
            "demand": {"past_periods": 0, "period_shift": 0}, "include_days_to_christmas": False,
            "time_features": None, "sample_features": None}


def _mlp(name, hidden, out=None, ub_mult=None):
    p = {"name": name, "inner_layer_activations": {"master": "elu"}, "output_layer_activation": {"master": None},
         "neurons_per_hidden_layer": {"master": list(hidden)}, "output_sizes": {"master": out}, "initial_bias": None}
    if ub_mult is not None:
        p["warehouse_upper_bound_mult"] = ub_mult
    return p


def one_store(lost, poisson):
    demand = ({"sample_across_stores": False, "expand": True, "mean": 5.0, "distribution": "poisson", "clip": True}
              if poisson else
              {"sample_across_stores": False, "expand": True, "mean": 5.0, "std": 1.6, "distribution": "normal", "clip": True})
    return {
        "seeds": dict(_SEEDS),
        "problem_params": {"n_stores": 1, "n_warehouses": 0, "n_extra_echelons": 0, "lost_demand": lost,
                           "maximize_profit": False},
        "observation_params": _obs(False, False),
        "store_params": {"demand": demand, "lead_time": _const(4), "holding_cost": _const(1),
                         "underage_cost": _const(9.0),
                         "initial_inventory": {"sample": True, "range_mult": [0, 1], "inventory_periods": 4}},
        "warehouse_params": None, "echelon_params": None,
    }


def one_warehouse(n_stores=16):
    return {
        "seeds": dict(_SEEDS),
        "problem_params": {"n_stores": n_stores, "n_warehouses": 1, "n_extra_echelons": 0, "lost_demand": True,
                           "maximize_profit": False},
        "observation_params": _obs(True, True),
        "store_params": {
            "demand": {"sample_across_stores": True, "mean_range": [2.5, 7.5], "coef_of_var_range": [0.25, 0.50],
                       "distribution": "normal", "correlation": 0.5, "clip": True},
            "lead_time": _per_store(2, 4), "holding_cost": _per_store(0.7, 1.3), "underage_cost": _per_store(6.3, 11.7),
            "initial_inventory": {"sample": True, "range_mult": [0, 1], "inventory_periods": 3}},
        "warehouse_params": {"holding_cost": 0.3, "lead_time": 3}, "echelon_params": None,
    }


def serial_system():
    s = one_store(lost=False, poisson=False)
    s["store_params"]["demand"]["std"] = 2.0
    s["problem_params"].update({"n_warehouses": 1, "n_extra_echelons": 2})
    s["observation_params"] = _obs(True, False)
    s["warehouse_params"] = {"holding_cost": 0.5, "lead_time": 3}
    s["echelon_params"] = {"holding_cost": [0.1, 0.2], "lead_time": [2, 4]}
    return s


def many_warehouses(n_stores=64, n_warehouses=3, seed=0):
    """cfg5: synthetic [Wn][S] adjacency (every store served by 1-2 warehouses) and [S][Wn] lead times in 1..6
    (0 = not connected), in the format of the reference's many_warehouses_lost_demand.yml:31-34,79-91."""
    import random
    rnd = random.Random(seed)
    adj = [[0] * n_stores for _ in range(n_warehouses)]
    lead = [[0] * n_warehouses for _ in range(n_stores)]
    for s in range(n_stores):
        for w in rnd.sample(range(n_warehouses), rnd.choice([1, 2])):
            adj[w][s] = 1
            lead[s][w] = rnd.randint(1, 6)
    s_ = one_warehouse(n_stores)
    s_["problem_params"].update({"n_warehouses": n_warehouses, "warehouse_store_adjacency": adj})
    s_["store_params"]["lead_time"] = _const(lead)
    s_["warehouse_params"] = {"holding_cost": [0.3, 0.4, 0.2][:n_warehouses], "lead_time": 3,
                              "edge_cost": [0.5, 1.5, 0.7][:n_warehouses]}
    return s_


def many_warehouses_dense(n_stores=16, n_warehouses=3, seed=0):
    """A many-warehouse graph the reference's GNN handles consistently (its action columns are "j-th connected warehouse",
    neural_networks.py:1423-1428: only with every store connected to every warehouse is column j warehouse j)."""
    import random
    rnd = random.Random(seed)
    s_ = many_warehouses(n_stores, n_warehouses, seed)
    s_["problem_params"]["warehouse_store_adjacency"] = [[1] * n_stores for _ in range(n_warehouses)]
    s_["store_params"]["lead_time"] = _const([[rnd.randint(1, 6) for _ in range(n_warehouses)] for _ in range(n_stores)])
    return s_


def _closed_form(name, n_out, out_act="softplus", bias=10.0):
    """base_stock.yml / capped_base_stock.yml (softplus output, initial bias 10) and echelon_stock.yml (neither)."""
    return {"name": name, "inner_layer_activations": {"master": None}, "output_layer_activation": {"master": out_act},
            "neurons_per_hidden_layer": {"master": []}, "output_sizes": {"master": n_out}, "initial_bias": {"master": bias}}


def gnn_policy():
    """gnn.yml of the reference: five 32-wide MLPs (elu inside; softplus on the per-edge output, bias 5.0)."""
    mods = ("initial_node", "initial_edge", "node_update", "edge_update", "output")
    return {"name": "gnn",
            "inner_layer_activations": {m: "elu" for m in mods},
            "output_layer_activation": {**{m: "elu" for m in mods}, "output": "softplus"},
            "neurons_per_hidden_layer": {m: [32, 32] for m in mods},
            "output_sizes": {**{m: 32 for m in mods}, "output": 1},
            "initial_bias": {"output": 5.0}, "gradient_clipping_norm_value": 1.0}


def real_data(n_products=288, n_stores=21, n_warehouses=3, weeks=171, past_periods=16, seed=0):
    """The shape of the reference's real-data setting (many_warehouses_real_data_lost_demand.yml: 288 Favorita products x 21
    stores x 171 weeks, 3 warehouses, 16 past demands + days-from-christmas in the observation, profit objective) on SYNTHETIC
    stand-in files - the Favorita blobs are not redistributable with this repository and cannot travel to the GPU box: a
    weekly-sales tensor file and a dates csv with the same layout are written to a temporary directory, so the whole real-data
    pipeline (file-backed demand, period split, past-demand window, time features) runs as it does upstream."""
    import os
    import random
    import tempfile
    import numpy as np
    import torch
    rnd = random.Random(seed)
    d = tempfile.mkdtemp(prefix="nic_real_data_")
    gen = torch.Generator().manual_seed(seed)
    level = torch.rand(n_products, n_stores, 1, generator=gen) * 20.0 + 1.0
    season = 1.0 + 0.3 * torch.sin(torch.arange(weeks, dtype=torch.float32) * (2 * 3.14159265 / 52.0))
    sales = torch.clamp(level * season + torch.randn(n_products, n_stores, weeks, generator=gen) * level.sqrt() * 2.0, min=0.0)
    torch.save(sales.round(decimals=3), os.path.join(d, "weekly_sales.pt"))
    with open(os.path.join(d, "dates_with_info.csv"), "w") as f:
        f.write("week,days_from_christmas\n")
        for w in range(weeks):
            f.write(f"{w},{(w * 7 + 180) % 365 - 182}\n")
    adj = [[1] * n_stores for _ in range(n_warehouses)]
    for w in range(n_warehouses):
        for s_ in rnd.sample(range(n_stores), min(2, n_stores - 1)):   # (two stores per warehouse lose their edge)
            adj[w][s_] = 0
    for s_ in range(n_stores):  # every store keeps at least one supplier
        if not any(adj[w][s_] for w in range(n_warehouses)):
            adj[0][s_] = 1
    lead = [[(rnd.randint(1, 2) if w % 3 == 1 else rnd.randint(4, 6)) for w in range(max(n_warehouses, 3))][:n_warehouses]
            for _ in range(n_stores)]
    cyc = lambda v: [v[w % len(v)] for w in range(n_warehouses)]  # noqa: E731   (more than three warehouses: the values repeat)
    feats = {"holding_costs": True, "underage_costs": True, "lead_times": True, "mean": False, "std": False}
    return {
        "seeds": dict(_SEEDS, warehouse=10),
        "problem_params": {"n_stores": n_stores, "n_warehouses": n_warehouses, "n_extra_echelons": 0, "lost_demand": True,
                           "maximize_profit": True, "warehouse_store_adjacency": adj},
        "observation_params": {"include_warehouse_inventory": True, "include_static_features": feats,
                               "demand": {"past_periods": past_periods, "period_shift": past_periods},
                               "time_features_file": os.path.join(d, "dates_with_info.csv"),
                               "time_features": ["days_from_christmas"], "sample_features": None,
                               "include_days_to_christmas": True},
        "store_params": {
            "demand": {"distribution": "real", "file_location": os.path.join(d, "weekly_sales.pt"), "sample_across_stores": False,
                       "expand": False, "clip": False, "decimals": 3},
            "lead_time": _const(lead), "holding_cost": _per_store(0.7, 1.3),
            "underage_cost": {"sample_across_stores": True, "vary_across_samples": True, "expand": False, "range": [6.3, 11.7]},
            "initial_inventory": {"sample": False, "inventory_periods": 6}},
        "warehouse_params": {"holding_cost": cyc([0.3, 0.4, 0.2]), "lead_time": 3, "edge_cost": cyc([0.5, 1.5, 0.7])},
        "echelon_params": None,
    }


def real_data_one_store(n_products=32768, weeks=171, past_periods=16, seed=0):
    """The shape of one_store_real_data_lost_demand.yml (32,768 (product, store) series x 171 weeks, lead times 4..6 and underage
    costs per sample, 16 past demands + days-from-christmas in the observation, profit objective) on SYNTHETIC stand-in files
    (see `real_data`)."""
    import os
    import tempfile
    import torch
    d = tempfile.mkdtemp(prefix="nic_real_data_1s_")
    gen = torch.Generator().manual_seed(seed)
    level = torch.rand(n_products, 1, 1, generator=gen) * 20.0 + 1.0
    season = 1.0 + 0.3 * torch.sin(torch.arange(weeks, dtype=torch.float32) * (2 * 3.14159265 / 52.0))
    sales = torch.clamp(level * season + torch.randn(n_products, 1, weeks, generator=gen) * level.sqrt() * 2.0, min=0.0)
    torch.save(sales.round(decimals=3), os.path.join(d, "weekly_sales.pt"))
    with open(os.path.join(d, "dates_with_info.csv"), "w") as f:
        f.write("week,days_from_christmas\n")
        for w in range(weeks):
            f.write(f"{w},{(w * 7 + 180) % 365 - 182}\n")
    feats = {"holding_costs": True, "underage_costs": True, "lead_times": True, "upper_bounds": False}
    return {
        "seeds": dict(_SEEDS),
        "problem_params": {"n_stores": 1, "n_warehouses": 0, "n_extra_echelons": 0, "lost_demand": True, "maximize_profit": True},
        "observation_params": {"include_warehouse_inventory": False, "include_static_features": feats,
                               "demand": {"past_periods": past_periods, "period_shift": past_periods},
                               "time_features_file": os.path.join(d, "dates_with_info.csv"),
                               "time_features": ["days_from_christmas"], "sample_features": None},
        "store_params": {
            "demand": {"distribution": "real", "file_location": os.path.join(d, "weekly_sales.pt"), "sample_across_stores": False,
                       "expand": False, "clip": False, "decimals": 3},
            "lead_time": {"sample_across_stores": False, "vary_across_samples": True, "expand": False, "range": [4, 7]},
            "holding_cost": _const(1.0),
            "underage_cost": {"sample_across_stores": False, "vary_across_samples": True, "expand": False, "range": [6.3, 11.7]},
            "initial_inventory": {"sample": False, "inventory_periods": 6}},
        "warehouse_params": None,
        "echelon_params": None,
    }


def standin_forecaster(lead_times=(4, 5, 6), seed=0):
    """A file in the format of the reference's `quantile_forecasters/<id>.pt` (FullyConnectedForecaster [128, 128] over 16 past
    demands + days from christmas, 19 quantiles x 3 lead times) with RANDOM weights, monotone in the quantile: the shipped file
    does not travel to the GPU box; the quantile policies' arithmetic and cost do not depend on what the numbers are."""
    import os
    import tempfile
    import torch
    gen = torch.Generator().manual_seed(seed)
    n_out = 19 * len(lead_times)
    w = lambda *shape: torch.randn(*shape, generator=gen) * 0.05   # noqa: E731
    bias = (torch.arange(19, dtype=torch.float32).repeat_interleave(len(lead_times)) + 5.0) * 4.0   # (quantile-major rows)
    state = {"net.0.weight": w(128, 17), "net.0.bias": w(128), "net.2.weight": w(128, 128), "net.2.bias": w(128),
             "net.4.weight": w(n_out, 128), "net.4.bias": bias}
    path = os.path.join(tempfile.mkdtemp(prefix="nic_forecaster_"), "forecaster.pt")
    torch.save(state, path)
    return path


def transformed_nv_policy():
    """transformed_nv.yml: a 32 x 32 sigmoid net over the newsvendor quantile, inverted by the frozen quantile forecaster."""
    return {"name": "transformed_nv", "inner_layer_activations": {"master": "elu"}, "output_layer_activation": {"master": "sigmoid"},
            "neurons_per_hidden_layer": {"master": [32, 32]}, "initial_bias": None, "output_sizes": {"master": 1},
            "forecaster_location": standin_forecaster(), "forecaster_lead_times": [4, 5, 6]}


def data_driven_policy():
    """data_driven_net.yml: one 64 x 64 MLP over every observed feature, relu outputs."""
    return {"name": "data_driven", "inner_layer_activations": {"master": "elu"}, "output_layer_activation": {"master": "relu"},
            "neurons_per_hidden_layer": {"master": [64, 64]}, "initial_bias": {"master": 1.0}, "output_sizes": {"master": None}}


def symmetry_aware_policy(context=64):
    """Layer sizes are this repository's choice (no YAML survives upstream): a `context`-wide context vector, 32-wide store and
    warehouse nets; softplus on the store net's desired order, sigmoid x upper bound on the warehouse's."""
    mods = ("context", "store", "warehouse")
    return {"name": "symmetry_aware",
            "inner_layer_activations": {m: "elu" for m in mods},
            "output_layer_activation": {"context": "elu", "store": "softplus", "warehouse": "sigmoid"},
            "neurons_per_hidden_layer": {"context": [256], "store": [32, 32], "warehouse": [32, 32]},
            "output_sizes": {"context": context, "store": 1, "warehouse": 1},
            "initial_bias": None, "warehouse_upper_bound_mult": 4}


WORKLOADS = {
    # name: (setting builder, policy dict, scenarios per GPU, periods, description)
    "cfg1": (lambda: one_store(True, True), _mlp("vanilla_one_store", [32, 32, 32], 1), 256, 50,
             "one_store_lost + vanilla_one_store, 256 scenarios x T=50"),
    "cfg2": (lambda: one_store(False, False), _mlp("vanilla_one_store", [32, 32, 32], 1), 32768, 100,
             "one_store_backlogged + vanilla_one_store, 32768 scenarios x T=100"),
    "cfg3": (lambda: one_warehouse(16), _mlp("vanilla_warehouse", [512, 512, 512], None, 4), 65536, 100,
             "one_warehouse_lost_demand, 16 stores, 65536 scenarios x T=100, vanilla_warehouse 512x3"),
    # cfg3's shard on ONE of 8 GPUs when the 65,536 scenarios are strong-scaled (round 4: the small-batch regime)
    "cfg3_shard8": (lambda: one_warehouse(16), _mlp("vanilla_warehouse", [512, 512, 512], None, 4), 8192, 100,
                    "one_warehouse_lost_demand, 16 stores, 8192 scenarios (= cfg3's 65536 / 8 GPUs) x T=100, vanilla_warehouse 512x3"),
    # the reference's SHIPPED training batch (one_warehouse_lost_demand.yml:22,31-34 + vanilla_warehouse.yml): 5 stores, batches
    # of 1,024 scenarios x T=50
    "cfg3_batch1024": (lambda: one_warehouse(5), _mlp("vanilla_warehouse", [512, 512, 512], None, 4), 1024, 50,
                       "one_warehouse_lost_demand as shipped (5 stores), one batch of 1024 scenarios x T=50, vanilla_warehouse 512x3"),
    "cfg4": (serial_system, _mlp("vanilla_serial", [32, 32], 4, 4), 16384, 100,
             "serial_system 4 echelons, 16384 scenarios/GPU x T=100, vanilla_serial"),
    "cfg5": (lambda: many_warehouses(64, 3), _mlp("vanilla_warehouse", [512, 512, 512], None, 4), 32768, 70,
             "many_warehouses_lost_demand 3x64 stores, 32768 scenarios/GPU x T=70, vanilla_warehouse 512x3"),
    # closed-form policies (base_stock.yml, echelon_stock.yml): a few trainable levels, everything else arithmetic - the
    # "env-only" number of SURVEY 8(d): whole horizon + gradient in one launch, the demand trace is the only HBM stream
    "base_stock": (lambda: one_store(False, False), _closed_form("base_stock", 1), 32768, 100,
                   "one_store_backlogged + base_stock, 32768 scenarios x T=100"),
    "base_stock_1m": (lambda: one_store(False, False), _closed_form("base_stock", 1), 1 << 20, 100,
                      "one_store_backlogged + base_stock, 1,048,576 scenarios x T=100 (enough chains in flight to be HBM-bound)"),
    "echelon_stock": (serial_system, _closed_form("echelon_stock", 4, None, None), 131072, 100,
                      "serial_system 4 echelons + echelon_stock, 131072 scenarios x T=100"),
    # BASELINE cfg3 "as worded": the symmetry-aware policy (recovered per SURVEY 2.2; parity unpinned) on the generic route
    "cfg3_symmetry_aware": (lambda: one_warehouse(16), symmetry_aware_policy(), 8192, 50,
                            "one_warehouse_lost_demand, 16 stores, 8192 scenarios x T=50, symmetry_aware (context 256->64, store / "
                            "warehouse nets 32x2); generic route"),
    # the reference's real-data training batch (72 of 288 products x 21 stores x 3 warehouses, T = 95 of 111 train
    # weeks after the 16-week past-demand window).  Since round 3 on the MLP engine (per-period kernels, data_driven head);
    # `bench.py --generic-route` runs it through Simulator.step + HipLinear layers + autograd instead (`--graph`: replayed)
    "real_data_driven": (real_data, data_driven_policy(), 72, 95,
                         "many_warehouses_real_data_lost_demand shape (synthetic stand-in files), 72 products x 21 stores x "
                         "3 warehouses x T=95, data_driven 64x64"),
    # the GNN policy on cfg3's graph (fused gather-MLP kernels over the static supply graph, gnn_rollout.py)
    "gnn": (lambda: one_warehouse(16), gnn_policy(), 8192, 50,
            "one_warehouse_lost_demand, 16 stores, 8192 scenarios x T=50, gnn (5 x 32-wide MLPs, 1 message-passing step)"),
    # the same engine on a many-warehouse graph (round 3): 3 warehouses x 16 stores, all connected: 19 nodes, 70 edges
    "gnn_many_warehouses": (lambda: many_warehouses_dense(16, 3), gnn_policy(), 8192, 50,
                            "many_warehouses_lost_demand shape, 3 warehouses x 16 stores (dense), 8192 scenarios x T=50, gnn"),
}


def many_warehouses_shipped():
    """many_warehouses_lost_demand.yml exactly as the reference ships it (:22-34, 66-110): 10 stores, 2 warehouses."""
    s_ = one_warehouse(10)
    s_["problem_params"].update({"n_warehouses": 2, "warehouse_store_adjacency": [[0, 0, 1, 1, 1, 0, 1, 1, 1, 0],
                                                                                   [1, 1, 1, 1, 1, 1, 0, 1, 1, 1]]})
    s_["store_params"]["lead_time"] = _const([[0, 2], [0, 1], [6, 2], [6, 2], [6, 1], [0, 2], [6, 0], [5, 2], [6, 2], [0, 2]])
    s_["warehouse_params"] = {"holding_cost": [0.3, 0.4], "lead_time": 3, "edge_cost": [0.5, 1.5]}
    return s_


def _as_shipped(setting):
    """The keys main_run reads besides the setting itself, with the values both warehouse YAMLs ship
    (one_warehouse_lost_demand.yml:10-45): 8,192 training samples in batches of 1,024 x 50 periods, one dev batch of 8,192 x 100."""
    s_ = dict(setting)
    s_["test_seeds"] = dict(_SEEDS, demand=65)
    s_["sample_data_params"] = {"split_by_period": False}
    s_["params_by_dataset"] = {"train": {"n_samples": 8192, "batch_size": 1024, "periods": 50, "ignore_periods": 30},
                               "dev": {"n_samples": 8192, "batch_size": 8192, "periods": 100, "ignore_periods": 60},
                               "test": {"n_samples": 8192, "batch_size": 8192, "periods": 5000, "ignore_periods": 3000}}
    return s_


def _hyperparams(policy, lr=0.0003):
    """vanilla_warehouse.yml's trainer / optimizer blocks around a policy dict."""
    return {"trainer_params": {"epochs": 20000, "do_dev_every_n_epochs": 10, "early_stopping_patience_epochs": 500,
                               "print_results_every_n_epochs": 1, "save_model": False, "epochs_between_save": 10,
                               "choose_best_model_on": "dev_loss", "load_previous_model": False, "load_model_path": None},
            "optimizer_params": {"learning_rate": lr}, "nn_params": policy}


# The reference's shipped YAML pairs as they are (main_run.py train <setting> vanilla_warehouse): bench.py times whole TRAINING
# EPOCHS of these through `Trainer.do_one_epoch` - shuffled device-resident batches, rollout, backward, Adam (trainer.py:143-179).
EPOCH_WORKLOADS = {
    "cfg3_yaml": (lambda: _as_shipped(one_warehouse(5)), lambda: _hyperparams(_mlp("vanilla_warehouse", [512, 512, 512], None, 4)),
                  "one_warehouse_lost_demand.yml + vanilla_warehouse.yml as shipped: 5 stores, 8192 samples in batches of 1024 x "
                  "T=50 (ignore 30), one training epoch through Trainer.do_one_epoch"),
    "cfg5_yaml": (lambda: _as_shipped(many_warehouses_shipped()),
                  lambda: _hyperparams(_mlp("vanilla_warehouse", [512, 512, 512], None, 4)),
                  "many_warehouses_lost_demand.yml + vanilla_warehouse.yml as shipped: 10 stores x 2 warehouses, 8192 samples in "
                  "batches of 1024 x T=50 (ignore 30), one training epoch through Trainer.do_one_epoch"),
}


def _real_data_as_shipped():
    """many_warehouses_real_data_lost_demand.yml:14-57 around the stand-in files of `real_data()`: 288 products, datasets split BY
    PERIOD (train weeks 0-111, dev 88-141, test 118-171), training batches of 72 products x 95 weeks (the first 16 ignored)."""
    s_ = real_data()
    s_["test_seeds"] = dict(_SEEDS, demand=65)
    s_["sample_data_params"] = {"split_by_period": True, "train_periods": "(0, 111)", "dev_periods": "(88, 141)",
                                "test_periods": "(118, 171)"}
    s_["params_by_dataset"] = {"train": {"n_samples": 288, "batch_size": 72, "periods": 95, "ignore_periods": 16},
                               "dev": {"n_samples": 288, "batch_size": 288, "periods": 37, "ignore_periods": 16},
                               "test": {"n_samples": 288, "batch_size": 288, "periods": 37, "ignore_periods": 16}}
    return s_


EPOCH_WORKLOADS["real_data_yaml"] = (
    _real_data_as_shipped, lambda: _hyperparams(data_driven_policy(), lr=0.003),
    "many_warehouses_real_data_lost_demand.yml + data_driven_net.yml as shipped (synthetic stand-in files): 288 products x 21 stores "
    "x 3 warehouses, 4 batches of 72 x T=95 (ignore 16), one training epoch through Trainer.do_one_epoch")


def _real_one_store_as_shipped():
    """one_store_real_data_lost_demand.yml:19-45: 32,768 series split BY PERIOD, training batches of 8,192 x 95 weeks (16 ignored),
    dev / test batches of 32,768 x 37."""
    s_ = real_data_one_store()
    s_["test_seeds"] = dict(_SEEDS, demand=65)
    s_["sample_data_params"] = {"split_by_period": True, "train_periods": "(0, 111)", "dev_periods": "(88, 141)",
                                "test_periods": "(118, 171)"}
    s_["params_by_dataset"] = {"train": {"n_samples": 32768, "batch_size": 8192, "periods": 95, "ignore_periods": 16},
                               "dev": {"n_samples": 32768, "batch_size": 32768, "periods": 37, "ignore_periods": 16},
                               "test": {"n_samples": 32768, "batch_size": 32768, "periods": 37, "ignore_periods": 16}}
    return s_


EPOCH_WORKLOADS["one_store_real_yaml"] = (
    _real_one_store_as_shipped, lambda: _hyperparams(data_driven_policy(), lr=0.003),
    "one_store_real_data_lost_demand.yml + data_driven_net.yml as shipped (synthetic stand-in files): 32768 series x 1 store, 4 batches "
    "of 8192 x T=95 (ignore 16), one training epoch through Trainer.do_one_epoch")
EPOCH_WORKLOADS["one_store_real_transformed_nv_yaml"] = (
    _real_one_store_as_shipped, lambda: _hyperparams(transformed_nv_policy(), lr=0.03),
    "one_store_real_data_lost_demand.yml + transformed_nv.yml as shipped (stand-in files, stand-in forecaster weights): 32768 series x "
    "1 store, 4 batches of 8192 x T=95 (ignore 16), one training epoch through Trainer.do_one_epoch")
EPOCH_WORKLOADS["gnn_yaml"] = (
    lambda: _as_shipped(one_warehouse(5)), lambda: _hyperparams(gnn_policy(), lr=0.001),
    "one_warehouse_lost_demand.yml + gnn.yml as shipped: 5 stores, 8192 samples in batches of 1024 x T=50 (ignore 30), one training "
    "epoch through Trainer.do_one_epoch on the fused gather-MLP engine")


def get_epoch(name):
    setting, hyper, desc = EPOCH_WORKLOADS[name]
    return copy.deepcopy(setting()), copy.deepcopy(hyper()), desc


def get(name):
    build, policy, n, T, desc = WORKLOADS[name]
    return copy.deepcopy(build()), copy.deepcopy(policy), n, T, desc
