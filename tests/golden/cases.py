"""Parity cases shared by make_golden.py (reference side) and the tests (oracle / HIP side).

Each case names a reference setting + policy YAML, the overrides applied to them, and the
(small) sizes at which the reference was run.  The five `cfg*` families are BASELINE.json's
five configurations scaled down to sizes the CPU finishes in well under a second.
"""


def _cfg5_3x64():
    """BASELINE cfg5's real topology: the [3][64] adjacency and [64][3] lead-time matrix (lead times 1..6, 0 = not
    connected) that bench.py's `cfg5` workload uses (`workloads.many_warehouses(64, 3)`), in the reference's YAML format."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from neural_inventory_control_amd import workloads
    s = workloads.many_warehouses(64, 3)
    return s["problem_params"]["warehouse_store_adjacency"], s["store_params"]["lead_time"]["value"]


_ADJ64, _LEAD64 = _cfg5_3x64()

CASES = {
    # cfg1: one_store_lost + vanilla_one_store (Poisson demand, lost sales)
    "cfg1_one_store_lost_vanilla": dict(
        setting="one_store_lost", policy="vanilla_one_store", n=48, periods=14, ignore=5, torch_seed=11),
    # cfg2: one_store_backlogged + vanilla_one_store / closed-form policies
    "cfg2_one_store_backlogged_vanilla": dict(
        setting="one_store_backlogged", policy="vanilla_one_store", n=40, periods=16, ignore=6, torch_seed=12),
    "cfg2_one_store_backlogged_base_stock": dict(
        setting="one_store_backlogged", policy="base_stock", n=40, periods=16, ignore=6, torch_seed=13),
    "cfg2_one_store_backlogged_capped": dict(
        setting="one_store_backlogged", policy="capped_base_stock", n=40, periods=16, ignore=6, torch_seed=14),
    # cfg3: one_warehouse_lost_demand with 16 stores + vanilla_warehouse (hidden width cut to keep the fixture small)
    "cfg3_one_warehouse_16_vanilla": dict(
        setting="one_warehouse_lost_demand", policy="vanilla_warehouse", n=24, periods=12, ignore=4, torch_seed=15,
        problem_overrides={"n_stores": 16}, hidden=[64, 64, 64]),
    "cfg3_one_warehouse_5_vanilla": dict(
        setting="one_warehouse_lost_demand", policy="vanilla_warehouse", n=20, periods=10, ignore=3, torch_seed=16,
        hidden=[48, 48]),
    # cfg4: serial_system (store + warehouse + 2 echelons) + vanilla_serial / echelon_stock
    "cfg4_serial_vanilla": dict(
        setting="serial_system", policy="vanilla_serial", n=32, periods=15, ignore=5, torch_seed=17),
    "cfg4_serial_echelon_stock": dict(
        setting="serial_system", policy="echelon_stock", n=32, periods=15, ignore=5, torch_seed=18),
    # cfg5: many_warehouses_lost_demand (shipped 2x10 adjacency) + vanilla_warehouse
    "cfg5_many_warehouses_2x10_vanilla": dict(
        setting="many_warehouses_lost_demand", policy="vanilla_warehouse", n=20, periods=12, ignore=4, torch_seed=19,
        hidden=[64, 64]),
    # cfg5 scaled: 3 warehouses x 8 stores, synthetic adjacency / lead-time matrix in the YAML's format
    "cfg5_many_warehouses_3x8_vanilla": dict(
        setting="many_warehouses_lost_demand", policy="vanilla_warehouse", n=16, periods=10, ignore=3, torch_seed=20,
        hidden=[32, 32],
        problem_overrides={
            "n_stores": 8, "n_warehouses": 3,
            "warehouse_store_adjacency": [[1, 1, 0, 0, 1, 0, 1, 0],
                                          [0, 1, 1, 1, 0, 0, 1, 1],
                                          [1, 0, 0, 1, 0, 1, 0, 1]]},
        store_overrides={"lead_time": {"sample_across_stores": False, "vary_across_samples": False, "expand": True,
                                       "value": [[2, 0, 3], [1, 4, 0], [0, 2, 0], [0, 6, 1],
                                                 [3, 0, 0], [0, 0, 2], [5, 2, 0], [0, 1, 3]]}},
        warehouse_overrides={"holding_cost": [0.3, 0.4, 0.2], "lead_time": 3, "edge_cost": [0.5, 1.5, 0.7]}),
    # cfg5 at its REAL topology: 3 warehouses x 64 stores (16 stores per lane of the env / head kernels' quad walk, Ws = 6,
    # the 3-warehouse shipment exchange and the 195-row logits layer), small hidden width and batch
    "cfg5_many_warehouses_3x64_vanilla": dict(
        setting="many_warehouses_lost_demand", policy="vanilla_warehouse", n=16, periods=10, ignore=3, torch_seed=24,
        hidden=[64, 64],
        problem_overrides={"n_stores": 64, "n_warehouses": 3, "warehouse_store_adjacency": _ADJ64},
        store_overrides={"lead_time": {"sample_across_stores": False, "vary_across_samples": False, "expand": True,
                                       "value": _LEAD64}},
        warehouse_overrides={"holding_cost": [0.3, 0.4, 0.2], "lead_time": 3, "edge_cost": [0.5, 1.5, 0.7]}),
    # the reference's transshipment setting (backlogged demand, warehouse that cannot hold stock: the softmax head has no
    # 'keep' column) + its vanilla_transshipment policy file (VanillaWarehouse with transshipment: True)
    "x_transshipment_backlogged_vanilla": dict(
        setting="transshipment_backlogged", policy="vanilla_transshipment", n=24, periods=12, ignore=4, torch_seed=21,
        hidden=[32, 32]),
    # the GNN policy (gnn.yml) on the one-warehouse setting, 3 stores (shipped) and 16 stores (cfg3's graph)
    "f1_one_warehouse_gnn": dict(
        setting="one_warehouse_lost_demand", policy="gnn", n=16, periods=8, ignore=3, torch_seed=22),
    "f1_one_warehouse_16_gnn": dict(
        setting="one_warehouse_lost_demand", policy="gnn", n=12, periods=6, ignore=2, torch_seed=23,
        problem_overrides={"n_stores": 16}),
    # gnn_transshipment.yml (GNN with transshipment: True - no self loop at the warehouse, allocation ratio not capped at 1)
    "f1_one_warehouse_gnn_transshipment": dict(
        setting="one_warehouse_lost_demand", policy="gnn_transshipment", n=12, periods=6, ignore=2, torch_seed=42),
    # the GNN on the many-warehouse graph.  Upstream writes store s's j-th incoming edge into action column j (not into the
    # column of the warehouse the edge comes from, neural_networks.py:1423-1428) and only runs when some store is connected to
    # every warehouse (otherwise the action tensor is narrower than the lead-time matrix and the env step raises): the shipped
    # 2 x 10 adjacency has stores with one warehouse (0, 1, 5, 9 -> warehouse 1 only: their order lands in warehouse 0's column,
    # whose lead time is 0, and the env step's flat-index put (environment.py:422-432) then adds it to the element BEFORE the
    # store's pipeline - the last slot of the previous store, for store 0 of the previous SCENARIO).  This fixture pins that
    # upstream behaviour for the oracle; the HIP path drops such orders instead
    "f1_many_warehouses_2x10_gnn": dict(
        setting="many_warehouses_lost_demand", policy="gnn", n=10, periods=14, ignore=4, torch_seed=51),
    # ... and a graph upstream handles consistently: 3 warehouses x 8 stores, every store connected to every warehouse (column j
    # IS warehouse j, no order ever meets a lead time of 0), heterogeneous lead times, edge costs in the warehouse features
    "f1_many_warehouses_3x8_dense_gnn": dict(
        setting="many_warehouses_lost_demand", policy="gnn", n=8, periods=14, ignore=4, torch_seed=52,
        problem_overrides={
            "n_stores": 8, "n_warehouses": 3,
            "warehouse_store_adjacency": [[1] * 8, [1] * 8, [1] * 8]},
        store_overrides={"lead_time": {"sample_across_stores": False, "vary_across_samples": False, "expand": True,
                                       "value": [[2, 5, 3], [1, 4, 2], [3, 2, 6], [4, 6, 1],
                                                 [3, 1, 1], [2, 3, 2], [5, 2, 1], [4, 1, 3]]}},
        warehouse_overrides={"holding_cost": [0.3, 0.4, 0.2], "lead_time": 3, "edge_cost": [0.5, 1.5, 0.7]}),
    # the real-data path.  Favorita weekly sales (288 products x 21 stores x 171 weeks, shipped with the reference),
    # 3 warehouses, profit objective, past-demand window (16) + days-from-christmas in the observation, period_shift 16,
    # datasets split BY PERIOD; data_driven_net = MLP over all features + proportional allocation of warehouse stock
    "f4_real_many_warehouses_data_driven": dict(
        setting="many_warehouses_real_data_lost_demand", policy="data_driven_net", n=12, periods=10, ignore=3, torch_seed=31,
        hidden=[32, 32], real=True, period_range="(0, 48)"),
    "f4_real_many_warehouses_just_in_time": dict(
        setting="many_warehouses_real_data_lost_demand", policy="just_in_time", n=12, periods=10, ignore=3, torch_seed=32,
        real=True, period_range="(0, 48)"),
    # the quantile (generalised newsvendor) policies on the one-store real-data setting.  Its data blob is not shipped, so the
    # generator derives one from the shipped 21-store file: every (product, store) series becomes a one-store sample
    "f4_real_one_store_transformed_nv": dict(
        setting="one_store_real_data_lost_demand", policy="transformed_nv", n=40, periods=12, ignore=4, torch_seed=33,
        real=True, period_range="(0, 48)", one_store_from_21=True),
    "f4_real_one_store_fixed_quantile": dict(
        setting="one_store_real_data_lost_demand", policy="fixed_quantile", n=40, periods=12, ignore=4, torch_seed=34,
        real=True, period_range="(0, 48)", one_store_from_21=True),
    "f4_real_one_store_quantile_nv": dict(
        setting="one_store_real_data_lost_demand", policy="quantile_nv", n=40, periods=12, ignore=4, torch_seed=35,
        real=True, period_range="(0, 48)", one_store_from_21=True),
    "f4_real_one_store_returns_nv": dict(
        setting="one_store_real_data_lost_demand", policy="returns_nv", n=40, periods=12, ignore=4, torch_seed=36,
        real=True, period_range="(0, 48)", one_store_from_21=True),
    "f4_real_one_store_data_driven": dict(
        setting="one_store_real_data_lost_demand", policy="data_driven_net", n=40, periods=12, ignore=4, torch_seed=43,
        hidden=[32, 32], real=True, period_range="(0, 48)", one_store_from_21=True),
    # the reference's "read file" example: per-sample lead times (4..6), holding and underage costs from tensors on disk, a per-sample
    # feature column (store number) from a CSV, profit objective, data-driven net
    "f4_real_one_store_read_files_data_driven": dict(
        setting="one_store_real_data_read_file_example", policy="data_driven_net", n=40, periods=12, ignore=4, torch_seed=44,
        hidden=[32, 32], real=True, period_range="(0, 48)", one_store_from_21=True),
    "f4_real_one_store_just_in_time": dict(
        setting="one_store_real_data_lost_demand", policy="just_in_time", n=40, periods=12, ignore=4, torch_seed=37,
        real=True, period_range="(0, 48)", one_store_from_21=True),
}


def apply_overrides(case, config_setting, config_hyper):
    """Returns deep-copied (setting, hyper) dicts with the case's overrides applied."""
    import copy
    cs, ch = copy.deepcopy(config_setting), copy.deepcopy(config_hyper)
    cs["problem_params"].update(case.get("problem_overrides", {}))
    cs["store_params"].update(case.get("store_overrides", {}))
    if case.get("warehouse_overrides"):
        cs["warehouse_params"] = dict(case["warehouse_overrides"])
    if case.get("hidden") is not None:
        ch["nn_params"]["neurons_per_hidden_layer"]["master"] = list(case["hidden"])
    return cs, ch


# "Slim" fixtures: batch sizes at which storing inputs and per-period traces would take megabytes.  The fixture keeps the config,
# the weights, rewards (T, B), totals and gradients, plus float64 checksums of every input tensor; the tests regenerate the inputs
# from the seeds with this repository's `Scenario` host path (seed-for-seed the reference's, pinned by the other fixtures) and
# check the checksums before comparing anything.
SLIM_CASES = {
    # round 4: the reference's SHIPPED training batch (one_warehouse_lost_demand.yml: 5 stores, batch_size 1024, periods 50,
    # ignore 30) - the small-batch route of the engine (64 x 128 / 32 x 128 GEMM tiles, weight gradients split over period
    # groups); hidden width 256 keeps weights + gradients at 1 MB while still taking the 256 x 256 weight-gradient tiles
    "cfg3_shipped_batch1024_vanilla": dict(
        setting="one_warehouse_lost_demand", policy="vanilla_warehouse", n=1024, periods=50, ignore=30, torch_seed=61,
        hidden=[256, 256, 256], slim=True),
}
