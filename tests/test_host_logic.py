"""Host-side logic of the product package that needs no GPU: scenario generation (bit-equal to the reference's golden
data), dataset plumbing, the policy factory, layout helpers, and the no-CPU-fallback contract."""
import copy
import ctypes
import re
import os

import numpy as np
import pytest
import torch

from golden_io import Golden, case_names
from neural_inventory_control_amd import _lib, layout
from neural_inventory_control_amd.data_handling import DatasetCreator, DeviceBatches, MyDataset, Scenario, Scenarios
from neural_inventory_control_amd.environment import Simulator
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator, VanillaWarehouse
from neural_inventory_control_amd.rollout import FusedRollout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scenario(g):
    c = g.fresh_config()
    sc = Scenario(c["periods"], c["problem_params"], c["store_params"], c["warehouse_params"], c["echelon_params"],
                  c["n"], c["observation_params"], c["seeds"])
    return c, sc


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("f4_real")])
def test_scenario_matches_reference_data(name):
    g = Golden(name)
    c, sc = _scenario(g)
    data, ref = sc.get_data(), g.data
    assert set(data) == set(ref)
    for k in ref:
        assert data[k].dtype == torch.float32 and torch.equal(data[k], ref[k]), k
    assert c["seeds"]["demand"] == int(g.z["mutated_demand_seed"])
    np.testing.assert_array_equal(np.asarray(c["store_params"]["demand"]["mean"], dtype=np.float64), g.z["mutated_mean"])
    assert Scenarios is Scenario


def test_dataset_split_and_errors():
    g = Golden("cfg3_one_warehouse_5_vanilla")
    c, sc = _scenario(g)
    train, dev = DatasetCreator().create_datasets(sc, split=True, by_sample_indexes=True, sample_index_for_split=6)
    assert len(dev) == 6 and len(train) == c["n"] - 6
    assert torch.equal(dev[2]["demands"], sc.get_data()["demands"][2])
    assert torch.equal(train[0]["demands"], sc.get_data()["demands"][6])
    whole = DatasetCreator().create_datasets(sc, split=False)
    assert isinstance(whole, MyDataset) and len(whole) == c["n"]
    with pytest.raises(NotImplementedError):
        DatasetCreator().create_datasets(sc, split=True)
    # split by period keeps sample-indexed tensors and slices period-indexed ones (data_handling.py:431-448)
    sc.split_by["period"].append("demands")
    parts = DatasetCreator().split_by_period(sc, ["(0, 4)", "(4, 10)"])
    assert parts[0]["demands"].shape[2] == 4 and parts[1]["demands"].shape[2] == 6
    bad = g.fresh_config()
    bad["warehouse_params"]["holding_cost"] = [0.1, 0.2]
    with pytest.raises(ValueError):
        Scenario(5, bad["problem_params"], bad["store_params"], bad["warehouse_params"], None, 4,
                 bad["observation_params"], bad["seeds"])


def test_device_batches_cover_dataset_and_shard():
    g = Golden("cfg1_one_store_lost_vanilla")
    c, sc = _scenario(g)
    ds = DatasetCreator().create_datasets(sc, split=False)
    full = sc.get_data()["demands"]
    seen = torch.cat([b["demands"] for b in DeviceBatches(ds, 20, shuffle=False, device="cpu")])
    assert torch.equal(seen, full)
    # two ranks: each global batch is split in contiguous halves; together they cover it exactly once
    r0 = list(DeviceBatches(ds, 20, shuffle=True, device="cpu", seed=3, rank=0, world_size=2))
    r1 = list(DeviceBatches(ds, 20, shuffle=True, device="cpu", seed=3, rank=1, world_size=2))
    both = torch.cat([torch.cat([a["demands"], b["demands"]]) for a, b in zip(r0, r1)])
    assert sorted(both.sum(dim=(1, 2)).tolist()) == sorted(full.sum(dim=(1, 2)).tolist())


def test_policy_factory_and_state_dict_layout():
    g = Golden("cfg3_one_warehouse_5_vanilla")
    c, sc = _scenario(g)
    model = NeuralNetworkCreator().create_neural_network(sc, c["nn_params"], device="cpu")
    assert isinstance(model, VanillaWarehouse) and model.trainable
    assert float(model.warehouse_upper_bound) == float(g.z["warehouse_upper_bound"][0])
    assert FusedRollout.supports(model)
    with pytest.raises(KeyError):
        NeuralNetworkCreator().get_architecture("no_such_policy")   # unknown names raise KeyError as upstream (:1519-1536)
    # ("symmetry_aware" is not registered upstream - SURVEY facts; here it names this repository's recovered policy, round 3)
    assert NeuralNetworkCreator().get_architecture("symmetry_aware").__name__ == "SymmetryAware"
    # materialise like the engine does and load the reference's weights by key
    eng = FusedRollout.__new__(FusedRollout)
    eng.model = model
    F = 5 * 3 + 3
    eng.materialize(F)
    assert set(model.state_dict().keys()) == set(g.params.keys())
    model.load_state_dict(g.params)
    assert NeuralNetworkCreator().set_default_output_size("master", {"n_stores": 64, "n_warehouses": 3}) == 195


def test_layout_round_trip_and_uniform_tables():
    t = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    s = layout.to_soa(t)
    assert s.shape == (3, 4, 64) and torch.equal(layout.ref_view(s, 2), t)
    assert float(s[..., 2:].abs().sum()) == 0
    uni = torch.tensor([1.0, 2.0, 3.0]).expand(5, 3)
    tab = layout.Table.from_ref(uni, 64)
    assert tab.scn_stride == 0 and tab.loc_stride == 1
    tab2 = layout.Table.from_ref(uni.contiguous(), 64)  # materialised by collate, still recognised as uniform
    assert tab2.scn_stride == 0
    var = torch.rand(5, 3)
    tab3 = layout.Table.from_ref(var, 64)
    assert tab3.scn_stride == 1 and tab3.loc_stride == 64
    with pytest.raises(ValueError):
        layout.EnvProblem({"n_stores": 1, "n_warehouses": 0, "n_extra_echelons": 0, "lost_demand": True,
                           "maximize_profit": False},
                          {"initial_inventories": torch.zeros(4, 1, 1), "underage_costs": torch.ones(4, 1),
                           "holding_costs": torch.ones(4, 1), "lead_times": torch.ones(4, 1, 1)}, "cpu")


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads on a machine without a GPU and exports exactly what include/nic_rollout.h declares."""
    header = open(os.path.join(ROOT, "include", "nic_rollout.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(nic_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.PROTOTYPES.keys()), declared ^ set(_lib.PROTOTYPES.keys())
    lib = _lib.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nic_abi_version() == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour on a machine WITHOUT a GPU")
def test_no_cpu_fallback():
    g = Golden("cfg1_one_store_lost_vanilla")
    c = g.fresh_config()
    with pytest.raises(_lib.NicUnavailableError):
        Simulator(device="cpu").reset(c["periods"], c["problem_params"], g.data, c["observation_params"])
    from neural_inventory_control_amd import ops
    with pytest.raises(_lib.NicUnavailableError):
        ops.linear_fwd(torch.zeros(4, 4), None, torch.zeros(4, 64), torch.zeros(4, 64), 4, 0)
    with pytest.raises(_lib.NicUnavailableError):
        _lib.load_library("/nonexistent/libnic_hip.so")


def test_main_run_cli_surface():
    """The driver keeps the reference's positional arguments (main_run.py:7-19) and rejects real-data settings it cannot
    serve; nothing here touches the GPU."""
    from neural_inventory_control_amd import main_run
    with pytest.raises(SystemExit):
        main_run.main(["deploy"])
    with pytest.raises(FileNotFoundError):
        main_run.main(["train", "no_such_setting", "no_such_policy", "--config-dir", "/nonexistent"])
    assert main_run.SETTING_KEYS[0] == "seeds" and "nn_params" in main_run.HYPERPARAM_KEYS


@pytest.mark.parametrize("name", ["f1_one_warehouse_gnn", "f1_one_warehouse_gnn_transshipment", "f1_many_warehouses_2x10_gnn",
                                  "f1_many_warehouses_3x8_dense_gnn"])
def test_gnn_engine_graph_plan_equals_the_oracles_graph(name):
    """`GraphPlan` (the GNN engine's compiled supply graph) against the oracle's restatement of neural_networks.py:757-1062 on
    the fixtures' settings: edge list and order, degrees behind the 1/sqrt normalisation, allocation groups, and the
    edge -> action-column mapping - including upstream's "j-th connected warehouse" columns (:1423-1428), which the
    many-warehouse fixtures exercise (stores that see one warehouse of two)."""
    import oracle.inventory_oracle as orc
    from neural_inventory_control_amd.gnn_rollout import GnnRollout, GraphPlan
    g = Golden(name)
    c = g.fresh_config()
    prob = c["problem_params"]
    S, Wn = prob["n_stores"], prob["n_warehouses"]
    trans = bool(c["nn_params"].get("transshipment", False))
    obs = {"lead_times": g.data["lead_times"], "warehouse_lead_times": g.data["warehouse_lead_times"]}
    og = orc.gnn_graph(prob, obs, trans)
    conn = prob["warehouse_store_adjacency"] if Wn > 1 else [[1] * S]
    P = GraphPlan(S, conn, trans, "cpu")
    n_int = len(og["internal"])
    assert P.n_int == n_int and P.n_nodes == og["n_nodes"]
    assert P.src[:n_int].tolist() == [a for a, _ in og["internal"]] and P.tgt[:n_int].tolist() == [b for _, b in og["internal"]]
    assert P.n_self == len(og["supplying"]) and P.n_edges == n_int + Wn + S + len(og["supplying"])
    torch.testing.assert_close(P.in_scale, torch.tensor([1.0 / d ** 0.5 for d in og["in_deg"]]))
    torch.testing.assert_close(P.out_scale, torch.tensor([1.0 / d ** 0.5 for d in og["out_deg"]]))
    # action columns: row s * Wn + j of the orders buffer <-> mapping["stores"][s][j]; warehouse orders <-> supplier edges
    rows = P.order_row.tolist()
    for s_, edges in enumerate(og["mapping"]["stores"]):
        for j, e in enumerate(edges):
            assert rows[e] == s_ * Wn + j
    for w, (e,) in enumerate(og["mapping"]["warehouses"]):
        assert rows[e] == S * Wn + w
    assert all(r == -1 for r in rows[n_int + Wn:])
    # allocation groups = every supplying node's outgoing internal edges + its self loop.  (The engine numbers the self loops
    # BEFORE the demand edges - the reference after them - so that the edges whose output matters are the first `n_live`.)
    n_self = len(og["supplying"])
    assert P.n_live == n_int + Wn + n_self and P.e_demand == P.n_live
    for w, (first, count, e_self, e_sup) in enumerate(P.groups.tolist()):
        assert list(range(first, first + count)) == [i for i, (a, _) in enumerate(og["internal"]) if a == w]
        assert e_sup == n_int + w
        assert e_self == (n_int + Wn + og["supplying"].index(w) if w in og["supplying"] else -1)
    assert P.src[P.e_demand:].tolist() == list(range(Wn, Wn + S)) and P.tgt[P.e_demand:].tolist() == [-1] * S
    # a node's aggregation lists name its edges in the REFERENCE's order (internal, supplier, demand, self loop)
    r2e = list(range(n_int + Wn)) + [P.e_demand + s_ for s_ in range(S)] + [n_int + Wn + k for k in range(n_self)]
    off, items = P.agg_off.tolist(), P.agg_items.tolist()
    n_e = n_int + Wn + S + n_self
    r_src = [a for a, _ in og["internal"]] + [-1] * Wn + list(range(Wn, Wn + S)) + og["supplying"]
    r_tgt = [b for _, b in og["internal"]] + list(range(Wn)) + [-1] * S + og["supplying"]
    for n in range(P.n_nodes):
        want_in = [r2e[r] for r in range(n_e) if r_tgt[r] == n and not (n_int + Wn <= r < n_int + Wn + S)]
        want_out = [r2e[r] for r in range(n_e) if r_src[r] == n and not (n_int <= r < n_int + Wn)]
        assert items[off[n]:off[n + 1]] == want_in and items[off[P.n_nodes + n]:off[P.n_nodes + n + 1]] == want_out
    # upstream's column quirk is present exactly where a store's j-th connected warehouse is not warehouse j
    expect = [(b - Wn, j, a) for s_ in range(S)
              for j, (a, b) in enumerate([(a, b) for (a, b) in og["internal"] if b - Wn == s_]) if j != a]
    assert sorted(P.misplaced) == sorted(expect) == sorted(og["misplaced"])
    if name == "f1_many_warehouses_2x10_gnn":
        assert [m[0] for m in P.misplaced] == [0, 1, 5, 9]      # the stores served by warehouse 1 only
    model_ok = type("GNN", (), {"nn_args": c["nn_params"]})()
    assert GnnRollout.supports(model_ok, prob)
    narrow = dict(prob, n_warehouses=2, n_stores=2, warehouse_store_adjacency=[[1, 0], [0, 1]])
    assert not GnnRollout.supports(model_ok, narrow)            # no store sees both warehouses: upstream itself raises there


def test_mlp_engine_routing_rules_for_the_data_driven_policy():
    """Host-side rules only (no device): `FusedRollout.supports` takes `data_driven` with its ELU / ReLU activations,
    `observation_ok` wants exactly the observation that policy reads (past-demand window + days_from_christmas; sample features
    may ride along, DataDrivenNet.forward ignores them) and keeps the vanilla policies on plain observations; `input_rows` is
    the reference's concatenated feature count (neural_networks.py:452-470)."""
    g = Golden("f4_real_many_warehouses_data_driven")
    c = g.fresh_config()

    class _Sc:
        problem_params = c["problem_params"]
        store_params = {"demand": {}}
    model = NeuralNetworkCreator().create_neural_network(_Sc(), c["nn_params"], device="cpu")
    assert FusedRollout.supports(model)
    obs = c["observation_params"]
    assert FusedRollout.observation_ok(model, obs, g.data)
    assert FusedRollout.observation_ok(model, dict(obs, sample_features=["store_nbr"]), g.data)
    assert not FusedRollout.observation_ok(model, dict(obs, time_features=None), g.data)
    assert not FusedRollout.observation_ok(model, dict(obs, demand={"past_periods": 0, "period_shift": 0}), g.data)
    relu_inside = copy.deepcopy(c["nn_params"])
    relu_inside["inner_layer_activations"]["master"] = "relu"
    assert not FusedRollout.supports(NeuralNetworkCreator().create_neural_network(_Sc(), relu_inside, device="cpu"))
    eng = FusedRollout.__new__(FusedRollout)
    eng.head = "data_driven"
    assert eng.input_rows(g.data, obs) == g.params["net.master.0.weight"].shape[1]
    gv = Golden("cfg3_one_warehouse_5_vanilla")
    cv, scv = _scenario(gv)
    vanilla = NeuralNetworkCreator().create_neural_network(scv, cv["nn_params"], device="cpu")
    assert FusedRollout.observation_ok(vanilla, cv["observation_params"]) and not FusedRollout.observation_ok(vanilla, obs)


def test_factory_builds_gnn_policy_with_reference_state_dict_keys():
    """`gnn` in the factory registry (neural_networks.py:1519-1536): five named MLPs, bias 5.0 on the output layer, and the
    reference's state-dict key layout (checkpoints interchange)."""
    from neural_inventory_control_amd import workloads
    from neural_inventory_control_amd.neural_networks import GNN, NeuralNetworkCreator
    setting, policy, _, _, _ = workloads.get("gnn")

    class _Sc:
        problem_params = setting["problem_params"]
        store_params = {"demand": {"mean": [5.0] * 16}}
    model = NeuralNetworkCreator().create_neural_network(_Sc(), policy, device="cpu")
    assert isinstance(model, GNN) and model.gradient_clipping_norm_value == 1.0
    assert list(model.net.keys()) == ["initial_node", "initial_edge", "node_update", "edge_update", "output"]
    out_last = [m for m in model.net["output"] if hasattr(m, "bias")][-1]
    assert float(out_last.bias.detach()[0]) == 5.0
    keys = [k for k, _ in model.named_parameters() if "UninitializedParameter" not in k]
    assert "net.output.4.bias" in keys and "net.initial_node.0.weight" in keys


REAL_CASES = [n for n in case_names() if n.startswith("f4_real")]


@pytest.mark.parametrize("name", REAL_CASES)
def test_real_data_scenario_and_period_split_equal_reference(name):
    """`Scenario(periods=None)` on the shipped Favorita files + `DatasetCreator.split_by_period` reproduce the
    reference's dataset dict (fixture `data/*`) bit for bit.  Needs the reference's data files (this container)."""
    import os
    import tempfile
    g = Golden(name)
    c = g.fresh_config()
    ref_root = "/root/reference"
    if not os.path.isdir(os.path.join(ref_root, "data_files")):
        pytest.skip("the reference's data files are not on this machine")
    sp = c["store_params"]
    cwd = os.getcwd()
    os.chdir(ref_root)
    try:
        if sp["demand"]["file_location"].startswith("<derived"):
            src = torch.load("data_files/favorita_21_stores/weekly_sales.pt", map_location="cpu")
            loc = os.path.join(tempfile.mkdtemp(), "one.pt")
            torch.save(src.reshape(-1, 1, src.shape[2]).clone(), loc)
            sp["demand"]["file_location"] = loc
        sc = Scenario(None, c["problem_params"], sp, c["warehouse_params"], c["echelon_params"], c["n"],
                      c["observation_params"], c["seeds"])
        (ds,) = DatasetCreator().create_datasets(sc, split=True, by_period=True, periods_for_split=[c["period_range"]])
    finally:
        os.chdir(cwd)
    ref = g.data
    assert set(ds.data) == set(ref)
    for k, v in ref.items():
        assert ds.data[k].dtype == torch.float32 and torch.equal(ds.data[k], v), k
    assert sc.split_by["period"] == ["demands"] + list(c["observation_params"]["time_features"])


def test_sharded_host_generation_with_per_sample_tables_equals_single_process(tmp_path):
    """Shards of a job whose tables vary ACROSS SAMPLES (drawn per scenario, or read from a per-sample file): rows
    [lo, lo + n) of every tensor equal the single-process dataset - per-sample costs, per-sample lead times (whose GLOBAL
    maximum sets the number of pipeline slots, data_handling.py:302) and file-backed tables."""
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    n_total, T = 23, 6
    path = str(tmp_path / "holding.pt")
    torch.save(torch.rand(40, 1, generator=torch.Generator().manual_seed(5)) + 0.5, path)

    def build(n, lo, total):
        s, _, _, _, _ = workloads.get("cfg2")
        s["store_params"]["underage_cost"] = {"sample_across_stores": False, "vary_across_samples": True, "expand": False,
                                              "range": [5.0, 12.0]}
        # the largest lead time (8) is not drawn for the first 9 scenarios: a shard that looked at its own rows only would
        # build fewer pipeline slots (and draw its multipliers from a different place of the stream)
        s["store_params"]["lead_time"] = {"sample_across_stores": False, "vary_across_samples": True, "expand": False,
                                          "range": [2, 9]}
        s["store_params"]["holding_cost"] = {"file_location": path}
        obs = defaultdict(lambda: None, s["observation_params"])
        return Scenario(T, s["problem_params"], s["store_params"], None, None, n, obs, s["seeds"], scenario_offset=lo,
                        num_total=total).get_data()

    whole = build(n_total, 0, n_total)
    assert whole["initial_inventories"].shape[2] == int(whole["lead_times"].max())
    cuts = [0, 9, 17, n_total]
    parts = [build(b - a, a, n_total) for a, b in zip(cuts[:-1], cuts[1:])]
    assert len({int(p["lead_times"].max()) for p in parts}) > 1   # the shards do see different local maxima
    for k, v in whole.items():
        assert torch.equal(torch.cat([p[k] for p in parts], dim=0), v), k


def test_symmetry_aware_policy_is_registered_and_its_oracle_allocations_are_feasible():
    """BASELINE cfg3's "symmetry-aware policy net" (not in the reference's source): the factory builds it with the
    three shared nets, and the CPU restatement allocates no more than the warehouse has on hand."""
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from oracle import inventory_oracle as orc
    setting, policy, _, _, _ = workloads.get("cfg3_symmetry_aware")
    obs = defaultdict(lambda: None, setting["observation_params"])
    data = orc.generate_scenario_data(6, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                                      setting["echelon_params"], 9, obs, setting["seeds"])

    class _Sc:
        problem_params = setting["problem_params"]
        store_params = setting["store_params"]
    model = NeuralNetworkCreator().create_neural_network(_Sc(), policy, device="cpu")
    assert type(model).__name__ == "SymmetryAware" and set(model.net.keys()) == {"context", "store", "warehouse"}
    assert float(model.warehouse_upper_bound) == pytest.approx(4 * float(np.sum(setting["store_params"]["demand"]["mean"])), rel=1e-6)
    # oracle side: random weights of the right shapes (the product module's lazy layers need the device to materialise)
    gen = torch.Generator().manual_seed(0)
    F_ctx = 16 * 3 + 3
    dims = {"context": [F_ctx, 256, 64], "store": [3 + 4 + 64, 32, 32, 1], "warehouse": [3 + 64, 32, 32, 1]}
    sd = {}
    for m, ds in dims.items():
        for li, (k, n) in enumerate(zip(ds[:-1], ds[1:])):
            sd[f"net.{m}.{2 * li}.weight"] = torch.randn(n, k, generator=gen) * 0.2
            sd[f"net.{m}.{2 * li}.bias"] = torch.randn(n, generator=gen) * 0.1
    pol = orc.policy_from_state_dict(policy, sd, setting["problem_params"], model.warehouse_upper_bound)
    env = orc.env_reset(6, setting["problem_params"], dict(data), obs)
    act = orc.policy_act(pol, env.obs)
    assert act["stores"].shape == (9, 16, 1) and act["warehouses"].shape == (9, 1, 1)
    assert bool((act["stores"] >= 0).all())
    assert bool((act["stores"].sum(dim=(1, 2)) <= env.obs["warehouse_inventories"][:, 0, 0] + 1e-4).all())
    res, _, grads = orc.train_step_gradients(pol, 6, setting["problem_params"], data, obs)
    assert torch.isfinite(res.total) and all(torch.isfinite(g).all() for g in grads)


def test_whole_horizon_routes_refuse_histories_beyond_32_bit_offsets():
    """csrc/horizon_rollout.hip addresses every history with 32-bit element offsets: the host side must route long evaluation
    horizons on big batches to the per-period kernels / the generic loop instead of handing the launcher sizes it refuses."""
    from types import SimpleNamespace
    from neural_inventory_control_amd import horizon_rollout as hz
    real = SimpleNamespace(S=21, Ws=6, Wn=3, Ww=3, nsup=3, ldb=128)
    assert hz.offsets_ok(real, 95, 111, hz.MAX_HIDDEN) and hz.offsets_ok(real, 5000, 5016, hz.MAX_HIDDEN)
    big = SimpleNamespace(S=21, Ws=6, Wn=3, Ww=3, nsup=3, ldb=8192)
    assert hz.offsets_ok(big, 95, 111, hz.MAX_HIDDEN)
    assert not hz.offsets_ok(big, 5000, 5016, hz.MAX_HIDDEN)          # 135 rows x 5,000 x 8,192 elements
    wide = SimpleNamespace(S=64, Ws=2, Wn=0, Ww=0, nsup=1, ldb=8192)
    assert not hz.offsets_ok(wide, 10, 5000, 0)                       # the demand trace alone: 5,002 x 64 x 8,192


def test_small_rollout_reduce_plan_and_argument_checks():
    """nic_small_rollout_reduce (host side only: no GPU): the scratch size it asks for - (column chunks x row groups) x 256 partial
    sums + two floats per cost block -, and the argument errors it reports instead of launching."""
    from neural_inventory_control_amd import small_rollout as sr
    assert sr.small_rollout_reduce_scratch(2048, 2212, 100 * 32768) == 9 * 64 * 256 + 2 * 256 + 4
    assert sr.small_rollout_reduce_scratch(5, 70, 96) == 1 * 5 * 256 + 2 * 1 + 4      # one row per group, one cost block
    assert sr.small_rollout_reduce_scratch(100, 256, 0) == 1 * 50 * 256 + 4            # 64 groups of 2 rows -> 50 non-empty ones
    assert sr.small_rollout_reduce_scratch(0, 0, 4096) == 2 * 4 + 4
    lib = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    for args in ((None, 4, 8, 8, None, None, 0, 0, None, p),          # nothing to reduce
                 (p, 4, 8, 8, None, None, 0, 0, None, p),             # slab without grad
                 (p, 4, 4, 8, p, None, 0, 0, None, p),                # row stride < columns
                 (None, 0, 0, 0, None, p, 6, 0, p, p),                # costs not a multiple of 4 floats
                 (p, 4, 8, 8, p, None, 0, 0, None, None)):            # no scratch
        assert lib.nic_small_rollout_reduce(*args, None) != 0
        assert b"nic_small_rollout_reduce" in lib.nic_last_error()


def test_load_model_puts_step_counters_where_this_optimizers_implementation_wants_them(tmp_path):
    """`Trainer.load_model` (trainer.py:300-312) into an optimizer whose implementation keeps `step` on the parameter's device
    (fused / capturable Adam - what main_run builds on a GPU) and into a default one: the checkpoint holds host floats, torch
    decides from the SAVED groups' flags where a counter goes, so the loader must hand it this optimizer's own flags.  Checked
    here with parameters on the `meta` device (no GPU needed to see where a counter lands)."""
    from neural_inventory_control_amd.trainer import Trainer, _portable_optimizer_state
    torch.manual_seed(0)
    src = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(src.parameters(), lr=1e-2)
    src(torch.ones(4, 3)).sum().backward()
    opt.step()
    opt.step()
    path = tmp_path / "ck.pt"
    torch.save({"model_state_dict": src.state_dict(), "optimizer_state_dict": _portable_optimizer_state(opt.state_dict()),
                "all_train_losses": [1.0], "all_dev_losses": [2.0], "all_test_losses": [], "warehouse_upper_bound": None}, path)

    class OnMeta(torch.nn.Linear):  # parameters live on `meta`; the weights themselves are not what this test looks at
        def load_state_dict(self, sd, *a, **k):
            return None

    for kw, on_device in (({"capturable": True}, True), ({}, False)):
        dst = OnMeta(3, 2, device="meta")
        opt2 = torch.optim.Adam(dst.parameters(), lr=1e-2, **kw)
        Trainer(device="cpu").load_model(dst, opt2, str(path))
        for g in opt2.param_groups:
            assert bool(g.get("capturable")) == on_device
            for p in g["params"]:
                step = opt2.state[p]["step"]
                assert torch.is_tensor(step) and step.dtype == torch.float32
                assert (step.device.type == "meta") == on_device
                if not on_device:
                    assert float(step) == 2.0


def test_demand_trace_view_is_taken_only_for_full_kernel_layout_rows():
    """layout.demand_trace_soa: a (B, S, T) batch that is a `ref_view` of a [T][S][ldb] trace with B == ldb comes back as that
    storage; ragged rows (padding lanes of unknown content), other strides and other dtypes are transposed into a fresh buffer."""
    import torch
    from neural_inventory_control_amd.layout import demand_trace_soa, ref_view
    for T, S, ld, B in ((5, 3, 64, 64), (5, 1, 128, 128), (1, 1, 64, 64), (5, 3, 64, 50), (4, 2, 64, 1)):
        soa = torch.randn(T, S, ld)
        soa[:, :, B:] = 0
        d = soa[:, :, :B].permute(2, 1, 0)
        v = demand_trace_soa(d, ld)
        assert torch.equal(v, soa) and v.shape == (T, S, ld)
        assert (v.data_ptr() == soa.data_ptr()) == (B == ld and B > 1)
        c = d.clone()   # (what a captured step keeps as its static input: the strides of a dense view survive a clone)
        assert torch.equal(demand_trace_soa(c, ld), soa)
        x = torch.randn(B, S, T)
        out = demand_trace_soa(x, ld)
        assert torch.equal(out[:, :, :B], x.permute(2, 1, 0)) and not out[:, :, B:].any()
        assert out.data_ptr() != x.data_ptr() or (S == 1 and T == 1)   # (one store, one period: the two layouts coincide)
    wide = torch.randn(6, 2, 128)   # a (B = 64)-column window of a 128-wide trace: stride 128, not this batch's ldb
    w = wide[:, :, :64].permute(2, 1, 0)
    assert demand_trace_soa(w, 64).data_ptr() != wide.data_ptr()
    dbl = torch.randn(4, 2, 64, dtype=torch.float64).permute(2, 1, 0)
    assert demand_trace_soa(dbl, 64).dtype == torch.float32


def test_weight_pack_plan_reproduces_every_packer_with_one_gather():
    """ops.WeightPackPlan: the packers run once on stand-in parameters holding their own positions; afterwards ONE concatenation + ONE
    gather refresh all their buffers - same contents as each packer's own `pack()`, also after the parameters change, and the
    buffers are views of one allocation at 256-byte offsets."""
    import torch
    from neural_inventory_control_amd import ops
    torch.manual_seed(3)
    lins = [torch.nn.Linear(37, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)]
    outs = [torch.nn.Linear(45, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 1)]
    mk = lambda: [ops.GnnPeriodBwdPack(lins, 32, 2, "cpu"), ops.GnnPeriodBwdPack(outs, 1, 1, "cpu"),
                  ops.GnnPeriodBwdPack(lins, 32, 1, "cpu")]   # noqa: E731  (the same layers packed twice, differently)
    direct, planned = mk(), mk()
    plan = ops.WeightPackPlan([(p, ["buf"]) for p in planned], "cpu")
    for rnd in range(2):
        for p in direct:
            p.pack()
        plan.pack()
        for a, b in zip(direct, planned):
            assert torch.equal(a.buf, b.buf) and a.buf.abs().sum() > 0
            assert b.buf.untyped_storage().data_ptr() == plan.buf.untyped_storage().data_ptr() and b.buf.storage_offset() % 64 == 0
        with torch.no_grad():
            for lin in lins + outs:
                lin.weight.add_(1.0)
                lin.bias.mul_(2.0)


def test_weight_pack_plan_reads_a_parameter_replaced_on_its_layer():
    """The plan keeps the LAYERS: a `lin.weight = nn.Parameter(...)` after the plan was built is what the next pack reads (the
    packers' own pack() always read the live attribute); a changed shape is refused."""
    import torch
    from neural_inventory_control_amd import ops
    lins = [torch.nn.Linear(20, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)]
    direct, planned = ops.GnnPeriodBwdPack(lins, 32, 1, "cpu"), ops.GnnPeriodBwdPack(lins, 32, 1, "cpu")
    plan = ops.WeightPackPlan([(planned, ["buf"])], "cpu")
    lins[1].weight = torch.nn.Parameter(torch.randn(32, 32))
    direct.pack()
    plan.pack()
    assert torch.equal(direct.buf, planned.buf)
    lins[0].weight = torch.nn.Parameter(torch.randn(32, 21))
    with pytest.raises(RuntimeError):
        plan.pack()
