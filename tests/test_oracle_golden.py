"""Oracle (oracle/inventory_oracle.py) against the committed golden vectors produced by the upstream reference.

Bit-exact: the oracle is an op-for-op PyTorch-CPU restatement, so data tensors, per-period rewards, states and
the scalar totals must be EQUAL; parameter gradients are compared exactly as well (same autograd graph shape).
"""
import numpy as np
import pytest
import torch

from golden_io import Golden, case_names, check_slim_inputs, slim_case_names
from oracle import inventory_oracle as orc


REF_ROOT = "/root/reference"


def _build(g, need_data=True):
    """Config of a fixture and, for synthetic cases, the oracle's own scenario data.  Real-data cases read the
    Favorita files the reference ships: their data is rebuilt only where /root/reference exists (this container)."""
    import os
    c = g.fresh_config()
    if not c.get("real"):
        data = orc.generate_scenario_data(c["periods"], c["problem_params"], c["store_params"], c["warehouse_params"],
                                          c["echelon_params"], c["n"], c["observation_params"], c["seeds"])
        return c, data
    if not need_data:
        return c, None
    if not os.path.isdir(os.path.join(REF_ROOT, "data_files")):
        pytest.skip("the reference's data files are not on this machine")
    sp = c["store_params"]
    loc = sp["demand"]["file_location"]
    cwd = os.getcwd()
    os.chdir(REF_ROOT)  # the YAMLs hold paths relative to the reference root
    try:
        if loc.startswith("<derived"):  # one-store blob derived from the shipped 21-store file (tests/golden/make_golden.py)
            import tempfile
            src = torch.load("data_files/favorita_21_stores/weekly_sales.pt", map_location="cpu")
            loc = os.path.join(tempfile.mkdtemp(), "one.pt")
            torch.save(src.reshape(-1, 1, src.shape[2]).clone(), loc)
            sp["demand"]["file_location"] = loc
        full = orc.generate_scenario_data(None, c["problem_params"], sp, c["warehouse_params"], c["echelon_params"], c["n"],
                                          c["observation_params"], c["seeds"])
    finally:
        os.chdir(cwd)
    (data,) = orc.split_data_by_period(full, [c["period_range"]], c["observation_params"], c["problem_params"])
    return c, data


@pytest.mark.parametrize("name", case_names())
def test_scenario_data_bit_equal(name):
    g = Golden(name)
    c, data = _build(g)
    ref = g.data
    assert set(data.keys()) == set(ref.keys())
    for k in ref:
        assert data[k].dtype == torch.float32
        assert tuple(data[k].shape) == tuple(ref[k].shape), k
        assert torch.equal(data[k], ref[k]), k
    # the reference mutates its config in place; downstream code depends on it
    assert c["seeds"]["demand"] == int(g.z["mutated_demand_seed"])
    if not c.get("real"):
        np.testing.assert_array_equal(np.asarray(c["store_params"]["demand"]["mean"], dtype=np.float64), g.z["mutated_mean"])


@pytest.mark.parametrize("name", case_names())
def test_rollout_and_gradients_bit_equal(name):
    g = Golden(name)
    c, _ = _build(g, need_data=False)
    data = g.data
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"],
                                     warehouse_upper_bound=g.tensor("warehouse_upper_bound"), forecaster_state=g.forecaster)
    if not g.grads:  # non-trainable policies (quantile_nv, returns_nv, just_in_time): forward only
        with torch.no_grad():
            res = orc.rollout(pol, c["periods"], c["problem_params"], data, c["observation_params"], c["ignore"])
        mean_loss = res.total / (c["n"] * c["periods"] * c["problem_params"]["n_stores"])
        grads = []
    else:
        res, mean_loss, grads = orc.train_step_gradients(pol, c["periods"], c["problem_params"], data,
                                                         c["observation_params"], c["ignore"])
    assert torch.equal(res.per_period, g.tensor("rewards"))
    assert float(res.total) == float(g.z["total"])
    assert float(res.reported) == float(g.z["reported"])
    assert float(mean_loss) == float(g.z["mean_loss"])
    for k, v in g.states(c["periods"]).items():
        assert torch.equal(res.final_obs[k], v), k
    ref_grads = g.grads
    if hasattr(pol, "param_keys"):  # multi-module policies list their state-dict keys in parameters() order
        keys = pol.param_keys()
    else:
        keys = sorted(ref_grads.keys(), key=lambda s: (int(s.split(".")[2]), s.split(".")[3] != "weight"))
    assert len(keys) == len(grads)
    for k, mine in zip(keys, grads):
        if c.get("real"):
            # the quantile policies interpolate between forecaster outputs; gradients agree to the last bits, not bitwise
            assert float((mine - ref_grads[k]).norm()) <= 1e-6 * float(ref_grads[k].norm()) + 1e-12, k
            continue
        if hasattr(pol, "param_keys"):
            # GNN: the forward (rewards, states, totals above) is bit-equal; its autograd graph has hundreds of fan-in points
            # whose accumulation order is an implementation detail of how the graph was built — gradients agree to ~1 ulp
            assert float((mine - ref_grads[k]).norm() / (ref_grads[k].norm() + 1e-30)) < 1e-6, k
        else:
            assert torch.equal(mine, ref_grads[k]), (k, (mine - ref_grads[k]).abs().max())


@pytest.mark.parametrize("name", case_names())
def test_states_and_actions_trace(name):
    g = Golden(name)
    c, _ = _build(g, need_data=False)
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"],
                                     warehouse_upper_bound=g.tensor("warehouse_upper_bound"), forecaster_state=g.forecaster)
    env = orc.env_reset(c["periods"], c["problem_params"], g.data, c["observation_params"])
    with torch.no_grad():
        for t in range(c["periods"]):
            for k, v in g.states(t).items():
                assert torch.equal(env.obs[k], v), (t, k)
            for k, v in g.features(t).items():  # past-demand window / time features of period t (real-data settings)
                assert torch.equal(env.obs[k], v), (t, k)
            obs_in = dict(env.obs)
            obs_in["internal_data"] = {"demands": env.demands, "period_shift": env.period_shift}
            a = orc.policy_act(pol, obs_in)
            for k, v in g.actions(t).items():
                assert torch.equal(a[k], v), (t, k)
            orc.env_step(env, a)


def test_checkpoint_known_answer():
    """The reference's shipped checkpoint stores best dev loss 6.854347610473633; SURVEY §4 KAT."""
    g = Golden("checkpoint_kat")
    c = g.fresh_config()
    data = orc.generate_scenario_data(c["scenario_periods"], c["problem_params"], c["store_params"], None, None,
                                      c["scenario_samples"], c["observation_params"], c["seeds"])
    dev = {k: v[:c["dev_samples"]] for k, v in data.items()}
    assert float(dev["demands"].double().sum()) == float(g.z["demand_checksum"])
    assert float(dev["initial_inventories"].double().sum()) == float(g.z["init_inv_checksum"])
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"])
    with torch.no_grad():
        res = orc.rollout(pol, c["periods"], c["problem_params"], dev, c["observation_params"], c["ignore"])
    _, reported = orc.epoch_losses(float(res.total), float(res.reported), c["dev_samples"], c["periods"], c["ignore"], 1)
    assert float(res.reported) == float(g.z["reported"])
    assert abs(reported - 6.854347610473633) < 5e-7
    assert abs(reported - float(g.z["stored_best_dev_loss"])) < 5e-7


def test_semantics_probe_worked_example():
    """inv=[2,4,0,0], d=3, a=5, L=4, lost demand, p=9,h=1 -> cost 9, inv'=[4,0,0,5]."""
    prob = {"n_stores": 1, "n_warehouses": 0, "n_extra_echelons": 0, "lost_demand": True, "maximize_profit": False}
    data = {"demands": torch.tensor([[[3.0]]]), "initial_inventories": torch.tensor([[[2.0, 4.0, 0.0, 0.0]]]),
            "underage_costs": torch.tensor([[9.0]]), "holding_costs": torch.tensor([[1.0]]),
            "lead_times": torch.tensor([[[4.0]]])}
    obsp = {"include_warehouse_inventory": False, "demand": {"past_periods": 0, "period_shift": 0},
            "include_static_features": {"holding_costs": True, "underage_costs": True, "lead_times": True}}
    env = orc.env_reset(1, prob, data, obsp)
    r = orc.env_step(env, {"stores": torch.tensor([[[5.0]]])})
    assert r.tolist() == [9.0]
    assert env.obs["store_inventories"].tolist() == [[[4.0, 0.0, 0.0, 5.0]]]


def test_zero_order_has_no_placement_gradient():
    """environment.py:426-432: orders that are exactly 0 are filtered before the put -> zero gradient through the
    pipeline, while the warehouse outflow sum (environment.py:247) still sees them."""
    pipe = torch.zeros(2, 1, 3)
    orders = torch.tensor([[[0.0]], [[2.0]]], requires_grad=True)
    lead = torch.tensor([[[2.0]], [[2.0]]])
    new = orc.shift_pipeline_and_place(pipe, torch.zeros(2, 1), orders, lead)
    new.sum().backward()
    assert orders.grad.flatten().tolist() == [0.0, 1.0]


def test_zero_lead_orders_upstream_leak_and_the_drop_mode():
    """environment.py:422-432 with a lead time of 0: the flat position is base - 1, so upstream adds the order to the last slot of
    the PREVIOUS location - across scenarios for location 0, wrapping to the end of the batch for scenario 0.  The oracle
    reproduces that by default and discards such orders in its "drop" mode (what the HIP env step does)."""
    pipe = torch.zeros(2, 2, 3)
    orders = torch.tensor([[[5.0], [0.0]], [[7.0], [2.0]]])
    lead = torch.tensor([[[0.0], [2.0]], [[0.0], [2.0]]])
    up = orc.shift_pipeline_and_place(pipe, torch.zeros(2, 2), orders, lead)
    # scenario 0 / store 0's order of 5 wraps to the very last element; scenario 1 / store 0's 7 lands in scenario 0 / store 1
    assert up.tolist() == [[[0, 0, 0], [0, 0, 7]], [[0, 0, 0], [0, 2, 5]]]
    drop = orc.shift_pipeline_and_place(pipe, torch.zeros(2, 2), orders, lead, zero_lead_orders="drop")
    assert drop.tolist() == [[[0, 0, 0], [0, 0, 0]], [[0, 0, 0], [0, 2, 0]]]


def test_many_warehouse_gnn_fixture_pins_the_upstream_column_defect():
    """neural_networks.py:1423-1428 writes store s's j-th CONNECTED edge into action column j; on the shipped 2 x 10 adjacency
    stores 0, 1, 5, 9 (warehouse 1 only) therefore order in warehouse 0's column, where their lead time is 0, and the env step
    leaks those orders into the neighbouring store / scenario.  Upstream mode = the fixture bit for bit (test above, all
    cases): changing scenario 5's initial stock changes scenario 4's cost.  In drop mode the scenarios are independent."""
    g = Golden("f1_many_warehouses_2x10_gnn")
    c = g.fresh_config()
    og = orc.gnn_graph(c["problem_params"], {"lead_times": g.data["lead_times"], "warehouse_lead_times": g.data["warehouse_lead_times"]},
                       False)
    assert [m[0] for m in og["misplaced"]] == [0, 1, 5, 9]
    assert all(float(g.data["lead_times"][0, s_, col]) == 0.0 for s_, col, _ in og["misplaced"])
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"], g.tensor("warehouse_upper_bound"))

    def costs(data, mode):
        with torch.no_grad():
            return orc.rollout(pol, c["periods"], c["problem_params"], data, c["observation_params"], c["ignore"],
                               zero_lead_orders=mode).per_period.sum(dim=0)
    # starve scenario 5 (its stores order more): upstream books part of that on scenario 4's last store, drop mode does not
    poked = dict(g.data)
    poked["initial_inventories"] = g.data["initial_inventories"].clone()
    poked["initial_inventories"][5] = 0.0
    others = [b_ for b_ in range(c["n"]) if b_ != 5]
    assert torch.equal(costs(g.data, "drop")[others], costs(poked, "drop")[others])
    up, up_poked = costs(g.data, "upstream"), costs(poked, "upstream")
    assert float(up[4]) != float(up_poked[4]) and torch.equal(up[:4], up_poked[:4]) and torch.equal(up[6:], up_poked[6:])
    assert abs(float(costs(g.data, "drop").sum()) - float(g.z["total"])) > 1e-2 * float(g.z["total"])
    dense = Golden("f1_many_warehouses_3x8_dense_gnn")
    cd = dense.fresh_config()
    assert orc.gnn_graph(cd["problem_params"], {"lead_times": dense.data["lead_times"],
                                                "warehouse_lead_times": dense.data["warehouse_lead_times"]}, False)["misplaced"] == []


@pytest.mark.parametrize("name", slim_case_names())
def test_slim_fixture_at_the_shipped_batch_size(name):
    """Round 4: the reference's shipped training batch (1,024 scenarios x 50 periods).  The fixture keeps no inputs: the oracle's
    generator rebuilds them from the seeds (checksums pinned), then rewards, totals, final state and gradients are bit-equal."""
    g = Golden(name)
    c, data = _build(g)
    check_slim_inputs(g, data)
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"],
                                     warehouse_upper_bound=g.tensor("warehouse_upper_bound"))
    res, mean_loss, grads = orc.train_step_gradients(pol, c["periods"], c["problem_params"], data, c["observation_params"],
                                                     c["ignore"])
    assert torch.equal(res.per_period, g.tensor("rewards"))
    assert float(res.total) == float(g.z["total"]) and float(res.reported) == float(g.z["reported"])
    assert float(mean_loss) == float(g.z["mean_loss"])
    for k, v in g.states(1).items():   # (slim: states/0 = initial, states/1 = final)
        assert torch.equal(res.final_obs[k], v), k
    ref_grads = g.grads
    keys = sorted(ref_grads.keys(), key=lambda s_: (int(s_.split(".")[2]), s_.split(".")[3] != "weight"))
    for k, mine in zip(keys, grads):
        assert torch.equal(mine, ref_grads[k]), (k, float((mine - ref_grads[k]).abs().max()))
