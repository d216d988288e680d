"""Live pin of the oracle against the upstream reference imported from /root/reference (THIS CONTAINER ONLY;
skipped wherever the reference tree is absent, e.g. on the GPU box).  Complements the committed golden vectors with
fresh seeds / sizes and the variants the YAMLs do not exercise (profit objective, discrete allocation)."""
import copy
from collections import defaultdict

import pytest
import torch

import reference_harness as rh
from cases import CASES, apply_overrides
from oracle import inventory_oracle as orc

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="upstream reference not mounted")


def _run_both(case, n, periods, ignore, torch_seed, maximize_profit=False, discrete=False):
    ref = rh.load_reference()
    cs, ch = rh.load_reference_configs(case["setting"], case["policy"])
    cs, ch = apply_overrides(case, cs, ch)
    cs["problem_params"]["maximize_profit"] = maximize_profit
    cs_o = copy.deepcopy(cs)
    obs_r = defaultdict(lambda: None, cs["observation_params"])
    obs_o = defaultdict(lambda: None, cs_o["observation_params"])
    with rh.in_reference_dir():
        sc = ref.Scenario(periods, cs["problem_params"], cs["store_params"], cs["warehouse_params"],
                          cs["echelon_params"], n, obs_r, cs["seeds"])
        data_r = sc.get_data()
        torch.manual_seed(torch_seed)
        model = ref.NeuralNetworkCreator().create_neural_network(sc, ch["nn_params"], device="cpu")
        sim, tr = ref.Simulator(device="cpu"), ref.Trainer(device="cpu")
        o, _ = sim.reset(periods, cs["problem_params"], dict(data_r), obs_r)
        with torch.no_grad():
            oo = dict(o)
            oo["internal_data"] = sim._internal_data
            model(oo)
        model.zero_grad()
        total, rep = tr.simulate_batch(ref.PolicyLoss(), sim, model, periods, cs["problem_params"], dict(data_r),
                                       obs_r, ignore, discrete)
        (total / (n * periods * cs["problem_params"]["n_stores"])).backward()
    data_o = orc.generate_scenario_data(periods, cs_o["problem_params"], cs_o["store_params"], cs_o["warehouse_params"],
                                        cs_o["echelon_params"], n, obs_o, cs_o["seeds"])
    for k in data_r:
        assert torch.equal(data_r[k], data_o[k]), k
    wub = model.warehouse_upper_bound if torch.is_tensor(model.warehouse_upper_bound) else None
    pol = orc.policy_from_state_dict(ch["nn_params"], model.state_dict(), cs_o["problem_params"], wub)
    for p in pol.parameters():
        p.grad = None
    res = orc.rollout(pol, periods, cs_o["problem_params"], data_o, obs_o, ignore, discrete)
    (res.total / (n * periods * cs_o["problem_params"]["n_stores"])).backward()
    assert float(res.total.detach()) == float(total.detach())
    assert float(res.reported.detach()) == float(rep.detach())
    for (k, p), mine in zip(model.named_parameters(), pol.parameters()):
        if p.grad is None:
            assert mine.grad is None or float(mine.grad.abs().max()) == 0.0
        elif hasattr(pol, "param_keys"):  # GNN: forward bit-equal, gradients to ~1 ulp (see test_oracle_golden.py)
            assert float((p.grad - mine.grad).norm() / (p.grad.norm() + 1e-30)) < 1e-6, k
        else:
            assert torch.equal(p.grad, mine.grad), k
    for k in ("store_inventories", "warehouse_inventories", "echelon_inventories"):
        if k in sim.observation:
            assert torch.equal(sim.observation[k], res.final_obs[k]), k


@pytest.mark.parametrize("name", [k for k, v in CASES.items() if not v.get("real")])
def test_fresh_seed_and_size(name):
    _run_both(CASES[name], n=37, periods=9, ignore=2, torch_seed=1234)


@pytest.mark.parametrize("name", [k for k, v in CASES.items() if v.get("real")])
def test_real_data_cases_fresh_products_and_weeks(name):
    """SURVEY 8 f4 live: the reference and the oracle on the shipped Favorita data, other products / weeks / weights than the
    committed fixtures: data dict, totals, final state and gradients."""
    import os
    import tempfile
    case = CASES[name]
    ref = rh.load_reference()
    cs, ch = rh.load_reference_configs(case["setting"], case["policy"])
    cs, ch = apply_overrides(case, cs, ch)
    n, periods, ignore, rng = 23, 11, 2, "(30, 90)"
    if case.get("one_store_from_21"):
        src = torch.load(os.path.join(rh.REFERENCE_ROOT, "data_files/favorita_21_stores/weekly_sales.pt"), map_location="cpu")
        derived = os.path.join(tempfile.mkdtemp(), "one.pt")
        torch.save(src.reshape(-1, 1, src.shape[2])[100:].clone(), derived)
        cs["store_params"]["demand"]["file_location"] = derived
    cs_o = copy.deepcopy(cs)
    obs_r = defaultdict(lambda: None, cs["observation_params"])
    obs_o = defaultdict(lambda: None, cs_o["observation_params"])
    with rh.in_reference_dir():
        sc = ref.Scenario(None, cs["problem_params"], cs["store_params"], cs["warehouse_params"], cs["echelon_params"], n,
                          obs_r, cs["seeds"])
        (ds,) = ref.DatasetCreator().create_datasets(sc, split=True, by_period=True, periods_for_split=[rng])
        data_r = ds.data
        torch.manual_seed(77)
        model = ref.NeuralNetworkCreator().create_neural_network(sc, ch["nn_params"], device="cpu")
        sim, tr = ref.Simulator(device="cpu"), ref.Trainer(device="cpu")
        total, rep = tr.simulate_batch(ref.PolicyLoss(), sim, model, periods, cs["problem_params"], dict(data_r), obs_r,
                                       ignore, False)
        if total.requires_grad:
            (total / (n * periods * cs["problem_params"]["n_stores"])).backward()
        full = orc.generate_scenario_data(None, cs_o["problem_params"], cs_o["store_params"], cs_o["warehouse_params"],
                                          cs_o["echelon_params"], n, obs_o, cs_o["seeds"])
    (data_o,) = orc.split_data_by_period(full, [rng], obs_o, cs_o["problem_params"])
    assert set(data_r) == set(data_o)
    for k in data_r:
        assert torch.equal(data_r[k], data_o[k]), k
    lazy = torch.nn.parameter.UninitializedParameter
    state = {k: v for k, v in model.state_dict().items() if not isinstance(v, lazy)}
    fixed = getattr(model, "fixed_nets", None)
    pol = orc.policy_from_state_dict(ch["nn_params"], state, cs_o["problem_params"], None,
                                     forecaster_state=fixed["quantile_forecaster"].state_dict() if fixed else None)
    res = orc.rollout(pol, periods, cs_o["problem_params"], data_o, obs_o, ignore)
    assert float(res.total.detach()) == float(total.detach()) and float(res.reported.detach()) == float(rep.detach())
    if res.total.requires_grad:
        (res.total / (n * periods * cs_o["problem_params"]["n_stores"])).backward()
        for (k, p), mine in zip(((k, p) for k, p in model.named_parameters() if not isinstance(p, lazy)), pol.parameters()):
            if p.grad is not None and float(p.grad.abs().max()) > 0:
                assert float((p.grad - mine.grad).norm()) <= 1e-6 * float(p.grad.norm()), k
    for k in ("store_inventories", "warehouse_inventories"):
        if k in sim.observation:
            assert torch.equal(sim.observation[k], res.final_obs[k]), k


@pytest.mark.parametrize("name", ["cfg1_one_store_lost_vanilla", "cfg3_one_warehouse_5_vanilla"])
def test_profit_objective(name):
    _run_both(CASES[name], n=21, periods=7, ignore=0, torch_seed=5, maximize_profit=True)


@pytest.mark.parametrize("name", ["cfg1_one_store_lost_vanilla", "cfg2_one_store_backlogged_base_stock",
                                  "cfg5_many_warehouses_2x10_vanilla"])
def test_discrete_allocation(name):
    _run_both(CASES[name], n=21, periods=7, ignore=0, torch_seed=6, discrete=True)


# ---- the reference's own YAML files drive this package unchanged (the config schema is the API) ----------------------

_DRIVER_PAIRS = [("one_store_lost", "vanilla_one_store"), ("one_store_backlogged", "vanilla_one_store"),
                 ("one_store_backlogged", "base_stock"), ("one_store_backlogged", "capped_base_stock"),
                 ("one_warehouse_lost_demand", "vanilla_warehouse"), ("many_warehouses_lost_demand", "vanilla_warehouse"),
                 ("serial_system", "vanilla_serial"), ("serial_system", "echelon_stock"),
                 ("transshipment_backlogged", "vanilla_transshipment"), ("one_warehouse_lost_demand", "gnn")]


@pytest.mark.parametrize("setting,policy", _DRIVER_PAIRS)
def test_driver_builds_from_reference_config_files(setting, policy):
    """`main_run.build` on the reference's config_files (sample counts shrunk): scenario, datasets, loaders, policy,
    optimizer and trainer parameters are all constructed from the YAML exactly as the reference's main_run.py reads them."""
    import yaml
    from neural_inventory_control_amd import main_run
    import os
    cfg = os.path.join(rh.REFERENCE_ROOT, "config_files")
    cs = yaml.safe_load(open(f"{cfg}/settings/{setting}.yml"))
    ch = yaml.safe_load(open(f"{cfg}/policies_and_hyperparams/{policy}.yml"))
    for k in cs["params_by_dataset"]:
        cs["params_by_dataset"][k].update(n_samples=64, batch_size=32, periods=min(cs["params_by_dataset"][k]["periods"], 30))
    c = main_run.build(cs, ch, "cpu")
    assert len(c["data_loaders"]["train"].dataset) == 64 and len(c["data_loaders"]["test"].dataset) == 64
    assert type(c["model"]).__name__ in ("VanillaOneStore", "BaseStock", "CappedBaseStock", "VanillaWarehouse",
                                         "VanillaSerial", "EchelonStock", "GNN")
    batch = next(iter(c["data_loaders"]["dev"]))
    assert batch["demands"].shape[0] == 32 and batch["demands"].shape[1] == cs["problem_params"]["n_stores"]
