"""Generates tests/golden/*.npz by RUNNING THE UPSTREAM REFERENCE (this container only).

    python tests/golden/make_golden.py            # all cases + the checkpoint known-answer fixture

For each case in cases.py the reference's own `Scenario`, `NeuralNetworkCreator`, `Simulator`,
`PolicyLoss` and `Trainer.simulate_batch` are executed on CPU and their inputs / outputs are stored:

  cfg_json                      the (pre-mutation) YAML dicts after overrides, sizes, seeds
  data/<key>                    Scenario.get_data() tensors (float32)
  mutated_demand_seed, mutated_mean, mutated_std   the reference's in-place config mutations
  param/<state_dict key>        policy weights (after LazyLinear materialisation)
  warehouse_upper_bound
  rewards (T,B)                 per-period per-scenario cost returned by Simulator.step
  states/<t>/<inventories key>  state BEFORE period t (t = 0..T) for stores / warehouses / echelons
  actions/<t>/<key>             policy output at period t
  total, reported               Trainer.simulate_batch return values
  mean_loss                     total / (B*T*S)           (trainer.py:169)
  grad/<state_dict key>         d mean_loss / d param     (trainer.py:173)

The fixtures are DATA (inputs + expected outputs); no reference source text is stored.
"""
import copy
import json
import os
import sys
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_harness as rh  # noqa: E402
from cases import CASES, SLIM_CASES, apply_overrides  # noqa: E402


def run_case(name, case, ref):
    cs, ch = rh.load_reference_configs(case["setting"], case["policy"])
    cs, ch = apply_overrides(case, cs, ch)
    cfg_record = {
        "case": name, "setting": case["setting"], "policy": case["policy"],
        "n": case["n"], "periods": case["periods"], "ignore": case["ignore"],
        "seeds": copy.deepcopy(cs["seeds"]),
        "problem_params": copy.deepcopy(cs["problem_params"]),
        "observation_params": copy.deepcopy(cs["observation_params"]),
        "store_params": copy.deepcopy(cs["store_params"]),
        "warehouse_params": copy.deepcopy(cs["warehouse_params"]),
        "echelon_params": copy.deepcopy(cs["echelon_params"]),
        "nn_params": copy.deepcopy(ch["nn_params"]),
    }
    obs_params = defaultdict(lambda: None, cs["observation_params"])
    seeds = cs["seeds"]
    real = bool(case.get("real"))
    cfg_record["real"] = real
    if case.get("one_store_from_21"):
        # the one-store real-data blob is not shipped: derive it from the shipped 21-store file (every (product, store)
        # weekly-sales series becomes a one-store sample); the derived tensor lives in a temp dir, the fixture keeps `data/*`
        import tempfile
        src = torch.load(os.path.join(rh.REFERENCE_ROOT, "data_files/favorita_21_stores/weekly_sales.pt"), map_location="cpu")
        derived = os.path.join(tempfile.mkdtemp(), "weekly_sales_one_store.pt")
        torch.save(src.reshape(-1, 1, src.shape[2]).clone(), derived)
        cs["store_params"]["demand"]["file_location"] = derived
        cfg_record["store_params"]["demand"]["file_location"] = "<derived: favorita_21_stores/weekly_sales.pt reshaped to (6048, 1, 171)>"
    with rh.in_reference_dir():
        if real:  # one scenario over all weeks, datasets split by period (main_run.py:50-66); the fixture keeps one short range
            scenario = ref.Scenario(None, cs["problem_params"], cs["store_params"], cs["warehouse_params"],
                                    cs["echelon_params"], case["n"], obs_params, seeds)
            (dataset,) = ref.DatasetCreator().create_datasets(scenario, split=True, by_period=True,
                                                             periods_for_split=[case["period_range"]])
            data = dataset.data
            cfg_record["period_range"] = case["period_range"]
        else:
            scenario = ref.Scenario(case["periods"], cs["problem_params"], cs["store_params"], cs["warehouse_params"],
                                    cs["echelon_params"], case["n"], obs_params, seeds)
            data = scenario.get_data()
        torch.manual_seed(case["torch_seed"])
        model = ref.NeuralNetworkCreator().create_neural_network(scenario, ch["nn_params"], device="cpu")
        sim = ref.Simulator(device="cpu")
        trainer = ref.Trainer(device="cpu")
        loss_fn = ref.PolicyLoss()
        T, B, S = case["periods"], case["n"], cs["problem_params"]["n_stores"]

        # materialise LazyLinear layers with one throw-away forward
        obs0, _ = sim.reset(T, cs["problem_params"], dict(data), obs_params)
        with torch.no_grad():
            o = {k: v for k, v in obs0.items()}
            o["internal_data"] = sim._internal_data
            model(o)
        # perturb closed-form parameters a little so gradients are non-degenerate but keep defaults meaningful
        lazy = torch.nn.parameter.UninitializedParameter  # (policies that never call their `net`, e.g. just_in_time)
        state = {k: v.detach().clone() for k, v in model.state_dict().items() if not isinstance(v, lazy)}

        # --- pass 1: the reference's own simulate_batch + backward (the pinned numbers)
        model.zero_grad()
        total, reported = trainer.simulate_batch(loss_fn, sim, model, T, cs["problem_params"], dict(data),
                                                 obs_params, case["ignore"], False)
        mean_loss = total / (B * T * S)
        if model.trainable and total.requires_grad:
            mean_loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

        # --- pass 2: step-by-step trace (same calls as trainer.py:190-213) to record states/actions/rewards
        rewards, states, actions, features = [], [], [], []
        feat_keys = ["past_demands"] + list(obs_params["time_features"] or [])
        with torch.no_grad():
            obs, _ = sim.reset(T, cs["problem_params"], dict(data), obs_params)
            for t in range(T):
                states.append({k: v.clone() for k, v in obs.items() if k.endswith("inventories")})
                features.append({k: obs[k].clone() for k in feat_keys if k in obs})
                o = {k: v for k, v in obs.items()}
                o["internal_data"] = sim._internal_data
                a = model(o)
                actions.append({k: v.clone() for k, v in a.items()})
                obs, r, term, _, _ = sim.step(a)
                rewards.append(r.clone())
            states.append({k: v.clone() for k, v in obs.items() if k.endswith("inventories")})
        rewards = torch.stack(rewards)
        assert torch.equal(rewards.sum(dim=1).sum(), rewards.sum(dim=1).sum())
        # the traced pass must reproduce the trainer's totals bit-for-bit
        tot2 = 0
        for t in range(T):
            tot2 = tot2 + rewards[t].sum()
        assert float(tot2) == float(total.detach()), (float(tot2), float(total))

    slim = bool(case.get("slim"))
    cfg_record["slim"] = slim
    out = {"cfg_json": np.array(json.dumps(cfg_record))}
    for k, v in data.items():
        if slim:   # inputs are regenerated from the seeds by the tests; the fixture pins them by checksum and shape
            out["data_checksum/" + k] = np.array(float(v.double().sum()))
            out["data_abs_checksum/" + k] = np.array(float(v.double().abs().sum()))
            out["data_shape/" + k] = np.array(v.shape)
        else:
            out["data/" + k] = v.contiguous().numpy()
    out["mutated_demand_seed"] = np.array(seeds["demand"])
    dm = cs["store_params"]["demand"]
    out["mutated_mean"] = np.asarray(dm.get("mean", []), dtype=np.float64)
    if "std" in dm:
        out["mutated_std"] = np.asarray(dm["std"], dtype=np.float64)
    for k, v in state.items():
        out["param/" + k] = v.numpy()
    wub = model.warehouse_upper_bound
    out["warehouse_upper_bound"] = (wub.numpy() if torch.is_tensor(wub) else np.array([float(wub)], dtype=np.float32))
    out["rewards"] = rewards.numpy()
    if slim:
        states, actions, features = [states[0], states[-1]], [], []   # (stored as states/0 and states/1 = the final state)
    for t, st in enumerate(states):
        for k, v in st.items():
            out[f"states/{t}/{k}"] = v.contiguous().numpy()
    for t, ac in enumerate(actions):
        for k, v in ac.items():
            out[f"actions/{t}/{k}"] = v.contiguous().numpy()
    for t, ft in enumerate(features):  # observation features of period t (real-data settings: past-demand window, time features)
        for k, v in ft.items():
            out[f"features/{t}/{k}"] = v.contiguous().numpy()
    fixed = getattr(model, "fixed_nets", None)
    if fixed:  # the frozen quantile forecaster the policy loaded from the reference's quantile_forecasters/ (weights = data)
        for k, v in fixed["quantile_forecaster"].state_dict().items():
            out["forecaster/" + k] = v.detach().cpu().numpy()
    out["total"] = total.detach().numpy()
    out["reported"] = reported.detach().numpy()
    out["mean_loss"] = mean_loss.detach().numpy()
    for k, g in grads.items():
        out["grad/" + k] = g.numpy()
    return out


def make_checkpoint_kat(ref):
    """Known answer held by the reference itself: saved_models/2024_04_23/vanilla_one_store/1713902211.pt stores
    best dev loss 6.854347610473633 (epoch 397).  Re-evaluated here with the reference on one_store_lost seeds,
    Scenario(periods=50, n=65536), dev = first 32768 samples, T=50, ignore 30.  The fixture keeps the
    2,305 weights + expected numbers; the demand is regenerated by the test from the seeds (numpy legacy RNG)."""
    ckpt_path = os.path.join(rh.REFERENCE_ROOT, "saved_models/2024_04_23/vanilla_one_store/1713902211.pt")
    ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    cs, ch = rh.load_reference_configs("one_store_lost", "vanilla_one_store")
    cfg_record = {
        "seeds": copy.deepcopy(cs["seeds"]), "problem_params": copy.deepcopy(cs["problem_params"]),
        "observation_params": copy.deepcopy(cs["observation_params"]), "store_params": copy.deepcopy(cs["store_params"]),
        "warehouse_params": None, "echelon_params": None, "nn_params": copy.deepcopy(ch["nn_params"]),
        "scenario_periods": 50, "scenario_samples": 65536, "dev_samples": 32768, "periods": 50, "ignore": 30,
    }
    obs_params = defaultdict(lambda: None, cs["observation_params"])
    with rh.in_reference_dir():
        scenario = ref.Scenario(50, cs["problem_params"], cs["store_params"], None, None, 65536, obs_params, cs["seeds"])
        train_ds, dev_ds = ref.DatasetCreator().create_datasets(scenario, split=True, by_sample_indexes=True,
                                                               sample_index_for_split=32768)
        model = ref.NeuralNetworkCreator().create_neural_network(scenario, ch["nn_params"], device="cpu")
        sim = ref.Simulator(device="cpu")
        trainer = ref.Trainer(device="cpu")
        data = dev_ds.data
        obs0, _ = sim.reset(50, cs["problem_params"], dict(data), obs_params)
        with torch.no_grad():
            o = dict(obs0)
            o["internal_data"] = sim._internal_data
            model(o)
        model.load_state_dict(ck["model_state_dict"])
        with torch.no_grad():
            total, reported = trainer.simulate_batch(ref.PolicyLoss(), sim, model, 50, cs["problem_params"], dict(data),
                                                     obs_params, 30, False)
        dev_loss = reported.item() / (32768 * 20 * 1)
    print("checkpoint KAT: stored best dev loss", min(ck["all_dev_losses"]), "re-evaluated", dev_loss)
    out = {"cfg_json": np.array(json.dumps(cfg_record)),
           "stored_best_dev_loss": np.array(float(min(ck["all_dev_losses"]))),
           "reevaluated_dev_loss": np.array(dev_loss),
           "total": total.numpy(), "reported": reported.numpy(),
           "demand_checksum": np.array(float(data["demands"].double().sum())),
           "init_inv_checksum": np.array(float(data["initial_inventories"].double().sum()))}
    for k, v in ck["model_state_dict"].items():
        out["param/" + k] = v.cpu().numpy()
    return out


def main():
    ref = rh.load_reference()
    only = set(sys.argv[1:])
    for name, case in list(CASES.items()) + list(SLIM_CASES.items()):
        if only and name not in only:
            continue
        out = run_case(name, case, ref)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: total={float(out['total']):.6f} mean_loss={float(out['mean_loss']):.8f} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")
    if not only or "checkpoint_kat" in only:
        out = make_checkpoint_kat(ref)
        path = os.path.join(HERE, "checkpoint_kat.npz")
        np.savez_compressed(path, **out)
        print(f"checkpoint_kat -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
