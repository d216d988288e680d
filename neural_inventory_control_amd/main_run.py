"""Thin driver with the reference's command line (main_run.py:7-148) on top of this package:

    python -m neural_inventory_control_amd.main_run train|test <setting> <hyperparams> [--config-dir DIR]

`<setting>` / `<hyperparams>` are either paths to YAML files or names resolved as
`DIR/settings/<name>.yml` and `DIR/policies_and_hyperparams/<name>.yml` — the reference's `config_files/` tree works
unchanged (the YAML schema is the API; no config is shipped here).  Not accelerated code: it only wires Scenario ->
datasets -> device-resident batches -> policy -> Adam -> Trainer, on the GPU the process is bound to (one process per
GPU under `torch.distributed.run`; scenarios are then sharded across ranks).

Differences from the reference script, on purpose: `train` also runs the test pass afterwards (its README says so; its
`elif` at main_run.py:131 skips it) and batches are device-resident (no per-sample collate).  Real-data settings
(`split_by_period`) read the demand / feature files their YAML names, exactly like the reference.
"""
import argparse
import os
from collections import defaultdict

import torch
import yaml

from . import parallel
from .data_handling import DatasetCreator, DeviceBatches, Scenario
from .environment import Simulator
from .loss_functions import PolicyLoss
from .neural_networks import NeuralNetworkCreator
from .trainer import Trainer

SETTING_KEYS = ("seeds", "test_seeds", "problem_params", "params_by_dataset", "observation_params", "store_params",
                "warehouse_params", "echelon_params", "sample_data_params")
HYPERPARAM_KEYS = ("trainer_params", "optimizer_params", "nn_params")


def _load_yaml(arg, config_dir, sub):
    path = arg if os.path.isfile(arg) else os.path.join(config_dir, sub, arg + ".yml")
    with open(path) as f:
        return yaml.safe_load(f)


def materialize_policy(model, simulator, dataset, problem_params, observation_params, periods, device):
    """One forward of the policy on the first two samples (no gradients): creates the parameters of LazyLinear layers
    exactly as the reference's first training forward would (neural_networks.py:88)."""
    batch = {k: v[:2].to(device) for k, v in dataset.data.items()}
    with torch.no_grad():
        obs, _ = simulator.reset(periods, problem_params, batch, observation_params)
        obs = dict(obs)
        obs["internal_data"] = simulator._internal_data
        model(obs)


def build(config_setting, config_hyperparams, device, rank=0, world_size=1):
    """Everything main_run.py builds between reading the YAMLs and calling the trainer (main_run.py:33-118)."""
    (seeds, test_seeds, problem_params, params_by_dataset, observation_params, store_params, warehouse_params,
     echelon_params, sample_data_params) = [config_setting[k] for k in SETTING_KEYS]
    trainer_params, optimizer_params, nn_params = [config_hyperparams[k] for k in HYPERPARAM_KEYS]
    observation_params = defaultdict(lambda: None, observation_params)
    creator = DatasetCreator()
    if sample_data_params and sample_data_params.get("split_by_period"):
        # real data (main_run.py:50-66): train, dev and test sets are the SAME products over different week ranges - one
        # scenario, per-sample tensors shared, demands and time features sliced along the period axis
        scenario = Scenario(None, problem_params, store_params, warehouse_params, echelon_params,
                            params_by_dataset["train"]["n_samples"], observation_params, seeds)
        train_set, dev_set, test_set = creator.create_datasets(
            scenario, split=True, by_period=True,
            periods_for_split=[sample_data_params[k] for k in ("train_periods", "dev_periods", "test_periods")])
        test_scenario = scenario
    else:
        max_periods = max(params_by_dataset["train"]["periods"], params_by_dataset["dev"]["periods"])
        scenario = Scenario(max_periods, problem_params, store_params, warehouse_params, echelon_params,
                            params_by_dataset["train"]["n_samples"] + params_by_dataset["dev"]["n_samples"],
                            observation_params, seeds)
        train_set, dev_set = creator.create_datasets(scenario, split=True, by_sample_indexes=True,
                                                     sample_index_for_split=params_by_dataset["dev"]["n_samples"])
        test_scenario = Scenario(params_by_dataset["test"]["periods"], problem_params, store_params, warehouse_params,
                                 echelon_params, params_by_dataset["test"]["n_samples"], observation_params, test_seeds)
        test_set = creator.create_datasets(test_scenario, split=False)

    def loader(ds, key, shuffle):
        return DeviceBatches(ds, params_by_dataset[key]["batch_size"], shuffle=shuffle, device=device, rank=rank,
                             world_size=world_size)

    data_loaders = {"train": loader(train_set, "train", True), "dev": loader(dev_set, "dev", False),
                    "test": loader(test_set, "test", False)}
    model = NeuralNetworkCreator().create_neural_network(test_scenario, nn_params, device=device)
    simulator = Simulator(device=device)
    if world_size > 1 or parallel.active():
        # replicas must start from identical parameters: materialise the lazy layers with one throw-away forward, then
        # take rank 0's (only gradients are all-reduced afterwards)
        materialize_policy(model, simulator, train_set, problem_params, observation_params,
                           params_by_dataset["train"]["periods"], device)
        parallel.broadcast_model(model, src=0)
        if not parallel.parameters_in_sync(model):
            raise RuntimeError("policy parameters differ across ranks after the broadcast")
    # the reference's optimizer (main_run.py: torch.optim.Adam) - as ONE kernel per step where torch offers it (same update rule;
    # the default for-each form is six launches, 30 us of a step that is 1-3 ms on the whole-horizon routes)
    from torch.nn.parameter import UninitializedParameter
    fused = torch.device(device).type == "cuda" and not any(isinstance(p, UninitializedParameter) for p in model.parameters())
    optimizer = torch.optim.Adam(model.parameters(), lr=optimizer_params["learning_rate"], **({"fused": True} if fused else {}))
    trainer = Trainer(device=device)
    trainer_params = dict(trainer_params)
    # optional extension key: replay each generic-route training step from one HIP graph (Trainer.use_step_graph)
    trainer.use_step_graph = trainer_params.get("use_step_graph", "auto")   # true / false / "auto" (closed-form policies only)
    # ... and for the MLP engine's launch sequence: true / false / "auto" (default: decided by measurement, rollout.py)
    trainer.use_rollout_graph = trainer_params.get("use_rollout_graph", "auto")
    trainer_params["base_dir"] = trainer_params.get("base_dir", "saved_models")
    trainer_params["save_model_folders"] = [trainer.get_year_month_day(), nn_params["name"]]
    trainer_params["save_model_filename"] = trainer.get_time_stamp()
    if trainer_params.get("load_previous_model"):
        model, optimizer = trainer.load_model(model, optimizer, trainer_params["load_model_path"])
    return dict(model=model, optimizer=optimizer, trainer=trainer, simulator=simulator,
                loss_function=PolicyLoss(), data_loaders=data_loaders, problem_params=problem_params,
                observation_params=observation_params, params_by_dataset=params_by_dataset, trainer_params=trainer_params,
                store_params=store_params)


def run(mode, config_setting, config_hyperparams, device=None, epochs=None):
    """mode: 'train' (train, then test with the best dev parameters) or 'test'.  Returns the per-period test loss."""
    if mode not in ("train", "test"):
        raise ValueError(f"Invalid argument: {mode}")
    rank, world, dev = parallel.init_from_env()
    device = device or dev
    c = build(config_setting, config_hyperparams, device, rank, world)
    tr = c["trainer"]
    if mode == "train":
        tr.train(epochs if epochs is not None else c["trainer_params"]["epochs"], c["loss_function"], c["simulator"],
                 c["model"], c["data_loaders"], c["optimizer"], c["problem_params"], c["observation_params"],
                 c["params_by_dataset"], c["trainer_params"])
    # discrete allocation on the test set when demand is Poisson (main_run.py:132-142)
    _, report = tr.test(c["loss_function"], c["simulator"], c["model"], c["data_loaders"], c["optimizer"],
                        c["problem_params"], c["observation_params"], c["params_by_dataset"],
                        discrete_allocation=c["store_params"]["demand"]["distribution"] == "poisson")
    if rank == 0:
        print(f"Average per-period test loss: {report}")
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("mode", choices=["train", "test"])
    ap.add_argument("setting", nargs="?", default="one_store_lost")
    ap.add_argument("hyperparams", nargs="?", default="vanilla_one_store")
    ap.add_argument("--config-dir", default="config_files")
    ap.add_argument("--epochs", type=int, default=None, help="override trainer_params.epochs")
    args = ap.parse_args(argv)
    run(args.mode, _load_yaml(args.setting, args.config_dir, "settings"),
        _load_yaml(args.hyperparams, args.config_dir, "policies_and_hyperparams"), epochs=args.epochs)


if __name__ == "__main__":
    main()
