"""A user script written the way scripts against the reference are written (its README and main_run.py:1-3,75-143 are the
model for the STYLE — star-import of `trainer`, `Scenario` -> `DatasetCreator` -> torch `DataLoader` -> `NeuralNetworkCreator`
-> Adam -> `Trainer.train` / `.test` driven by a settings YAML and a hyper-parameter YAML in the reference's schema).
Builder-written test material: the only line that differs from what a reference user would type is the first import,
which puts the flat modules on sys.path.

    python reference_style_run.py <train|test> <settings.yml> <hyperparams.yml> <out_dir>
"""
import sys

import yaml

import neural_inventory_control_amd.compat  # noqa: F401  (the one added line)
from trainer import *  # noqa: F401,F403

mode, setting_file, hyper_file, out_dir = sys.argv[1:5]
with open(setting_file) as f:
    cfg = yaml.safe_load(f)
with open(hyper_file) as f:
    hyper = yaml.safe_load(f)

problem_params, store_params = cfg["problem_params"], cfg["store_params"]
warehouse_params, echelon_params = cfg["warehouse_params"], cfg["echelon_params"]
by_set = cfg["params_by_dataset"]
observation_params = DefaultDict(lambda: None, cfg["observation_params"])
device = "cuda:0" if torch.cuda.is_available() else "cpu"

creator = DatasetCreator()
n_dev = by_set["dev"]["n_samples"]
scenario = Scenario(max(by_set["train"]["periods"], by_set["dev"]["periods"]), problem_params, store_params,
                    warehouse_params, echelon_params, by_set["train"]["n_samples"] + n_dev, observation_params, cfg["seeds"])
train_set, dev_set = creator.create_datasets(scenario, split=True, by_sample_indexes=True, sample_index_for_split=n_dev)
test_scenario = Scenario(by_set["test"]["periods"], problem_params, store_params, warehouse_params, echelon_params,
                         by_set["test"]["n_samples"], observation_params, cfg["test_seeds"])
test_set = creator.create_datasets(test_scenario, split=False)
loaders = {"train": DataLoader(train_set, batch_size=by_set["train"]["batch_size"], shuffle=True),
           "dev": DataLoader(dev_set, batch_size=by_set["dev"]["batch_size"], shuffle=False),
           "test": DataLoader(test_set, batch_size=by_set["test"]["batch_size"], shuffle=False)}

model = NeuralNetworkCreator().create_neural_network(test_scenario, hyper["nn_params"], device=device)
optimizer = torch.optim.Adam(model.parameters(), lr=hyper["optimizer_params"]["learning_rate"])
loss_function, simulator, trainer = PolicyLoss(), Simulator(device=device), Trainer(device=device)

tp = hyper["trainer_params"]
tp["base_dir"] = out_dir
tp["save_model_folders"] = [trainer.get_year_month_day(), hyper["nn_params"]["name"]]
tp["save_model_filename"] = trainer.get_time_stamp()
if tp["load_previous_model"]:
    model, optimizer = trainer.load_model(model, optimizer, tp["load_model_path"])

if mode == "train":
    trainer.train(tp["epochs"], loss_function, simulator, model, loaders, optimizer, problem_params, observation_params,
                  by_set, tp)
    print("train losses:", " ".join(f"{x:.6f}" for x in trainer.all_train_losses))
    print("dev losses:", " ".join(f"{x:.6f}" for x in trainer.all_dev_losses))
_, reported = trainer.test(loss_function, simulator, model, loaders, optimizer, problem_params, observation_params, by_set,
                           discrete_allocation=store_params["demand"]["distribution"] == "poisson")
print(f"Average per-period test loss: {reported}")
