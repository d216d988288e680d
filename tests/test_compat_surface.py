"""module names are API.  `from trainer import *` (main_run.py:3 of the reference) must resolve against this repo
through the flat modules of `neural_inventory_control_amd/compat`.  Runs in a subprocess because other tests import the
upstream reference's own flat modules of the same names into this interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "scripts", "reference_style_run.py")
NAMES = ["Trainer", "Simulator", "Scenario", "DatasetCreator", "MyDataset", "PolicyLoss", "NeuralNetworkCreator",
         "MyNeuralNetwork", "DataLoader", "Dataset", "torch", "nn", "np", "pd", "DefaultDict", "copy", "datetime", "os"]


def _env(extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([p for p in (extra_path, ROOT, env.get("PYTHONPATH")) if p])
    return env


@pytest.mark.parametrize("how", ["import_line", "pythonpath"])
def test_star_import_chain_resolves(how):
    compat = os.path.join(ROOT, "neural_inventory_control_amd", "compat")
    first = "import neural_inventory_control_amd.compat\n" if how == "import_line" else ""
    code = first + (
        "from trainer import *\n"
        f"missing = [n for n in {NAMES!r} if n not in globals()]\n"
        "assert not missing, missing\n"
        "import trainer, environment, data_handling, neural_networks, loss_functions, shared_imports\n"
        "import neural_inventory_control_amd.trainer as pkg\n"
        "assert Trainer is pkg.Trainer and trainer.Trainer is pkg.Trainer\n"
        "assert environment.Scenario is data_handling.Scenario  # the reference's chain: environment star-imports data_handling\n"
        "t = Trainer(device='cpu'); assert t.all_train_losses == [] and callable(t.simulate_batch)\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp",
                       env=_env(compat if how == "pythonpath" else None))
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


@pytest.mark.gpu
def test_reference_style_script_trains_and_tests(tmp_path):
    """tests/scripts/reference_style_run.py (star-import of `trainer`, torch DataLoader with per-sample collate, Adam,
    Trainer.train / .test, the reference's YAML schema) end to end on the HIP engine."""
    import glob
    import yaml
    sys.path.insert(0, ROOT)
    from neural_inventory_control_amd import workloads
    setting, policy, _, _, _ = workloads.get("cfg2")
    setting["test_seeds"] = {k: v + 8 for k, v in setting["seeds"].items()}
    setting["params_by_dataset"] = {"train": {"n_samples": 512, "batch_size": 256, "periods": 30, "ignore_periods": 10},
                                    "dev": {"n_samples": 256, "batch_size": 256, "periods": 30, "ignore_periods": 10},
                                    "test": {"n_samples": 256, "batch_size": 256, "periods": 80, "ignore_periods": 30}}
    setting["sample_data_params"] = {"split_by_period": False}
    hyper = {"trainer_params": {"epochs": 30, "do_dev_every_n_epochs": 5, "print_results_every_n_epochs": 1000,
                                "save_model": True, "epochs_between_save": 1, "choose_best_model_on": "dev_loss",
                                "load_previous_model": False, "load_model_path": ""},
             "optimizer_params": {"learning_rate": 0.003}, "nn_params": policy}
    (tmp_path / "s.yml").write_text(yaml.safe_dump(setting))
    (tmp_path / "h.yml").write_text(yaml.safe_dump(hyper))
    r = subprocess.run([sys.executable, SCRIPT, "train", str(tmp_path / "s.yml"), str(tmp_path / "h.yml"), str(tmp_path)],
                       capture_output=True, text=True, cwd=str(tmp_path), env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = {ln.split(":")[0]: ln.split(":", 1)[1] for ln in r.stdout.splitlines() if ":" in ln}
    train = [float(x) for x in lines["train losses"].split()]
    assert len(train) == 30 and train[-1] < 0.8 * train[0]
    test_loss = float(lines["Average per-period test loss"])
    assert 0.0 < test_loss < train[0]
    saved = glob.glob(str(tmp_path / "*" / "vanilla_one_store" / "*.pt"))
    assert len(saved) == 1
    # `test` mode from the checkpoint the script wrote reproduces the test loss
    hyper["trainer_params"].update(load_previous_model=True, load_model_path=saved[0])
    (tmp_path / "h2.yml").write_text(yaml.safe_dump(hyper))
    r2 = subprocess.run([sys.executable, SCRIPT, "test", str(tmp_path / "s.yml"), str(tmp_path / "h2.yml"), str(tmp_path)],
                        capture_output=True, text=True, cwd=str(tmp_path), env=_env(), timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    again = float(r2.stdout.strip().split("Average per-period test loss:")[-1])
    assert abs(again - test_loss) <= 1e-5 * test_loss
