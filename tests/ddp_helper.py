"""Worker of tests/test_gpu_distributed.py: trains the small one-store experiment through `main_run.run` and prints the test
loss and a parameter checksum as one JSON line (rank 0).  Run alone or under torch.distributed.run (ranks may share one
GPU with NIC_DIST_BACKEND=gloo)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import main_run, parallel, workloads  # noqa: E402


def configs():
    setting, policy, _, _, _ = workloads.get("cfg1")
    setting["test_seeds"] = {k: v + 8 for k, v in setting["seeds"].items()}
    setting["sample_data_params"] = {"split_by_period": False}
    setting["params_by_dataset"] = {"train": {"n_samples": 4096, "batch_size": 1024, "periods": 30, "ignore_periods": 10},
                                    "dev": {"n_samples": 2048, "batch_size": 2048, "periods": 40, "ignore_periods": 20},
                                    "test": {"n_samples": 2048, "batch_size": 2048, "periods": 120, "ignore_periods": 60}}
    hyper = {"trainer_params": {"epochs": 6, "do_dev_every_n_epochs": 2, "print_results_every_n_epochs": 10 ** 6,
                                "save_model": False, "load_previous_model": False, "load_model_path": None,
                                "choose_best_model_on": "dev_loss", "epochs_between_save": 10},
             "optimizer_params": {"learning_rate": 0.01}, "nn_params": policy}
    return setting, hyper


if __name__ == "__main__":
    setting, hyper = configs()
    torch.manual_seed(0)
    loss = main_run.run("train", setting, hyper)
    if parallel.rank() == 0:
        print("DDP_RESULT " + json.dumps({"world": parallel.world_size(), "test_loss": loss,
                                           "backend": torch.distributed.get_backend() if parallel.initialized() else None}))
