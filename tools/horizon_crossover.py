"""Where the whole-horizon data_driven kernels stop paying: training-step time of the real-data setting's shape (21 stores x 3
warehouses x T=95, data_driven 64 x 64, synthetic stand-in files with as many products as the batch needs) on the whole-horizon
route and on the per-period kernels (eager and replayed from a HIP graph), over batch sizes.  Sets `FusedRollout.horizon_max_scenarios`.

    python tools/horizon_crossover.py > gpurun_out/<round>/horizon_crossover.json
"""
import json
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_inventory_control_amd import workloads  # noqa: E402
from neural_inventory_control_amd.data_handling import DatasetCreator, Scenario  # noqa: E402
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator  # noqa: E402
from neural_inventory_control_amd.rollout import FusedRollout  # noqa: E402

DEV = "cuda:0"


def main():
    T, out = 95, {}
    for n in tuple(int(v) for v in os.environ.get("HC_SIZES", "72,288,1024,2048,4096,8192,16384,32768").split(",")):
        setting = workloads.real_data(n_products=n, seed=1)
        obs = defaultdict(lambda: None, setting["observation_params"])
        shift = obs["demand"]["period_shift"]
        sc = Scenario(shift + T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                      setting["echelon_params"], n, obs, setting["seeds"], device=DEV)
        data = {k: v.to(DEV) for k, v in DatasetCreator().split_by_period(sc, [f"(0, {shift + T})"])[0].items()}
        torch.manual_seed(0)
        model = NeuralNetworkCreator().create_neural_network(sc, workloads.data_driven_policy(), device=DEV)
        rec = {}
        for label, horizon, graph in (("whole_horizon_ms", True, False), ("per_period_eager_ms", False, False),
                                      ("per_period_graph_ms", False, True)):
            eng = FusedRollout(model, setting["problem_params"], DEV)
            eng.materialize(eng.input_rows(data, obs))
            eng.use_horizon, eng.horizon_max_scenarios, eng.use_graph = horizon, 1 << 30, graph
            step = lambda: eng.run(data, T, 16, train=True, observation_params=obs)   # noqa: E731
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            reps = 10 if n <= 4096 else 4
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            rec[label] = round((time.perf_counter() - t0) / reps * 1e3, 3)
            assert (eng.horizon is not None) == horizon
            del eng
            torch.cuda.empty_cache()
        out[str(n)] = rec
        print(n, rec, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
