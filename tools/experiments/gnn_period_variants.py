import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import torch
from golden_io import Golden
from neural_inventory_control_amd.gnn_rollout import GnnRollout
from neural_inventory_control_amd.rollout import KernelTimer
import test_gpu_rollout as T
for name in T.GNN_CASES:
    g = Golden(name); c = g.fresh_config()
    data = {k: v.to("cuda") for k, v in g.data.items()}
    model = T._model(g, c)
    eng = GnnRollout(model, c["problem_params"], "cuda")
    eng.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
    T._load(model, g)
    eng.timer = KernelTimer(record_order=True)
    eng.run(data, c["periods"], c["ignore"], train=True, observation_params=c["observation_params"])
    print(name, data["initial_inventories"].shape, data["initial_warehouse_inventories"].shape, sorted({k for t, k in eng.timer.order if "period" in k}))
