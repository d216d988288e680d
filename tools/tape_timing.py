"""Training-step (trainable) / evaluation-pass (other) time of the real-data policies that need no network inside the period loop -
the quantile policies and just-in-time - on the tape route (tape_rollout.py: one batched pass for all periods' decisions + one
whole-horizon launch per direction) against the generic route (Simulator.step + autograd, period by period), on the reference-
generated fixtures (tests/golden: the only place the frozen forecaster's weights travel to the GPU box) at their own size and with
the batch tiled 7 x (280 scenarios, the reference's dev batch is 288).

    python tools/tape_timing.py > gpurun_out/<round>/tape_timing.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

from golden_io import Golden, case_names  # noqa: E402
from neural_inventory_control_amd.environment import Simulator  # noqa: E402
from neural_inventory_control_amd.loss_functions import PolicyLoss  # noqa: E402
from neural_inventory_control_amd.trainer import Trainer  # noqa: E402
import test_gpu_rollout as tg  # noqa: E402

DEV = "cuda:0"


def main():
    out = {}
    for name in [n for n in case_names() if n.startswith("f4_real") and not n.endswith("data_driven")]:
        for tile in tuple(int(v) for v in os.environ.get("TAPE_TILES", "1,7").split(",")):
            g = Golden(name)
            c = g.fresh_config()
            model = tg._model(g, c)
            data = {k: torch.cat([v] * tile, dim=0).to(DEV).contiguous() for k, v in g.data.items()}
            sim = Simulator(device=DEV)
            if g.params:
                obs, _ = sim.reset(c["periods"], c["problem_params"], data, c["observation_params"])
                with torch.no_grad():
                    o = dict(obs)
                    o["internal_data"] = sim._internal_data
                    model(o)
                tg._load(model, g)
            rec = {"scenarios": c["n"] * tile, "periods": c["periods"], "stores": c["problem_params"]["n_stores"]}
            for route in ("tape", "generic"):
                tr = Trainer(device=DEV)
                tr.use_fused_rollout = route == "tape"
                opt = torch.optim.Adam(model.parameters(), lr=1e-4) if any(p.requires_grad for p in model.parameters()) else None

                def step():
                    total, _ = tr.simulate_batch(PolicyLoss(), sim, model, c["periods"], c["problem_params"], data,
                                                 c["observation_params"], c["ignore"], False)
                    if total.requires_grad:
                        opt.zero_grad(set_to_none=True)
                        (total / (c["n"] * tile * c["periods"])).backward()
                        opt.step()
                    return total
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    tot = step()
                torch.cuda.synchronize()
                rec[route + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
                rec["trainable"] = bool(tot.requires_grad)
            if rec["trainable"]:   # the same step through Trainer.do_one_epoch (one batch), replayed from a HIP graph (use_step_graph = "auto")
                from neural_inventory_control_amd.data_handling import DeviceBatches, MyDataset
                ds = MyDataset(c["n"] * tile, {k: v.cpu() for k, v in data.items()})
                loader = DeviceBatches(ds, c["n"] * tile, shuffle=False, device=DEV)
                for label, mode in (("tape_epoch_eager_ms", False), ("tape_epoch_graph_ms", "auto")):
                    # (a fresh policy per mode: parameters that already went through eager backward passes on the default stream
                    # carry gradient accumulators bound to it, which a later capture on a side stream does not survive)
                    m2 = tg._model(g, g.fresh_config())
                    with torch.no_grad():
                        o = dict(sim.reset(c["periods"], c["problem_params"], data, c["observation_params"])[0])
                        o["internal_data"] = sim._internal_data
                        m2(o)
                    tg._load(m2, g)
                    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-4)
                    tr = Trainer(device=DEV)
                    tr.use_step_graph = mode
                    ep = lambda: tr.do_one_epoch(opt2, loader, PolicyLoss(), sim, m2, c["periods"], c["problem_params"],   # noqa: E731
                                                 c["observation_params"], train=True, ignore_periods=c["ignore"])
                    for _ in range(4):
                        ep()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        ep()
                    torch.cuda.synchronize()
                    rec[label] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
            rec["speedup"] = round(rec["generic_ms"] / rec["tape_ms"], 1)
            out[f"{name}_x{tile}"] = rec
            print(name, tile, rec, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
