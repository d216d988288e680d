#!/usr/bin/env python
"""Where does the HIP engine's gradient differ from the reference's golden vectors?  (GPU box; diagnostic, not a test.)

For every MLP golden case: per-tensor relative gradient error of
  fused      FusedRollout as shipped (MFMA GEMMs, fused thin-layer backward, all-period weight gradients)
  nothin     ... with the fused thin-layer backward off
  perperiod  ... with per-period weight gradients
  hybrid     tests/host_rollout.py on the device: the HIP env-step / head kernels in the same sweep, torch (rocBLAS)
             matmuls instead of the MFMA GEMMs
and the forward trajectory error (orders, states) per period of `fused` against the golden actions / states.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import host_rollout as hr  # noqa: E402
import kernel_checks as kc  # noqa: E402
from golden_io import Golden, case_names  # noqa: E402
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator  # noqa: E402
from neural_inventory_control_amd.rollout import FusedRollout  # noqa: E402

DEV = "cuda:0"
HEAD = {"vanilla_one_store": "softplus", "vanilla_warehouse": "warehouse", "vanilla_serial": "serial",
        "vanilla_transshipment": "warehouse"}


def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))


def keys_of(ref):
    return sorted(ref.keys(), key=lambda s: (int(s.split(".")[2]), s.split(".")[3] != "weight"))


def engine(g, c, **flags):
    class _Sc:
        pass
    sc = _Sc()
    sc.problem_params = c["problem_params"]
    sc.store_params = {"demand": {"mean": [float(x) for x in np.atleast_1d(g.z["mutated_mean"])]}}
    model = NeuralNetworkCreator().create_neural_network(sc, c["nn_params"], device=DEV)
    model.warehouse_upper_bound = g.tensor("warehouse_upper_bound").to(DEV)
    eng = FusedRollout(model, c["problem_params"], DEV)
    for k, v in flags.items():
        setattr(eng, k, v)
    data = {k: v.to(DEV) for k, v in g.data.items()}
    F = data["initial_inventories"].shape[1] * data["initial_inventories"].shape[2]
    if c["policy"] != "vanilla_one_store":
        F += sum(int(np.prod(data[k].shape[1:])) for k in ("initial_warehouse_inventories", "initial_echelon_inventories")
                 if k in data)
    eng.materialize(F)
    model.load_state_dict({k: v.to(DEV) for k, v in g.params.items()})
    total, _ = eng.run(data, c["periods"], c["ignore"], train=True, observation_params=c["observation_params"])
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    return eng, float(total), {k: named[k].grad.detach().cpu().clone() for k in named}


def main():
    out = {}
    be = kc.HipBackend()
    for name in [n for n in case_names() if n.endswith("vanilla")]:
        g = Golden(name)
        c = g.fresh_config()
        ref = g.grads
        keys = keys_of(ref)
        rec = {}
        for tag, flags in (("fused", {}), ("nothin", {"use_thin": False}), ("perperiod", {"batch_wgrad": False}),
                           ("nosmall", {"use_small": False})):
            eng, total, grads = engine(g, c, **flags)
            rec[tag] = {"small_route": eng.small is not None,
                        "total_rel": abs(total - float(g.z["total"])) / abs(float(g.z["total"])),
                        "grad_rel": [rel(grads[k], ref[k]) for k in keys]}
            if tag == "nosmall" and eng.small is None:
                # forward trajectory of the per-period route against the golden trajectory
                prob, T, B = eng.prob, c["periods"], c["n"]
                traj = []
                for t in range(T):
                    so, wo, eo = eng._order_views(eng.orders[t], prob)
                    a = g.actions(t)
                    d_ord = float((so[:, :, :B].permute(2, 0, 1).cpu() - a["stores"]).abs().max())
                    st = eng._views(eng.states[t + 1], prob)
                    nx = g.states(t + 1)
                    d_st = float((st.store[:, :, :B].permute(2, 0, 1).cpu() - nx["store_inventories"]).abs().max())
                    d_wh = (float((st.wh[:, :, :B].permute(2, 0, 1).cpu() - nx["warehouse_inventories"]).abs().max())
                            if prob.Wn else 0.0)
                    traj.append((d_ord, d_st, d_wh))
                rec["traj_max_abs(orders,store,wh)"] = [max(x[i] for x in traj) for i in range(3)]
                rec["traj_scale"] = float(g.states(T)["store_inventories"].abs().max())
        lay = [(g.params[f"net.master.{i}.weight"], g.params[f"net.master.{i}.bias"])
               for i in sorted({int(k.split(".")[2]) for k in g.params})]
        o = hr.run(be, c["problem_params"], g.data, lay, head=HEAD[c["policy"]], periods=c["periods"], ignore=c["ignore"],
                   ub=float(g.z["warehouse_upper_bound"][0]), adjacency=c["problem_params"].get("warehouse_store_adjacency"),
                   transshipment=c["nn_params"].get("transshipment", False))
        rec["hybrid"] = {"total_rel": abs(float(o["total"]) - float(g.z["total"])) / abs(float(g.z["total"])),
                         "grad_rel": [rel(a, ref[k]) for a, k in zip(o["grads"], keys)]}
        out[name] = rec
        print(name)
        for tag in ("fused", "nothin", "perperiod", "nosmall", "hybrid"):
            r = rec[tag]
            print(f"   {tag:10s} total {r['total_rel']:.1e}  grads " + " ".join(f"{e:.1e}" for e in r["grad_rel"]))
        if "traj_scale" in rec:
            print("   trajectory max abs diff (orders, store, wh):", rec["traj_max_abs(orders,store,wh)"], "scale", rec["traj_scale"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_parity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
