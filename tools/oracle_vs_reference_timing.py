#!/usr/bin/env python
"""Certifies `bench.py`'s `cpu_baseline`: the oracle (the CPU restatement that travels to the GPU box) and the
upstream reference imported from /root/reference, timed side by side in THIS container on the same inputs, same weights,
same thread count - one training step (rollout + backward) of the benchmark configuration on a bounded sample.

    python tools/oracle_vs_reference_timing.py [--scenarios 2048] [--threads 8] [--reps 3] [--out profiles/r03_oracle_vs_reference_timing.json]

Prints and writes {reference_s, oracle_s, ratio = oracle / reference, threads, shape, bit_equal_total}.  A ratio near 1 means
the `cpu_baseline` of the bench line is the reference's own CPU speed to that factor.
"""
import argparse
import copy
import json
import os
import statistics
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="cfg3_one_warehouse_16_vanilla")
    ap.add_argument("--scenarios", type=int, default=2048)
    ap.add_argument("--periods", type=int, default=100)
    ap.add_argument("--hidden", type=int, nargs="*", default=[512, 512, 512])
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import reference_harness as rh
    from cases import CASES, apply_overrides
    from oracle import inventory_oracle as orc
    if not rh.reference_available():
        sys.exit("the upstream reference is not mounted here (/root/reference): this tool runs in the build container only")
    torch.set_num_threads(args.threads)
    case = dict(CASES[args.case])
    case["hidden"] = list(args.hidden)
    ref = rh.load_reference()
    cs, ch = rh.load_reference_configs(case["setting"], case["policy"])
    cs, ch = apply_overrides(case, cs, ch)
    cs_o = copy.deepcopy(cs)
    n, T = args.scenarios, args.periods
    obs_r = defaultdict(lambda: None, cs["observation_params"])
    obs_o = defaultdict(lambda: None, cs_o["observation_params"])
    S = cs["problem_params"]["n_stores"]

    with rh.in_reference_dir():
        sc = ref.Scenario(T, cs["problem_params"], cs["store_params"], cs["warehouse_params"], cs["echelon_params"], n, obs_r,
                          cs["seeds"])
        data_r = sc.get_data()
        torch.manual_seed(1234)
        model = ref.NeuralNetworkCreator().create_neural_network(sc, ch["nn_params"], device="cpu")
        sim, tr, loss = ref.Simulator(device="cpu"), ref.Trainer(device="cpu"), ref.PolicyLoss()

        def ref_step(periods=T, data=data_r):
            model.zero_grad()
            total, _ = tr.simulate_batch(loss, sim, model, periods, cs["problem_params"], dict(data), obs_r, 0, False)
            (total / (len(data["demands"]) * periods * S)).backward()
            return float(total.detach())

        ref_step(min(T, 5), {k: v[:64] for k, v in data_r.items()})  # warm-up (materialises the lazy layers)
        data_o = orc.generate_scenario_data(T, cs_o["problem_params"], cs_o["store_params"], cs_o["warehouse_params"],
                                            cs_o["echelon_params"], n, obs_o, cs_o["seeds"])
        wub = model.warehouse_upper_bound if torch.is_tensor(model.warehouse_upper_bound) else None
        pol = orc.policy_from_state_dict(ch["nn_params"], model.state_dict(), cs_o["problem_params"], wub)
        orc.train_step_gradients(pol, min(T, 5), cs_o["problem_params"], {k: v[:64] for k, v in data_o.items()}, obs_o)
        # the two sides alternate, so that a drift of the host's load hits both alike
        t_ref, t_orc, tot_ref, tot_orc = [], [], None, None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            tot_ref = ref_step()
            t_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            res, _, _ = orc.train_step_gradients(pol, T, cs_o["problem_params"], data_o, obs_o)
            t_orc.append(time.perf_counter() - t0)
            tot_orc = float(res.total.detach())

    r, o = statistics.median(t_ref), statistics.median(t_orc)
    out = {
        "what": "one training step (rollout forward + backward), upstream reference vs the oracle, same inputs / weights / threads",
        "case": args.case, "shape": {"scenarios": n, "stores": S, "periods": T, "hidden": args.hidden},
        "threads": args.threads, "host_cores": os.cpu_count(), "reps": args.reps,
        "reference_s": {"median": r, "all": t_ref}, "oracle_s": {"median": o, "all": t_orc},
        "ratio_oracle_over_reference": o / r,
        "reference_scenario_steps_per_s": n * S * T / r, "oracle_scenario_steps_per_s": n * S * T / o,
        "bit_equal_total": tot_ref == tot_orc, "total": tot_ref,
        "torch": torch.__version__,
    }
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
