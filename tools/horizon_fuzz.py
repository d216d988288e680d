"""Differential fuzz of the whole-horizon data_driven kernels against the per-period route of the same engine over random shapes
(stores, warehouses, hidden widths, batch sizes, horizons, window lengths, weight scales): totals, per-period rewards, gradients.

    python tools/horizon_fuzz.py [seed] [iterations]
"""
import os, sys, random, traceback
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_inventory_control_amd import workloads
from neural_inventory_control_amd.data_handling import DatasetCreator, Scenario
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
from neural_inventory_control_amd.rollout import FusedRollout
DEV = "cuda:0"
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    S = rnd.choice([1, 2, 3, 5, 8, 13, 21, 30]); Wn = rnd.choice([1, 2, 3, 4])
    if Wn + S * Wn > 128: Wn = 1
    hidden = [rnd.choice([8, 16, 24, 32, 48, 64]), rnd.choice([8, 16, 32, 40, 64])]
    n = rnd.choice([1, 3, 16, 17, 31, 64, 100, 250]); T = rnd.choice([2, 3, 7, 12, 20]); P = rnd.choice([1, 2, 5, 8])
    try:
        setting = workloads.real_data(n_products=n, n_stores=S, n_warehouses=Wn, weeks=P + T + 4, past_periods=P, seed=it)
        policy = workloads.data_driven_policy(); policy["neurons_per_hidden_layer"] = {"master": hidden}
        obs = defaultdict(lambda: None, setting["observation_params"]); shift = obs["demand"]["period_shift"]
        sc = Scenario(shift + T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n, obs, setting["seeds"], device=DEV)
        data = {k: v.to(DEV) for k, v in DatasetCreator().split_by_period(sc, [f"(0, {shift + T})"])[0].items()}
        torch.manual_seed(it)
        model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
        eng = FusedRollout(model, setting["problem_params"], DEV); eng.materialize(eng.input_rows(data, obs))
        with torch.no_grad():
            for p_ in model.parameters(): p_.add_(rnd.choice([0.02, 0.1, 0.3]) * torch.randn_like(p_))
        out = {}
        for route in ("horizon", "periods"):
            eng.use_horizon = route == "horizon"
            tot, rep = eng.run(data, T, min(1, T - 1), train=True, observation_params=obs)
            torch.cuda.synchronize()
            assert (eng.horizon is not None) == (route == "horizon"), (route, S, Wn, hidden)
            out[route] = (float(tot), eng.per_period_rewards().clone(), [p_.grad.detach().clone() for p_ in model.parameters()])
        a, b = out["horizon"], out["periods"]
        e_tot = abs(a[0] - b[0]) / (abs(b[0]) + 1e-9)
        e_r = float((a[1] - b[1]).abs().max() / (b[1].abs().max() + 1e-9))
        e_g = max(float((x - y).norm() / (y.norm() + 1e-12)) for x, y in zip(a[2], b[2]))
        flag = "" if (e_tot < 1e-5 and e_r < 1e-4 and e_g < 1e-3) else "   <<<<<< CHECK"
        bad += bool(flag)
        print(f"{it:3d} S={S:2d} Wn={Wn} hidden={hidden} n={n:3d} T={T:2d} P={P}: total {e_tot:.1e} rewards {e_r:.1e} grads {e_g:.1e}{flag}", flush=True)
    except Exception as e:
        bad += 1
        print(f"{it:3d} S={S:2d} Wn={Wn} hidden={hidden} n={n:3d} T={T:2d} P={P}: EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
        traceback.print_exc(limit=3)
print("suspicious:", bad)
