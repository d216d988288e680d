"""Host-side breakdown of one FusedRollout step for a small workload (where Python overhead matters)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
setting, policy, sc, data, model, eng, n, T, desc = bench.build_case(wl, torch.device("cuda", 0), 0)
opt = torch.optim.Adam(model.parameters(), lr=3e-4)
def step():
    opt.zero_grad(set_to_none=True)
    eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
