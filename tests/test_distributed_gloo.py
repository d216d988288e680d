"""world_size-2 gloo tests of the data-parallel path (runs on CPU): the flat-buffer gradient all-reduce, scenario
sharding, and that two sharded ranks reproduce the single-process gradient of the same global batch.  The per-rank
gradients here come from the oracle (the HIP engine needs a GPU); what is under test is the sharding + collective."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from golden_io import Golden
    from neural_inventory_control_amd import parallel
    from oracle import inventory_oracle as orc
    r, w, dev = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    g = Golden("cfg3_one_warehouse_5_vanilla")
    c = g.fresh_config()
    data = g.data
    B, T, S = c["n"], c["periods"], c["problem_params"]["n_stores"]
    lo, hi = parallel.shard_range(B, rank, world)
    shard = {k: v[lo:hi] for k, v in data.items()}
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"], g.tensor("warehouse_upper_bound"))
    res = orc.rollout(pol, T, c["problem_params"], shard, c["observation_params"], c["ignore"])
    (res.total / (B * T * S)).backward()  # every rank divides by the GLOBAL B*T*S

    class M(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in ps])
    m = M(pol.parameters())
    for p, src in zip(m.ps, pol.parameters()):
        p.grad = src.grad.clone()
    total, reported = parallel.GradientAllReducer.get(m).all_reduce(res.total.detach(), res.reported.detach())
    if rank == 0:
        ret["grads"] = [p.grad.clone() for p in m.ps]
        ret["total"], ret["reported"] = float(total), float(reported)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process():
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    from golden_io import Golden
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    g = Golden("cfg3_one_warehouse_5_vanilla")
    ref = g.grads
    keys = sorted(ref.keys(), key=lambda s: (int(s.split(".")[2]), s.split(".")[3] != "weight"))
    for k, got in zip(keys, ret["grads"]):
        rel = float((got - ref[k]).norm() / (ref[k].norm() + 1e-30))
        assert rel < 2e-6, (k, rel)
    assert abs(ret["total"] - float(g.z["total"])) <= 1e-6 * abs(float(g.z["total"]))
    assert abs(ret["reported"] - float(g.z["reported"])) <= 1e-6 * abs(float(g.z["reported"]))


def test_shard_range_partitions():
    from neural_inventory_control_amd import parallel
    for n, w in ((10, 3), (65536, 8), (7, 8), (1, 1)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


# ---- sharded scenario generation, parameter broadcast, empty per-rank slices ---------------------------------

def _shard_worker(rank, world, port, ret):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from collections import defaultdict
    from neural_inventory_control_amd import parallel, workloads
    from neural_inventory_control_amd.data_handling import Scenario
    parallel.init_from_env(backend="gloo")
    out = {}
    for name, n_total, T in (("cfg3", 21, 7), ("cfg5", 10, 5), ("cfg1", 13, 6)):
        setting, _, _, _, _ = workloads.get(name)
        lo, hi = parallel.shard_range(n_total, rank, world)
        obs = defaultdict(lambda: None, setting["observation_params"])
        sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                      setting["echelon_params"], hi - lo, obs, setting["seeds"], scenario_offset=lo, num_total=n_total)
        out[name] = {k: v.clone() for k, v in sc.get_data().items()}
        # the device sampler's exchange: per-store float64 sums of the local traces, added over the ranks
        fake = object.__new__(Scenario)
        fake.num_samples, fake.num_total, fake.periods = hi - lo, n_total, T
        gen = torch.Generator().manual_seed(5)
        full = torch.rand(T, setting["problem_params"]["n_stores"], n_total, generator=gen)
        fake.demands_soa = torch.nn.functional.pad(full[:, :, lo:hi], (0, 64 - (hi - lo)))
        out[name + "_mean"] = fake.global_store_demand_mean(None)
        out[name + "_mean_ref"] = (full.double().sum(dim=(0, 2)) / (n_total * T)).float()
    # replicas that drew different initial weights are brought in line by the broadcast
    torch.manual_seed(100 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ELU(), torch.nn.Linear(7, 2))
    m.warehouse_upper_bound = torch.tensor([float(rank + 1)])
    out["sync_before"] = parallel.parameters_in_sync(m)
    parallel.broadcast_model(m, src=0)
    out["sync_after"] = parallel.parameters_in_sync(m)
    out["ub"] = float(m.warehouse_upper_bound)
    # reducer: gradient tensors keep their identity, contents become the sum over ranks
    for p in m.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    ids = [p.grad.data_ptr() for p in m.parameters()]
    tot, = parallel.GradientAllReducer.get(m).all_reduce(torch.tensor(float(rank)))
    out["reducer_ok"] = (all(p.grad.data_ptr() == i for p, i in zip(m.parameters(), ids))
                         and all(bool((p.grad == 3.0).all()) for p in m.parameters()) and float(tot) == 1.0)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_scenarios_equal_single_process_dataset():
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from neural_inventory_control_amd.data_handling import Scenario
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(2, 29500 + (os.getpid() + 7) % 2000, ret), nprocs=2, join=True)
    for name, n_total, T in (("cfg3", 21, 7), ("cfg5", 10, 5), ("cfg1", 13, 6)):
        setting, _, _, _, _ = workloads.get(name)
        obs = defaultdict(lambda: None, setting["observation_params"])
        whole = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                         setting["echelon_params"], n_total, obs, setting["seeds"]).get_data()
        for k, v in whole.items():
            cat = torch.cat([ret[0][name][k], ret[1][name][k]], dim=0)
            assert torch.equal(cat, v), (name, k)
        for r in (0, 1):
            assert torch.equal(ret[r][name + "_mean"], ret[r][name + "_mean_ref"]), name
    for r in (0, 1):
        assert ret[r]["sync_before"] is False and ret[r]["sync_after"] is True and ret[r]["ub"] == 1.0
        assert ret[r]["reducer_ok"]


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` must start N ranks itself or fail loudly - never report a 1-rank number as N GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no bench line at all
