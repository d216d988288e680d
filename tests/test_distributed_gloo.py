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


# ---- four ranks, an epoch whose batches do not divide evenly (round 4) -----------------------------------------------------

def _epoch_worker(rank, world, port, ret):
    """One epoch of a 22-scenario dataset in global batches of 9 (9, 9, 4) on four ranks: per-rank slices of a batch are
    ceil(batch / world) long - (3, 3, 3, 0) for a batch of 9 and (1, 1, 1, 1) for the last one - i.e. one rank holds NO scenario of
    the first two batches and still has to join their collectives.  Per-batch gradients come from the oracle; each rank divides by
    the GLOBAL batch like Trainer.do_one_epoch; the all-reduced gradient of every batch must be the single-process gradient."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from golden_io import Golden
    from neural_inventory_control_amd import parallel
    from neural_inventory_control_amd.data_handling import DeviceBatches, MyDataset
    from oracle import inventory_oracle as orc
    parallel.init_from_env(backend="gloo")
    g = Golden("cfg3_one_warehouse_16_vanilla")
    c = g.fresh_config()
    n = 22
    data = {k: v[:n] for k, v in g.data.items()}
    T, S = c["periods"], c["problem_params"]["n_stores"]
    loader = DeviceBatches(MyDataset(n, data), 9, shuffle=True, device="cpu", seed=3, rank=rank, world_size=world)
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"], g.tensor("warehouse_upper_bound"))

    class M(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in ps])
    m = M(pol.parameters())
    out = []
    for batch in loader:
        gb = loader.last_global_batch
        nb = len(batch["demands"])
        for p in pol.parameters():
            p.grad = None
        if nb:
            res = orc.rollout(pol, T, c["problem_params"], batch, c["observation_params"], c["ignore"])
            (res.total / (gb * T * S)).backward()
            tot, rep = res.total.detach(), res.reported.detach()
        else:
            tot = rep = torch.zeros(())
        for p, src in zip(m.ps, pol.parameters()):
            p.grad = src.grad.clone() if src.grad is not None else torch.zeros_like(p)
        tot, rep = parallel.GradientAllReducer.get(m).all_reduce(tot, rep)
        out.append({"local": nb, "global": gb, "total": float(tot), "reported": float(rep),
                    "grads": [p.grad.clone() for p in m.ps],
                    "rows": batch["initial_inventories"][:, 0, 0].clone()})
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_epoch_with_uneven_batches_matches_single_process():
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    from golden_io import Golden
    from neural_inventory_control_amd.data_handling import DeviceBatches, MyDataset
    from oracle import inventory_oracle as orc
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_epoch_worker, args=(4, 29500 + (os.getpid() + 13) % 2000, ret), nprocs=4, join=True)
    g = Golden("cfg3_one_warehouse_16_vanilla")
    c = g.fresh_config()
    n = 22
    data = {k: v[:n] for k, v in g.data.items()}
    T, S = c["periods"], c["problem_params"]["n_stores"]
    single = DeviceBatches(MyDataset(n, data), 9, shuffle=True, device="cpu", seed=3)   # the same permutation, one process
    pol = orc.policy_from_state_dict(c["nn_params"], g.params, c["problem_params"], g.tensor("warehouse_upper_bound"))
    locals_seen = []
    for i, batch in enumerate(single):
        gb = len(batch["demands"])
        for p in pol.parameters():
            p.grad = None
        res = orc.rollout(pol, T, c["problem_params"], batch, c["observation_params"], c["ignore"])
        (res.total / (gb * T * S)).backward()
        per_rank = [ret[r][i] for r in range(4)]
        locals_seen.append([x["local"] for x in per_rank])
        assert all(x["global"] == gb for x in per_rank) and sum(x["local"] for x in per_rank) == gb
        # the ranks' slices, in rank order, ARE the global batch (every scenario exactly once, same order)
        assert torch.equal(torch.cat([x["rows"] for x in per_rank]), batch["initial_inventories"][:, 0, 0])
        for r in range(4):
            assert abs(per_rank[r]["total"] - float(res.total)) <= 2e-6 * abs(float(res.total))
            assert abs(per_rank[r]["reported"] - float(res.reported)) <= 2e-6 * abs(float(res.reported))
            for got, p in zip(per_rank[r]["grads"], pol.parameters()):
                assert float((got - p.grad).norm() / (p.grad.norm() + 1e-30)) < 5e-6
    assert locals_seen == [[3, 3, 3, 0], [3, 3, 3, 0], [1, 1, 1, 1]]   # an empty slice on rank 3, a short last batch


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` must start N ranks itself or fail loudly - never report a 1-rank number as N GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no bench line at all
