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
