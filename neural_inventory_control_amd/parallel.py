"""Scenario-sharded data parallelism: one process per GPU, scenarios split in contiguous blocks, and exactly ONE
collective per optimizer step — a SUM all-reduce of a single flat fp32 buffer [grad_0 ... grad_n, total, reported]
over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; the only cross-scenario couplings of the path are the loss
sum (loss_functions.py:12), the parameter gradient, and the global demand mean used for initial inventories
(data_handling.py:298).  Message sizes are tiny (9 KB ... 3.3 MB), so the collective is latency-bound on xGMI: one
flat buffer, one launch, no bucketing.  Every rank divides by the GLOBAL B*T*S before backward, so the summed gradient
equals the single-GPU gradient up to fp32 summation order.
"""
import os

import torch
import torch.distributed as dist


def initialized():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if initialized() else 1


def rank():
    return dist.get_rank() if initialized() else 0


def init_from_env(backend=None):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 and not initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("NIC_DIST_BACKEND")  # e.g. gloo: two ranks sharing ONE GPU in tests
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, **kwargs)
    return rank(), world_size(), (torch.device("cuda", local) if use_cuda else torch.device("cpu"))


def shard_range(n_total, rank_, world):
    """Contiguous block [lo, hi) of scenarios owned by `rank_` (SURVEY §8e 'Partitioning')."""
    per = (n_total + world - 1) // world
    lo = min(rank_ * per, n_total)
    return lo, min(lo + per, n_total)


def all_reduce_scalars(*scalars):
    if world_size() == 1:
        return scalars
    buf = torch.stack([s.detach().float().reshape(()) for s in scalars])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return tuple(buf[i] for i in range(len(scalars)))


class GradientAllReducer:
    """Flat-buffer SUM all-reduce of every parameter gradient plus trailing scalars, once per optimizer step."""
    @classmethod
    def get(cls, model):
        """One reducer per model, stored ON the model (a table keyed by id(model) would outlive the model and could hand a
        new model at the same address another model's parameter list)."""
        r = model.__dict__.get("_nic_grad_reducer")
        if r is None:
            r = model.__dict__["_nic_grad_reducer"] = cls(model)
        return r

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = None

    def all_reduce(self, *scalars):
        params = self.params
        n = sum(p.numel() for p in params) + len(scalars)
        dev = params[0].device if params else scalars[0].device
        if self.flat is None or self.flat.numel() != n or self.flat.device != dev:
            self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            k = p.numel()
            if p.grad is None:
                self.flat[off:off + k].zero_()
            else:
                self.flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        for s in scalars:
            self.flat[off] = s.detach().float().reshape(())
            off += 1
        if world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        off = 0
        for p in params:
            k = p.numel()
            g = self.flat[off:off + k].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += k
        return tuple(self.flat[off + i].clone() for i in range(len(scalars)))
