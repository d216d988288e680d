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


def active():
    """True when the collectives of a sharded job must run: a process group exists.  (A group of ONE rank counts - that is
    how the RCCL code path is exercised on a single-GPU box, NIC_DIST_FORCE_INIT=1.)"""
    return initialized()


def init_from_env(backend=None):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world, device).
    NIC_DIST_FORCE_INIT=1 creates the process group even for WORLD_SIZE = 1, so that every collective of the sharded path
    (gradient all-reduce, parameter broadcast, global demand mean, barriers) really goes through RCCL with one rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get("NIC_DIST_FORCE_INIT") == "1") and not initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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


def device_identities(device):
    """[(rank, host, device uuid or PCI bus id)] of every rank, gathered over the process group: the evidence that an N-rank job
    ran on N DISTINCT devices (a launcher that maps two ranks to one GPU would otherwise still print a plausible number)."""
    import socket
    ident = "cpu"
    if device.type == "cuda":
        props = torch.cuda.get_device_properties(device)
        ident = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or device.index)
        if ident in ("None", ""):
            ident = f"cuda:{device.index}"
    mine = (rank(), socket.gethostname(), ident)
    if not active():
        return [mine]
    out = [None] * world_size()
    dist.all_gather_object(out, mine)
    return out


def shard_range(n_total, rank_, world):
    """Contiguous block [lo, hi) of scenarios owned by `rank_` (SURVEY §8e 'Partitioning')."""
    per = (n_total + world - 1) // world
    lo = min(rank_ * per, n_total)
    return lo, min(lo + per, n_total)


def _collective(fn, t):
    """Runs the in-place collective `fn(tensor)` on `t`, staging it where the backend can reach it: RCCL ("nccl") only
    moves device tensors, gloo only host tensors (the CPU tests; two test ranks sharing one GPU)."""
    backend = dist.get_backend()
    if backend == "gloo" and t.is_cuda:
        h = t.cpu()
        fn(h)
        t.copy_(h)
    elif backend == "nccl" and not t.is_cuda:
        d = t.cuda()
        fn(d)
        t.copy_(d)
    else:
        fn(t)
    return t


def all_reduce_sum(t):
    """SUM all-reduce of a small tensor over the ranks, in place on a contiguous copy (identity without a process group)."""
    if not active():
        return t
    t = t.contiguous()
    return _collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM), t)


def broadcast_model(model, src=0):
    """Every rank starts from rank `src`'s parameters, buffers and order upper bound (lazy layers must be materialised
    first).  Only gradients are all-reduced afterwards, so replicas that start equal stay equal; without this, equality
    would rest on every rank having drawn identical initial weights from identical RNG states."""
    if not active():
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    ub = getattr(model, "warehouse_upper_bound", None)
    if torch.is_tensor(ub):
        tensors.append(ub.data)
    for t in tensors:
        _collective(lambda x: dist.broadcast(x, src=src), t)


def parameters_in_sync(model):
    """True if every rank holds the same parameters: float64 (sum, sum of squares) per rank, MIN and MAX over ranks equal."""
    if not active():
        return True
    ps = [p.detach().double() for p in model.parameters()]
    mine = torch.stack([sum(p.sum() for p in ps), sum((p * p).sum() for p in ps)])
    lo, hi = mine.clone(), mine.clone()
    _collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN), lo)
    _collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.MAX), hi)
    return bool(torch.equal(lo.cpu(), hi.cpu()))


def all_reduce_scalars(*scalars):
    if not active():
        return scalars
    buf = torch.stack([s.detach().float().reshape(()) for s in scalars])
    buf = all_reduce_sum(buf)
    return tuple(buf[i] for i in range(len(scalars)))


class GradientAllReducer:
    """Flat-buffer SUM all-reduce of every parameter gradient plus trailing scalars, once per optimizer step.

    The flat buffer is carved into per-parameter views once; packing and unpacking are ONE multi-tensor copy each
    (`torch._foreach_copy_`: a fused launch on the device instead of one small copy per parameter on either side of a
    latency-bound collective).  Gradient tensors keep their identity (a captured training step accumulates into fixed
    `.grad` tensors; the fused engine hands out its own buffers), only their contents are replaced by the sum."""

    @classmethod
    def get(cls, model):
        """One reducer per model, stored ON the model (a table keyed by id(model) would outlive the model and could hand
        a new model at the same address another model's parameter list)."""
        r = model.__dict__.get("_nic_grad_reducer")
        if r is None:
            r = model.__dict__["_nic_grad_reducer"] = cls(model)
        return r

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat, self.views, self.n_scalars = None, None, 0
        self.timing = False      # bench.py: bracket the collective with device events (on the stream it is enqueued behind)
        self._events, self.calls = [], 0

    def collective_stats(self):
        """What the per-step collective was, for the bench line: bytes of the flat buffer, how many all-reduces ran and their mean
        duration (events recorded on torch's current stream right around the call; call after a synchronize)."""
        ms = [a.elapsed_time(b) for a, b in self._events]
        return {"allreduce_bytes": int(self.flat.numel() * 4) if self.flat is not None else 0, "allreduce_calls": self.calls,
                "allreduce_ms": (sum(ms) / len(ms)) if ms else None}

    def _layout(self, n_scalars, device):
        n = sum(p.numel() for p in self.params)
        if self.flat is None or self.flat.numel() != n + n_scalars or self.flat.device != device:
            self.flat = torch.zeros(n + n_scalars, device=device, dtype=torch.float32)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.tail = self.flat[n:]

    def all_reduce(self, *scalars):
        params = self.params
        dev = params[0].device if params else scalars[0].device
        self._layout(len(scalars), dev)
        src, dst = [], []
        for p, v in zip(params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad.detach() if p.grad.dtype == torch.float32 else p.grad.detach().float())
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        if scalars:   # (one launch: the stack writes straight into the buffer's tail)
            torch.stack([s.detach().float().reshape(()) for s in scalars], out=self.tail)
        if active():
            ev = None
            if self.timing and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM), self.flat)
            self.calls += 1
            if ev is not None:
                ev[1].record()
                self._events.append(ev)
        back_dst, back_src = [], []
        for p, v in zip(params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            elif p.grad.data_ptr() != v.data_ptr():
                back_dst.append(p.grad)
                back_src.append(v)
        if back_dst:
            torch._foreach_copy_(back_dst, back_src)
        out = self.tail.clone()   # (the caller's scalars must not change under the next step's collective)
        return tuple(out[i] for i in range(len(scalars)))
