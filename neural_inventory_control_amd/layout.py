"""Scenario-minor (SoA) memory layout helpers.

The reference keeps per-scenario quantities scenario-major, e.g. store pipelines as (B, S, Ws) with the slot index
contiguous (data_handling.py:305, environment.py:184-185).  On MI355X a wavefront is 64 lanes and the natural unit
of parallelism is the scenario, so every device buffer here is stored [rows...][ldb] with the SCENARIO index
contiguous (ldb = B rounded up to 64): lane i of a wave reads scenario b0+i and each load instruction is one
coalesced 256-byte access.  The reference's logical shapes are exposed as strided views (`ref_view`), so user
policies written against the reference's shapes keep working unchanged.

Pure tensor plumbing (torch ops + pointer arithmetic); no arithmetic of the hot path lives here.
"""
import torch

from . import _lib
from ._lib import NicEnvDims, NicEnvStepIO, NicTable2, NicTable3

WAVE = 64


def pad_ld(n, mult=WAVE):
    return ((int(n) + mult - 1) // mult) * mult


def to_soa(t, ldb=None):
    """(B, d1, d2, ...) -> contiguous [d1][d2]...[ldb] float32, zero in the padding columns."""
    B = t.shape[0]
    ldb = ldb or pad_ld(B)
    out = torch.zeros(tuple(t.shape[1:]) + (ldb,), dtype=torch.float32, device=t.device)
    out[..., :B] = t.movedim(0, -1)
    return out


def ref_view(soa, B):
    """[d1]...[ldb] storage -> logical (B, d1, ...) strided view (no copy)."""
    return soa[..., :B].movedim(-1, 0)


def is_soa_view(t):
    """True if `t` (logical (B, d1, ..)) is a view whose batch dim is the contiguous one (stride 1)."""
    return t.dim() >= 1 and t.stride(0) == 1 and all(s >= t.shape[0] for s in t.stride()[1:])


def demand_trace_soa(d, ldb, device=None):
    """`data["demands"]` (B, S, T) -> the kernels' [T][S][ldb] trace.  A batch that already IS such a trace seen through
    `ref_view` - what `Scenario(sampler="hip")` hands out, and what a captured step's static copy of it keeps - is returned as a
    view of its own storage when the batch fills its padded row (B == ldb: no padding lanes of unknown content; no zero-fill, no copy: at 10^6 chains x T=100 those were 0.22 ms of a 0.47 ms step); anything else
    is transposed into a fresh zero-padded buffer."""
    B, S, T = d.shape
    device = torch.device(device) if device is not None else d.device
    same_device = d.device.type == device.type and (device.index is None or d.device.index == device.index)   # ("cuda" = any index)
    st = d.stride()   # (the stride of a size-1 dimension is arbitrary)
    if (d.dtype == torch.float32 and same_device and B == ldb and B > 1 and st[0] == 1 and (S == 1 or st[1] == ldb)
            and (T == 1 or st[2] == S * ldb) and d.storage_offset() % 4 == 0
            and d.untyped_storage().nbytes() >= 4 * (d.storage_offset() + T * S * ldb)):
        return d.as_strided((T, S, ldb), (S * ldb, ldb, 1), d.storage_offset())
    out = torch.zeros(T, S, ldb, device=device)
    out[:, :, :B] = d.permute(2, 1, 0)
    return out


class Table:
    """A per-location table (loc[, supplier], scenario) with explicit element strides; keeps the tensor alive."""

    def __init__(self, tensor, loc_stride, scn_stride, sup_stride=0):
        self.tensor, self.loc_stride, self.scn_stride, self.sup_stride = tensor, loc_stride, scn_stride, sup_stride

    @staticmethod
    def null():
        return Table(None, 0, 0)

    @staticmethod
    def from_ref(t, ldb, device=None, detect_uniform=True):
        """(B, L) or (B, L, P) reference tensor -> compact device table.  Batch-broadcast inputs (`expand` views with
        stride 0 on the batch dim, data_handling.py:257-269) are stored once ([L] or [L][P]) with scn_stride 0."""
        if t is None:
            return Table.null()
        device = device or t.device
        uniform = t.stride(0) == 0 or t.shape[0] == 1
        if not uniform and detect_uniform:
            # collated batches materialise the reference's expand() views; one comparison per reset recovers them
            uniform = bool((t[1:] == t[:1]).all())
        if uniform:
            row = t[0].to(device=device, dtype=torch.float32).contiguous()
            if row.dim() == 1:
                return Table(row, 1, 0)
            return Table(row, row.stride(0), 0, row.stride(1))
        soa = to_soa(t.to(device), ldb)
        if soa.dim() == 2:
            return Table(soa, ldb, 1)
        return Table(soa, soa.stride(0), 1, soa.stride(1))

    @staticmethod
    def from_orders(t):
        """An action tensor exactly as the policy produced it: logical (B, L, P) with arbitrary strides."""
        if t.dtype != torch.float32:
            t = t.float()
        if t.dim() == 2:
            return Table(t, t.stride(1), t.stride(0))
        return Table(t, t.stride(1), t.stride(0), t.stride(2))

    def t2(self):
        return NicTable2(_lib.ptr(self.tensor), self.loc_stride, self.scn_stride)

    def t3(self):
        return NicTable3(_lib.ptr(self.tensor), self.loc_stride, self.sup_stride, self.scn_stride)


class EnvProblem:
    """Static description of one rollout (what Simulator.reset receives): sizes + cost / lead-time tables."""

    def __init__(self, problem_params, data, device):
        inv = data["initial_inventories"]
        self.B = int(inv.shape[0])
        self.ldb = pad_ld(self.B)
        self.S = int(problem_params["n_stores"])
        self.Wn = int(problem_params["n_warehouses"])
        self.E = int(problem_params["n_extra_echelons"])
        self.nsup = max(self.Wn, 1)
        self.Ws = int(inv.shape[2])
        self.Ww = int(data["initial_warehouse_inventories"].shape[2]) if self.Wn > 0 else 0
        self.We = int(data["initial_echelon_inventories"].shape[2]) if self.E > 0 else 0
        for name, w in (("store", self.Ws), ("warehouse", self.Ww), ("echelon", self.We)):
            if w == 1 or w > _lib.NIC_MAX_SLOTS:
                raise ValueError(f"{name} pipeline length {w} unsupported (need 2..{_lib.NIC_MAX_SLOTS}); "
                                 "the reference itself needs >= 2 slots (environment.py:407)")
        self.lost_demand = bool(problem_params["lost_demand"])
        self.maximize_profit = bool(problem_params["maximize_profit"])
        self.device = device
        ld = self.ldb
        self.underage = Table.from_ref(data["underage_costs"], ld, device)
        self.holding = Table.from_ref(data["holding_costs"], ld, device)
        self.lead = Table.from_ref(data["lead_times"], ld, device)
        self.wh_holding = Table.from_ref(data.get("warehouse_holding_costs") if self.Wn else None, ld, device)
        self.wh_lead = Table.from_ref(data.get("warehouse_lead_times") if self.Wn else None, ld, device)
        self.wh_edge = Table.from_ref(data.get("warehouse_edge_costs") if self.Wn else None, ld, device)
        self.ech_holding = Table.from_ref(data.get("echelon_holding_costs") if self.E else None, ld, device)
        self.ech_lead = Table.from_ref(data.get("echelon_lead_times") if self.E else None, ld, device)

    _TABLES = ("underage", "holding", "lead", "wh_holding", "wh_lead", "wh_edge", "ech_holding", "ech_lead")

    def same_layout(self, other):
        """True if `other` describes the same sizes and every table has the same shape / strides (so a captured launch
        sequence that points at this object's tables stays valid once their contents are refreshed)."""
        if (self.B, self.S, self.Wn, self.E, self.Ws, self.Ww, self.We, self.lost_demand, self.maximize_profit) != \
                (other.B, other.S, other.Wn, other.E, other.Ws, other.Ww, other.We, other.lost_demand, other.maximize_profit):
            return False
        for name in self._TABLES:
            a, b = getattr(self, name), getattr(other, name)
            if (a.tensor is None) != (b.tensor is None):
                return False
            if a.tensor is not None and (a.tensor.shape != b.tensor.shape or
                                         (a.loc_stride, a.scn_stride, a.sup_stride) != (b.loc_stride, b.scn_stride, b.sup_stride)):
                return False
        return True

    def copy_tables_from(self, other):
        for name in self._TABLES:
            a, b = getattr(self, name), getattr(other, name)
            if a.tensor is not None:
                a.tensor.copy_(b.tensor)

    def dims(self):
        return NicEnvDims(self.B, self.ldb, self.S, self.Wn, self.E, self.Ws, self.Ww, self.We,
                          int(self.lost_demand), int(self.maximize_profit))

    def make_io(self, store_inv, wh_inv, ech_inv, demand, store_orders, wh_orders, ech_orders):
        """store_inv/wh_inv/ech_inv: SoA tensors; demand + orders: `Table`s."""
        io = NicEnvStepIO()
        io.dims = self.dims()
        io.store_inv, io.wh_inv, io.ech_inv = _lib.ptr(store_inv), _lib.ptr(wh_inv), _lib.ptr(ech_inv)
        io.demand = demand.t2()
        io.store_orders = store_orders.t3()
        io.wh_orders = (wh_orders or Table.null()).t2()
        io.ech_orders = (ech_orders or Table.null()).t2()
        io.underage, io.holding, io.lead_times = self.underage.t2(), self.holding.t2(), self.lead.t3()
        io.wh_holding, io.wh_lead_times, io.wh_edge_costs = self.wh_holding.t2(), self.wh_lead.t2(), self.wh_edge.t2()
        io.ech_holding, io.ech_lead_times = self.ech_holding.t2(), self.ech_lead.t2()
        return io


class ProblemCache:
    """`EnvProblem`s of recently seen batches.  Building one compacts the static tables and checks them for
    scenario-uniformity, which reads a flag back from the device (a sync); a batch that presents the SAME tensors again
    (same storage, shape, strides and in-place version - every step of a benchmark, the fixed batches of an un-shuffled
    loader, the static inputs of a captured training step) reuses its entry, so a rollout contains no host sync.

    Each entry keeps STRONG references to the tensors its key was computed from: as long as the entry lives their storage
    cannot be freed and handed to a later batch with the same address and version (which would be a false hit running the
    step with another batch's cost / lead-time tables)."""

    STATIC_KEYS = ("underage_costs", "holding_costs", "lead_times", "warehouse_holding_costs", "warehouse_lead_times",
                   "warehouse_edge_costs", "echelon_holding_costs", "echelon_lead_times", "initial_inventories",
                   "initial_warehouse_inventories", "initial_echelon_inventories")

    def __init__(self, capacity=16):
        self.capacity = capacity
        self._entries = {}

    def get(self, problem_params, data, device):
        tensors = [(k, data[k]) for k in self.STATIC_KEYS if data.get(k) is not None]
        # the initial pipelines only contribute their SHAPES to an EnvProblem (their values are re-read every reset)
        key = (id(problem_params), bool(problem_params["lost_demand"]), bool(problem_params["maximize_profit"])) + tuple(
            (k, tuple(t.shape)) if k.startswith("initial_") else (k, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version)
            for k, t in tensors)
        hit = self._entries.get(key)
        if hit is None:
            if len(self._entries) >= self.capacity:
                self._entries.pop(next(iter(self._entries)))
            prob = EnvProblem(problem_params, data, device)
            hit = self._entries[key] = (prob, [t for k, t in tensors if not k.startswith("initial_")], problem_params)
        return hit[0]
