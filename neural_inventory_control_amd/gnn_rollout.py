"""`GnnRollout`: the unrolled rollout + backward of `Trainer.simulate_batch` for the GNN policy (`gnn.yml`;
neural_networks.py:742-1492 of the reference) on the warehouse -> store settings (one or several warehouses),
without an autograd graph.

The supply graph is static, so it is compiled once into entity maps and CSR lists (`GraphPlan`).  Per period the engine
launches five fused gather-MLP kernels (`nic_mlp3_fwd`, csrc/mlp3.hip: each evaluates one of the policy's 32-wide MLPs for every
(node or edge, scenario) column, gathering its concatenated input rows straight from the node / edge buffers through the maps),
two segment sums for the message aggregation, a few small tensor ops for the proportional allocation, and the env step.  The
backward sweep mirrors it: env backward, allocation adjoint, `nic_mlp3_bwd` per MLP, segment sums over the TRANSPOSED maps as
the adjoint of every gather (deterministic, no atomics); weight gradients are contractions of the stored pre-activation
gradients with the stored inputs (`nic_linear_wgrad`, slabs accumulated over the periods).  The reference's Python loops over
edges (:1229-1269) and nodes (:1474-1490) become launches over all scenarios at once; ~60 launches per period instead of ~380.

Everything is feature-major `[rows][entity][ldb]`.  Nodes = [warehouses..., stores...]; edges = [internal (warehouse -> store,
warehouse-major), supplier edges of the warehouses, self loops of the supplying warehouses, demand edges of the stores].  That is
the reference's order with the demand edges moved to the end: the edge-update and output MLPs skip them (nothing reads what
they would compute there), and every sum over a node's edges still adds them in the reference's order (`GraphPlan`).
"""
import torch

from . import _lib, ops
from .layout import demand_trace_soa, ProblemCache, Table
from .ops import EnvState, Mlp3Segment

MODULES = ("initial_node", "initial_edge", "node_update", "edge_update", "output")


def _csr(lists, device):
    offs, items = [0], []
    for l in lists:
        items += l
        offs.append(len(items))
    return (torch.tensor(offs, dtype=torch.int32, device=device),
            torch.tensor(items if items else [0], dtype=torch.int32, device=device))


class GraphPlan:
    """Static structure of the warehouse -> store supply graph (neural_networks.py:757-1062) as device index tensors.
    `conn` [W][S] = the setting's `warehouse_store_adjacency` (one warehouse: all ones).  Nodes = [warehouses..., stores...],
    internal edges warehouse-major (the order of `adjacency.nonzero()`), then one supplier edge per warehouse, one self loop per
    warehouse that supplies somebody (none under transshipment), one demand edge per store (see __init__ on that order)."""

    def __init__(self, S, conn, transshipment, device):
        Wn = len(conn)
        self.S, self.Wn, self.n_nodes = S, Wn, Wn + S
        internal = [(w, Wn + s) for w in range(Wn) for s in range(S) if conn[w][s]]
        n_int, n_sup, n_dem = len(internal), Wn, S
        self.n_int = n_int
        out_int = [sum(1 for a, _ in internal if a == w) for w in range(Wn)]
        supplying = [] if transshipment else [w for w in range(Wn) if out_int[w] > 0]
        n_self = self.n_self = len(supplying)
        E = self.n_edges = n_int + n_sup + n_dem + n_self
        # The reference's edge order is [internal, supplier, demand, self loops].  The engine keeps the demand edges LAST: what
        # the edge-update and output MLPs compute for them is read by nothing (one message-passing step: their updated embedding
        # only feeds the output MLP, and a demand edge's output is no action), so those two MLPs - forward and backward - run on
        # the first `n_live` edges only.  Every sum over a node's edges still adds them in the REFERENCE's order (the lists below
        # are built in that order and then renamed).
        r_src = [a for a, _ in internal] + [-1] * n_sup + [Wn + s for s in range(S)] + supplying
        r_tgt = [b for _, b in internal] + list(range(Wn)) + [-1] * n_dem + supplying
        r_demand = set(range(n_int + n_sup, n_int + n_sup + n_dem))
        r_supplier = set(range(n_int, n_int + n_sup))
        eng = list(range(n_int + n_sup)) + [n_int + n_sup + n_self + s for s in range(n_dem)] + \
              [n_int + n_sup + k for k in range(n_self)]                   # reference edge id -> engine edge id
        ref = [0] * E
        for r_, e_ in enumerate(eng):
            ref[e_] = r_
        src, tgt = [r_src[ref[e]] for e in range(E)], [r_tgt[ref[e]] for e in range(E)]
        self.n_live = n_int + n_sup + n_self
        self.e_supplier = n_int                                            # (of warehouse 0; warehouse w: n_int + w)
        self.e_self = n_int + n_sup if n_self else None                    # (of the first supplying warehouse)
        self.e_demand = self.n_live
        in_deg, out_deg = [0] * self.n_nodes, [0] * self.n_nodes
        for a, b in internal:
            out_deg[a] += 1
            in_deg[b] += 1
        for w in range(Wn):
            in_deg[w] += 1                  # supplier edge into every warehouse
        for s in range(S):
            out_deg[Wn + s] += 1            # demand edge out of every store
        for n in supplying:
            in_deg[n] += 1
            out_deg[n] += 1
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)  # noqa: E731
        self.src, self.tgt = i32(src), i32(tgt)
        # message aggregation (:1229-1296): a node sums the edges it is the target of (demand edges have no real target) /
        # the source of (supplier edges have no real source), in (reference) edge order, then divides by sqrt(degree)
        inc = [[eng[r] for r in range(E) if r_tgt[r] == n and r not in r_demand] for n in range(self.n_nodes)]
        out = [[eng[r] for r in range(E) if r_src[r] == n and r not in r_supplier] for n in range(self.n_nodes)]
        # both aggregations as ONE segment sum into a [rows][2 * n_nodes][ldb] buffer (incoming sums first, then outgoing)
        self.agg_off, self.agg_items = _csr(inc + out, device)
        self.n_agg_items = sum(len(l) for l in inc + out)
        self.in_scale = torch.tensor([1.0 / max(d, 1) ** 0.5 for d in in_deg], device=device)
        self.out_scale = torch.tensor([1.0 / max(d, 1) ** 0.5 for d in out_deg], device=device)
        self.agg_scale = torch.cat([self.in_scale, self.out_scale])
        # adjoints of the aggregation: edge e receives from its target's `incoming` / its source's `outgoing` gradient
        is_demand = [ref[e] in r_demand for e in range(E)]
        is_supplier = [ref[e] in r_supplier for e in range(E)]
        self.e_from_tgt = _csr([[tgt[e]] if (tgt[e] >= 0 and not is_demand[e]) else [] for e in range(E)], device)
        self.e_from_src = _csr([[src[e]] if (src[e] >= 0 and not is_supplier[e]) else [] for e in range(E)], device)
        self.e_in_scale = torch.tensor([self.in_scale[tgt[e]].item() if tgt[e] >= 0 else 0.0 for e in range(E)], device=device)
        self.e_out_scale = torch.tensor([self.out_scale[src[e]].item() if src[e] >= 0 else 0.0 for e in range(E)], device=device)
        # adjoints of the endpoint gathers: node n receives from the edges it is the source / target of
        self.n_as_src = _csr([[eng[r] for r in range(E) if r_src[r] == n] for n in range(self.n_nodes)], device)
        self.n_as_tgt = _csr([[eng[r] for r in range(E) if r_tgt[r] == n] for n in range(self.n_nodes)], device)
        # ... of the edge-update MLP, which leaves the demand edges out: their input-gradient rows are exact zeros upstream
        self.n_as_src_live = _csr([[eng[r] for r in range(E) if r_src[r] == n and r not in r_demand]
                                   for n in range(self.n_nodes)], device)
        self.n_as_tgt_live = _csr([[eng[r] for r in range(E) if r_tgt[r] == n and r not in r_demand]
                                   for n in range(self.n_nodes)], device)
        # (lengths of the four lists above in the order nic_gnn_period_bwd takes them: source / target over live edges, over all)
        self.list_len = [sum(1 for r in range(E) if r_src[r] >= 0 and r not in r_demand),
                         sum(1 for r in range(E) if r_tgt[r] >= 0 and r not in r_demand),
                         sum(1 for r in range(E) if r_src[r] >= 0), sum(1 for r in range(E) if r_tgt[r] >= 0)]
        # per-edge constant input row: lead time of internal / supplier edges, 0 for self loops and demand edges; refreshed on
        # the device from every batch's lead-time tensors through these two index lists (GnnRollout.run)
        self.lead = torch.zeros(1, E, dtype=torch.float32, device=device)
        self.lead_store = torch.tensor([b - Wn for _, b in internal], dtype=torch.long, device=device)
        self.lead_wh = torch.tensor([a for a, _ in internal], dtype=torch.long, device=device)
        # proportional allocation groups (:1435-1492): a warehouse's internal edges (contiguous) + its self loop
        first, groups = 0, []
        for w in range(Wn):
            e_self = n_int + n_sup + supplying.index(w) if w in supplying else -1
            groups.append([first, out_int[w], e_self, n_int + w])
            first += out_int[w]
        self.groups = i32(groups)
        # where an edge's quantity lands in the orders buffer [S][Wn] + [Wn].  Upstream writes store s's j-th INCOMING edge into
        # column j (:1423-1428) - for a store that is not connected to every warehouse this is not the column of the warehouse
        # the edge comes from (the env step reads lead time and shipment source by column).  Reproduced as is: it is what the
        # reference computes and what the fixtures pin; `misplaced` lists the (store, column, warehouse) triples it affects.
        order_row, seen, self.misplaced = [-1] * E, [0] * S, []
        for e, (a, b) in enumerate(internal):
            s_ = b - Wn
            order_row[e] = s_ * Wn + seen[s_]
            if seen[s_] != a:
                self.misplaced.append((s_, seen[s_], a))
            seen[s_] += 1
        for w in range(Wn):
            order_row[n_int + w] = S * Wn + w
        self.order_row = i32(order_row)
        self.transshipment = bool(transshipment)


class _Mlp:
    """Packed weights, gradient slabs and per-period history of one of the policy's five MLPs."""

    def __init__(self, name, linears, K, n_out, out_act, n_ent, ld, T, P, device, train, mode="hist", keep_inputs=True,
                 n_live=None, dense=False):
        self.name, self.linears, self.K, self.n_out, self.out_act, self.n_ent = name, linears, K, n_out, out_act, n_ent
        self.n_live = n_ent if n_live is None else n_live   # entities the MLP is evaluated for (the first n_live of its buffers)
        z = lambda *s: torch.zeros(*s, device=device)  # noqa: E731
        self.packed = z(32 * K + 32 + 32 * 32 + 32 + n_out * 32 + n_out)
        if dense:   # every column is a scenario: the forward writes all of what the backward reads - no 10-GB zero fill at set-up
            z = lambda *s: torch.empty(*s, device=device)  # noqa: E731
        self.Y = z(T if train else 1, n_out, n_ent, ld)   # (evaluation: one period's worth, reused - nothing reads an earlier one)
        self.P, self.G = P, (T + P - 1) // P
        self.hist_stride = 0
        self.native = False
        self.mode = mode if train else None     # "hist" | "fused" (see GnnRollout.fused_bwd)
        self.fused_bwd = mode == "fused"
        dims = [(32, K), (32, 32), (n_out, 32)]
        if train and mode == "hist":
            # stored hidden activations, weight gradients contracted inside the backward kernel (nic_mlp3_bwd_hist), which reads
            # the MLP inputs again from their per-period source buffers (or, keep_inputs, from a stored copy: same speed, 40 %
            # more history).  One slab slot per workgroup, accumulated over the periods
            self.hist_stride = P * n_ent * ld
            G = self.G
            self.X = z(G, K, P, n_ent, ld) if keep_inputs else None
            # without the stored inputs the hidden activations are private to the forward / backward kernels of this MLP: kept
            # in the kernels' native order, [period][entity x chunk][32 rows][32 scenarios] (one 4-KB block per wavefront access
            # instead of 32 pieces of 128 B spread over the buffer); same size
            self.native = not keep_inputs
            self.H1, self.H2 = (z(G, P, 32 * n_ent * ld), z(G, P, 32 * n_ent * ld)) if self.native else \
                               (z(G, 32, P, n_ent, ld), z(G, 32, P, n_ent, ld))
            self.slabs = [torch.zeros(ops.mlp3_bwd_hist_slots(), n, (k + 1 + 3) // 4 * 4, device=device) for n, k in dims]
            self.dX = torch.zeros(K, n_ent, ld, device=device)
            self.gw = [torch.zeros_like(m.weight) for m in linears]
            self.gb = [torch.zeros_like(m.bias) for m in linears]
        elif train and mode == "fused":
            # history-free backward (nic_mlp3_bwd_fused): the kernel re-gathers the inputs, recomputes the hidden layers and
            # contracts the weight gradients itself; one slab slot per wavefront, accumulated over the periods
            slots = ops.mlp3_bwd_fused_slots()
            self.slabs = [torch.zeros(slots, n, (k + 1 + 3) // 4 * 4, device=device) for n, k in dims]
            self.dX = torch.zeros(K, n_ent, ld, device=device)
            self.gw = [torch.zeros_like(m.weight) for m in linears]
            self.gb = [torch.zeros_like(m.bias) for m in linears]

    def hist(self, buf, t):
        """[rows][entity][ldb] view of period t inside a grouped history buffer (row stride = hist_stride); the native hidden
        histories: period t's block."""
        if self.native and buf is not self.X:
            return buf[t // self.P, t % self.P]
        return buf[t // self.P, :, t % self.P]

    def pack(self):
        self.packed.copy_(torch.cat([t.detach().reshape(-1) for m in self.linears for t in (m.weight, m.bias)]))
        tr = ops.mlp3_pack_transposed([(m.weight, m.bias) for m in self.linears], self.n_out)
        if getattr(self, "packed_t", None) is None:
            self.packed_t = tr.clone()
        else:
            self.packed_t.copy_(tr)


class GnnRollout:
    @staticmethod
    def supports(model, problem_params=None):
        if type(model).__name__ != "GNN" or not hasattr(model, "nn_args"):
            return False
        a = model.nn_args
        ok = all(a["inner_layer_activations"][m] == "elu" and list(a["neurons_per_hidden_layer"][m]) == [32, 32]
                 for m in MODULES)
        ok = ok and all(a["output_layer_activation"][m] == "elu" and a["output_sizes"][m] == 32 for m in MODULES[:-1])
        ok = ok and a["output_layer_activation"]["output"] == "softplus" and a["output_sizes"]["output"] == 1
        if problem_params is not None:
            ok = ok and problem_params["n_warehouses"] >= 1 and problem_params["n_extra_echelons"] == 0
            if ok and problem_params["n_warehouses"] > 1:
                # upstream's action tensor has as many columns as the best-connected store has warehouses; the env step needs
                # one per warehouse (otherwise the reference itself raises): some store must see every warehouse
                conn = problem_params.get("warehouse_store_adjacency")
                ok = conn is not None and max(sum(int(bool(conn[w][s_])) for w in range(len(conn)))
                                              for s_ in range(problem_params["n_stores"])) == problem_params["n_warehouses"]
        return bool(ok)

    def __init__(self, model, problem_params, device):
        _lib.require_device()
        if not self.supports(model, problem_params):
            raise ValueError("GnnRollout handles the gnn.yml architecture on warehouse -> store settings (no extra echelons)")
        self.model, self.problem_params, self.device = model, problem_params, torch.device(device)
        self.timer = None
        # Backward of the MLPs.  None = "hist" while the stored activations fit in HBM, else the history-free kernel.
        #   "hist"  stored inputs / hidden activations, weight gradients contracted inside the backward kernel (nic_mlp3_bwd_hist)
        #           (the round-2 form that also stored the pre-activation gradients for separate nic_linear_wgrad contractions -
        #           `fused_bwd = False` - was removed in round 6: it lost to "hist" by 10 % and was a test mode only)
        #   True    ("fused") nic_mlp3_bwd_fused: re-gather, recompute, in-kernel weight gradients; no per-period buffers at
        #           all, but its gathers are latency-exposed at two wavefronts per SIMD
        self.fused_bwd = None
        self.keep_inputs = False  # "hist": True = also keep a copy of the gathered MLP inputs (while it fits) instead of reading
        #                         them again from their per-period sources in the backward (same speed, 40 % more history)
        # replay the (static) launch sequence of a rollout from a HIP graph after one eager run: True / False, or "auto" (what
        # `Trainer` sets, as on the MLP engine) = decided by MEASUREMENT on the second training run of a shape - if the host needs
        # longer to enqueue the ~27 launches per period than the GPU needs to run them, later runs are replayed (the reference's
        # shipped batch of 1,024 scenarios: 19.2 -> 14.6 ms per step; 8,192 scenarios are GPU-bound and stay eager)
        self.use_graph = False
        self.fuse_alloc_env = True   # one-warehouse graphs: allocation head + env step (and their adjoints) in one launch each
        # Forward of a period as ONE launch (csrc/gnn_period.hip: the five MLPs on embeddings held in LDS, + allocation and env step
        # on one-warehouse graphs) instead of ~8: "auto" = wherever the graph's embeddings fit in LDS and the backward reads the
        # native histories (or nothing); True raises where that does not hold; False = the per-MLP launches
        self.use_period_kernel = "auto"
        # Backward of a period's five MLPs + every adjoint gather / aggregation + the row adds into the state gradient as ONE launch
        # (csrc/gnn_period_bwd.hip, round 6) instead of five `nic_mlp3_bwd_hist` + three `nic_segment_sum_terms` launches + two
        # tensor ops: "auto" = wherever the backward reads native histories ("hist" without the stored input copy) and the node
        # features have at most 32 rows; True raises where that does not hold; False = the per-MLP launches.  Any graph size.
        self.use_period_bwd = "auto"
        # What happens to a non-zero store order booked on a column whose lead time is 0 (only the GNN on a sparse many-warehouse
        # graph produces such orders: upstream writes store s's j-th CONNECTED edge into action column j, neural_networks.py:1423-1428).
        #   "drop"      (default) the HIP env step discards it (it still leaves the warehouse): scenarios stay independent, which
        #               is what scenario sharding needs
        #   "upstream"  the reference's own result: its flat-index put (environment.py:415-432) adds the order to the element IN
        #               FRONT of the store's pipeline - the last slot of the previous store, for store 0 of the previous SCENARIO,
        #               for scenario 0 of the batch's last scenario - reproduced here as a fix-up after the env step and its adjoint
        #               in the sweep, so that this engine matches the reference's golden numbers on such graphs (single process only:
        #               the coupling crosses shard boundaries)
        self.zero_lead_orders = "drop"
        self._auto_graph = None
        self.auto_graph_probe = None
        self._probs = ProblemCache()
        self._key = None

    def _k(self, tag, fn, *a, **kw):
        return fn(*a, **kw) if self.timer is None else self.timer.call(tag, fn, *a, **kw)

    def _graph_on(self):
        return self.use_graph is True or (self.use_graph == "auto" and self._auto_graph is True)

    # ---- setup --------------------------------------------------------------------------------------------------------------
    def shapes_ok(self, data):
        return "mean" in data and "std" in data and data["lead_times"].shape[2] == self.problem_params["n_warehouses"]

    def _linears(self, name):
        return [m for m in self.model.net[name] if isinstance(m, torch.nn.Linear)]

    def materialize(self, Dn):
        """Creates the lazy first layers without a forward pass (same default init as the reference's first call)."""
        from .rollout import FusedRollout
        for name, k in zip(MODULES, (Dn, 65, 96, 96, 32)):
            shim = type("S", (), {"master_linears": lambda s, n=name: self._linears(n)})()
            FusedRollout.materialize(type("E", (), {"model": shim})(), k)

    def _setup(self, prob, data, T, train):
        key = (prob.B, T, bool(train), prob.S, prob.Wn, prob.Ws, prob.Ww, self.fused_bwd, self.keep_inputs,
               data.get("warehouse_edge_costs") is not None, self.use_period_kernel, self.use_period_bwd)
        if key == self._key:
            return
        dev, ld, S = self.device, prob.ldb, prob.S
        self.mlp = None  # release the previous shapes' buffers before sizing the new ones
        if ld % 32:
            raise ValueError("ldb must be a multiple of 32")
        Wn = prob.Wn
        conn = self.problem_params.get("warehouse_store_adjacency") if Wn > 1 else [[1] * S]
        self.plan = GraphPlan(S, conn, getattr(self.model, "transshipment", False), dev)
        P = self.plan
        self.max_inv = max(prob.Ws, prob.Ww)
        self.has_edge_cost = data.get("warehouse_edge_costs") is not None
        self.max_st = max(4, 2 if self.has_edge_cost else 1)
        self.Dn = self.max_inv + self.max_st
        self.materialize(self.Dn)
        z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
        N, E = P.n_nodes, P.n_edges
        self.F_store, self.F_wh = S * prob.Ws, Wn * prob.Ww
        f_tot = self.F_store + self.F_wh
        self.states = z(T + 1, f_tot, ld)
        self.orders = z(T, S * Wn + Wn, ld)   # store orders [S][Wn] (columns a store has no edge for stay 0), warehouse orders [Wn]
        self.rewards = z(T, ld)
        Tp = T if train else 1   # periods of the policy's intermediate buffers kept: all for the backward sweep, one for an evaluation
        self._Tp = Tp           # (at the reference's test horizon, T = 5,000, the per-period copies were 83 GB for the outputs alone)
        self.feat = z(Tp, self.Dn, N, ld)
        # state row -> feature row (slot k of store s -> row k of node Wn + s, slot k of warehouse w -> row k of node w): the
        # pipeline part of the node features is ONE index_copy_ per period
        rows = [k * N + Wn + s_ for s_ in range(S) for k in range(prob.Ws)] + [k * N + w for w in range(Wn) for k in range(prob.Ww)]
        self.feat_rows = torch.tensor(rows, dtype=torch.long, device=dev)
        A = _lib
        ks = (self.Dn, 65, 96, 96, 32)
        ents = (N, E, N, E, E)
        P_ = max(1, min(T, (1 << 28) // (96 * E * ld)))   # periods per history row (see _Mlp)
        if self.fused_bwd is False or self.fused_bwd == "gemm":
            raise ValueError("fused_bwd = False (stored pre-activation gradients + GEMM contractions) was removed in round 6: use 'hist'")
        mode = {None: None, True: "fused", "hist": "hist", "fused": "fused"}[self.fused_bwd]
        self._keep_inputs = True
        if mode is None and train:
            # rows of history per period: H1 + H2 of the five MLPs, and their gathered inputs
            h_bytes = 4 * T * ld * 64 * (2 * N + 3 * E)
            x_bytes = 4 * T * ld * (N * (self.Dn + 96) + E * (65 + 96 + 32))
            free = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                    if dev.type == "cuda" else 0)
            mode = "fused" if h_bytes > 0.6 * free else "hist"
            self._keep_inputs = h_bytes + x_bytes <= 0.6 * free
        self._mode_now = mode or "hist"
        self._fused_bwd_now = self._mode_now == "fused"
        native_hist = self._mode_now == "hist" and not (self._keep_inputs and bool(self.keep_inputs))
        will_period = bool(self.use_period_kernel) and ops.gnn_period_ok(N, E, self.Dn) and \
            ((not train) or self._mode_now == "fused" or native_hist)
        dense = will_period and train and prob.B == ld and native_hist
        self.mlp = {name: _Mlp(name, self._linears(name), k, 1 if name == "output" else 32,
                               A.NIC_MLP3_ACT_SOFTPLUS if name == "output" else A.NIC_MLP3_ACT_ELU, ne, ld, T, P_, dev, train,
                               self._mode_now, self._keep_inputs and bool(self.keep_inputs),
                               n_live=P.n_live if name in ("edge_update", "output") else None,
                               dense=dense)
                    for name, k, ne in zip(MODULES, ks, ents)}
        self._graphs, self._eager_runs, self._auto_graph = {}, 0, None   # (a new shape is measured afresh)
        self._period, self._pdesc, self.edge_scratch = False, {}, None
        self._slab_flat = None
        if train:   # the fifteen weight-gradient slabs as views of one allocation: one fill per step zeroes them all
            sizes = [sl.numel() for m in self.mlp.values() for sl in m.slabs]
            self._slab_flat = torch.zeros(sum(sizes), device=dev)
            at = 0
            for m in self.mlp.values():
                for i, sl in enumerate(m.slabs):
                    m.slabs[i] = self._slab_flat[at:at + sl.numel()].view(sl.shape)
                    at += sl.numel()
        if self.use_period_kernel:
            fits = ops.gnn_period_ok(N, E, self.Dn)
            hist_ok = (not train) or self._mode_now == "fused" or (self._mode_now == "hist" and self.mlp["output"].native)
            if fits and hist_ok:
                kmaps = ops.gnn_period_kmaps(self.Dn)
                self.ppack = {name: ops.GnnPeriodPack(self._linears(name), kmaps[name], 1 if name == "output" else 32, dev)
                              for name in MODULES}
                i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)  # noqa: E731
                self.node_row0 = i32([self.F_store + w * prob.Ww for w in range(Wn)] + [s_ * prob.Ws for s_ in range(S)])
                self.node_slots = i32([prob.Ww] * Wn + [prob.Ws] * S)
                self._period = True
                self.edge_scratch = torch.empty(ops.gnn_period_edge_scratch_floats(E, prob.B), device=dev) if fits == 2 else None
            elif self.use_period_kernel is True:
                raise ValueError("use_period_kernel: " + ("the backward mode keeps row-layout histories" if fits else
                                                          f"{N} nodes + {E} edges do not fit in a workgroup's LDS"))
        self._period_bwd, self._bdesc = False, {}
        if train and self.use_period_bwd:
            ok = self._mode_now == "hist" and self.mlp["output"].native and self.Dn <= 32
            if ok:
                self.bpack = {name: ops.GnnPeriodBwdPack(self._linears(name), 1 if name == "output" else 32, sg, dev)
                              for name, sg in zip(MODULES, (1, 2, 3, 3, 1))}
                n_blocks = (prob.B + 15) // 16
                self._n_sub = 2 if n_blocks >= 2 * 256 else 1
                self.bscratch = torch.empty(ops.gnn_period_bwd_scratch_floats(N, E, P.n_live, prob.B, self._n_sub), device=dev)
                i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)  # noqa: E731
                self.node_row0 = i32([self.F_store + w * prob.Ww for w in range(Wn)] + [s_ * prob.Ws for s_ in range(S)])
                self.node_slots = i32([prob.Ww] * Wn + [prob.Ws] * S)
                self._period_bwd = True
            elif self.use_period_bwd is True:
                raise ValueError("use_period_bwd: the backward mode keeps row-layout histories / stored inputs, or more than 32 "
                                 "node feature rows")
        zb = (lambda *s_: torch.empty(*s_, device=dev)) if dense else z   # (written in full by the period kernel before any read)
        self.agg = zb(Tp, 32, 2 * N, ld)   # message aggregation: [:, :N] over incoming edges, [:, N:] over outgoing edges
        self.nodes1, self.edges1 = zb(Tp, 32, N, ld), zb(Tp, 32, E, ld)
        self.sums, self.ratio, self.scale = z(T, Wn, ld), z(T, Wn, ld), z(T, Wn, ld)
        if train:
            self.g_state = [z(f_tot, ld), z(f_tot, ld)]
            self.g_orders = z(S * Wn + Wn, ld)
            self.g_reward = z(ld)
            self.d_nodes0, self.d_nodes1 = z(32, N, ld), z(32, N, ld)
            self.d_edges0, self.d_edges1 = z(32, E, ld), z(32, E, ld)
            self.d_out = z(1, E, ld)
        # every packed form of the weights (per-MLP launches, forward period kernel, backward period kernel) from ONE gather per run
        items = [(m, ["packed", "packed_t"]) for m in self.mlp.values()]
        if self._period:
            items += [(pk, ["buf"]) for pk in self.ppack.values()]
        if train and self._period_bwd:
            items += [(pk, ["buf"]) for pk in self.bpack.values()]
        self._wplan = ops.WeightPackPlan(items, dev)
        self._key = key

    def _views(self, block, prob):
        store = block[:self.F_store].view(prob.S, prob.Ws, -1)
        wh = block[self.F_store:].view(prob.Wn, prob.Ww, -1)
        return EnvState(store, wh, None)

    def _order_tables(self, block, prob):
        ld, n = prob.ldb, prob.S * prob.Wn
        so, wo = block[:n].view(prob.S, prob.Wn, -1), block[n:]
        return Table(so, prob.Wn * ld, 1, ld), Table(wo, ld, 1)

    # ---- one batch ------------------------------------------------------------------------------------------------------------
    def run(self, data, periods, ignore_periods=0, train=True, observation_params=None, demand_soa=None, grad_scale=None,
            assign_grads=True, discrete_allocation=False):
        """Same contract as `FusedRollout.run`: returns (total, reported); with `train`, d(mean_loss)/d(theta) in `param.grad`."""
        if discrete_allocation:
            raise ValueError("discrete allocation takes the generic route")
        prob = self._probs.get(self.problem_params, data, self.device)
        T, B, ld, S = periods, prob.B, prob.ldb, prob.S
        self._setup(prob, data, T, train)
        self.prob = prob
        P = self.plan
        shift = observation_params["demand"]["period_shift"] if observation_params else 0
        if demand_soa is None:
            demand_soa = demand_trace_soa(data["demands"], ld, self.device)
        if demand_soa.shape[0] < T + shift:
            raise ValueError("Current period is greater than the number of periods in the data")
        self.demand = demand_soa
        self._wplan.pack()
        self._train = bool(train)
        # per-edge lead-time input rows: sample 0 of THIS batch stands for the batch, as upstream re-reads it every forward
        # (:984) - refreshed on the device (no sync, capturable), so a later batch of the same shape never sees stale values
        self._zl_pairs = ()
        if self.zero_lead_orders == "upstream" and P.Wn > 1:   # (one warehouse: every store has its edge - the rule changes nothing)
            from . import parallel
            if parallel.active() and parallel.world_size() > 1:
                raise ValueError("zero_lead_orders='upstream' couples neighbouring scenarios across shard boundaries: single process only")
            pairs = getattr(prob, "_zero_lead_pairs", None)
            if pairs is None:   # once per presented batch (ProblemCache pins the tensors an entry was keyed on): no sync on later runs
                lt = data["lead_times"].detach()
                if not bool((lt == lt[:1]).all()):
                    raise ValueError("zero_lead_orders='upstream' on the fused GNN route needs lead times that are the same for "
                                     "every scenario of the batch (the Simulator route takes per-scenario lead times)")
                lt0 = lt[0].cpu()
                pairs = prob._zero_lead_pairs = tuple((s_, j_) for s_ in range(lt0.shape[0]) for j_ in range(lt0.shape[1])
                                                      if float(lt0[s_, j_]) == 0.0)
            self._zl_pairs = pairs
        if (self.zero_lead_orders, self._zl_pairs) != getattr(self, "_graph_rule", None):   # captured launches embody the rule
            self._graph_rule, self._graphs, self._eager_runs = (self.zero_lead_orders, self._zl_pairs), {}, 0
            self._pdesc, self._bdesc = {}, {}   # (... and so do the cached launch descriptors)
        P.lead[0, :P.n_int].copy_(data["lead_times"][0][P.lead_store, P.lead_wh])
        P.lead[0, P.n_int:P.n_int + P.Wn].copy_(data["warehouse_lead_times"][0, :P.Wn])
        s0 = self._views(self.states[0], prob)
        s0.store[:, :, :B].copy_(data["initial_inventories"].permute(1, 2, 0))
        s0.wh[:, :, :B].copy_(data["initial_warehouse_inventories"].permute(1, 2, 0))
        # static node features (rows max_inv..): warehouses [holding, (edge cost)], stores [holding, underage, mean, std]
        f, mi, Wn = self.feat, self.max_inv, P.Wn
        f[:, mi, :Wn, :B] = data["warehouse_holding_costs"].t()
        if self.has_edge_cost:
            f[:, mi + 1, :Wn, :B] = data["warehouse_edge_costs"].t()
        for r, k in enumerate(("holding_costs", "underage_costs", "mean", "std")):
            f[:, mi + r, Wn:, :B] = data[k].t()
        if self._graph_on():  # captured launches point at engine-owned buffers: keep the demand trace in one of them
            if getattr(self, "_demand_buf", None) is None or self._demand_buf.shape != demand_soa.shape:
                self._demand_buf, self._graphs, self._eager_runs = torch.empty_like(demand_soa), {}, 0
            if self._demand_buf.data_ptr() != demand_soa.data_ptr():
                self._demand_buf.copy_(demand_soa)
            demand_soa = self.demand = self._demand_buf
            if getattr(self, "_graph_prob", None) is not None and self._graph_prob.same_layout(prob):
                self._graph_prob.copy_tables_from(prob)
                prob = self.prob = self._graph_prob
            else:
                self._graph_prob, self._graphs, self._eager_runs = prob, {}, 0

        def forward():
            for t in range(T):
                self._forward_period(t, prob, demand_soa, shift)
        probe = None
        if self.use_graph == "auto" and self._auto_graph is None and train and self._eager_runs >= 1 and self.timer is None:
            import time
            probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            torch.cuda.synchronize()
            probe[0].record()
            probe_t0 = time.perf_counter()
        self._replay_or_capture(("fwd", T, shift, bool(train)), forward)
        total = self.rewards.sum()
        reported = self.rewards[ignore_periods:].sum() if ignore_periods else total
        if not train:
            self._eager_runs += 1
            return total, reported
        if grad_scale is None:
            grad_scale = 1.0 / (B * T * self.problem_params["n_stores"])
        self.g_reward.zero_()
        self.g_reward[:B] = grad_scale
        self._slab_flat.zero_()
        self.g_state[0].zero_()

        def backward():
            g_next, g_cur = self.g_state
            for t in range(T - 1, -1, -1):
                self._backward_period(t, prob, demand_soa, shift, g_next, g_cur)
                g_next, g_cur = g_cur, g_next
            self._weight_gradients(T, prob)
        self._replay_or_capture(("bwd", T, shift), backward)
        if probe is not None:   # host time to enqueue the step against GPU time to run it (one synchronisation, once per shape)
            host_ms = (time.perf_counter() - probe_t0) * 1e3
            probe[1].record()
            probe[1].synchronize()
            gpu_ms = probe[0].elapsed_time(probe[1])
            self._auto_graph = host_ms > 0.85 * gpu_ms
            self.auto_graph_probe = {"host_enqueue_ms": host_ms, "gpu_ms": gpu_ms, "replay": self._auto_graph}
        self._eager_runs += 1
        if assign_grads:
            for p, g in self.param_grads():
                p.grad = g
        return total, reported

    def param_grads(self):
        """[(parameter, gradient buffer of the last training run)] - engine-owned buffers, overwritten by the next run."""
        return [(t, g) for m in self.mlp.values() for i, lin in enumerate(m.linears)
                for t, g in ((lin.weight, m.gw[i]), (lin.bias, m.gb[i]))]

    def _replay_or_capture(self, name, fn):
        """The launch sequence of a rollout is identical from call to call (same buffers, same shapes): after one eager run
        it is captured into a HIP graph and replayed, which removes the host cost of ~3,000 launches and descriptor builds."""
        if not self._graph_on() or self.timer is not None or self._eager_runs < 1:
            return fn()
        g = self._graphs.get(name)
        if g is None:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                fn()
            self._graphs[name] = g
        g.replay()

    def _weight_gradients(self, T, prob):
        """dW = sum over the slab slots the backward kernels added their per-workgroup weight gradients to."""
        for m in self.mlp.values():
            for i, lin in enumerate(m.linears):
                ops.wgrad_reduce(m.slabs[i], m.gw[i], m.gb[i], lin.weight.shape[1], 1.0)

    def _desc(self, m, segs, prob):
        return ops.mlp3_desc(segs, m.packed, m.n_live, prob.B, prob.ldb, m.n_out, m.out_act, m.hist_stride, m.packed_t,
                             hist_native=m.native, ent_row_stride=m.n_ent * prob.ldb if m.n_live != m.n_ent else 0)

    def _segments(self, t):
        """Input segments of the five MLPs at period t (the graph's gathers; no concatenation is materialised)."""
        P, M = self.plan, self.mlp
        t = t % self._Tp
        nodes0, edges0 = M["initial_node"].Y[t], M["initial_edge"].Y[t]
        return {
            "initial_node": [Mlp3Segment(self.feat[t])],
            "initial_edge": [Mlp3Segment(nodes0, P.src), Mlp3Segment(nodes0, P.tgt), Mlp3Segment(P.lead, None, per_scenario=False)],
            "node_update": [Mlp3Segment(nodes0), Mlp3Segment(self.agg[t][:, :P.n_nodes]), Mlp3Segment(self.agg[t][:, P.n_nodes:])],
            "edge_update": [Mlp3Segment(edges0), Mlp3Segment(self.nodes1[t], P.src), Mlp3Segment(self.nodes1[t], P.tgt)],
            "output": [Mlp3Segment(self.edges1[t])],
        }

    def _run_mlp(self, name, t, segs, prob, residual=None, Ysum=None):
        m = self.mlp[name]
        hist = ((m.hist(m.X, t) if m.X is not None else None, m.hist(m.H1, t), m.hist(m.H2, t)) if hasattr(m, "H1")
                else (None, None, None))
        # rows the launch moves beyond [inputs | hidden activations | output]: the sum it also writes (the residual it adds is one
        # of its input segments in both uses here, i.e. already counted)
        m.fold_rows = m.n_out if Ysum is not None else 0
        self._k("mlp3_fwd_" + name, ops.mlp3_fwd, self._desc(m, segs[name], prob), m.Y[t % self._Tp], *hist, residual, Ysum)

    def _period_desc(self, t, prob, demand_soa, shift):
        """`NicGnnPeriod` of period t (cached: every buffer it names is engine-owned and fixed for a shape)."""
        key = (demand_soa.data_ptr(), shift, self._train, id(prob))
        d = self._pdesc.get(t)
        if d is not None and d._key == key:
            return d
        P, ld, S, train = self.plan, prob.ldb, prob.S, self._train
        d = _lib.NicGnnPeriod()
        d.n_nodes, d.n_edges, d.n_live, d.n_scenarios, d.ldb = P.n_nodes, P.n_edges, P.n_live, prob.B, ld
        d.Dn, d.max_inv, d.store_feat = self.Dn, self.max_inv, int(train)
        p = _lib.ptr
        d.src, d.tgt, d.agg_off, d.agg_items, d.agg_scale = p(P.src), p(P.tgt), p(P.agg_off), p(P.agg_items), p(P.agg_scale)
        d.lead, d.node_row0, d.node_slots = p(P.lead), p(self.node_row0), p(self.node_slots)
        d.n_agg_items = int(P.n_agg_items)
        tp = t % self._Tp
        d.state, d.feat = p(self.states[t]), p(self.feat[tp])
        d.agg = p(self.agg[tp]) if train else None
        for i, name in enumerate(MODULES):
            m, q = self.mlp[name], d.mlp[i]
            q.wpk, q.row_stride = p(self.ppack[name].buf), m.n_ent * ld
            q.Y = p(m.Y[tp]) if (train or name == "output") else None
            if train and m.mode == "hist":
                q.H1, q.H2 = p(m.hist(m.H1, t)), p(m.hist(m.H2, t))
        if train:
            d.mlp[2].Ysum, d.mlp[3].Ysum = p(self.nodes1[tp]), p(self.edges1[tp])
        d.fuse_env = int(P.Wn == 1 and bool(self.fuse_alloc_env))
        if d.fuse_env:
            st, nxt, orders = self._views(self.states[t], prob), self._views(self.states[t + 1], prob), self.orders[t]
            d.io = prob.make_io(st.store, st.wh, None, Table(demand_soa[t + shift], ld, 1), Table(orders[:S].view(S, 1, -1), ld, 1, ld),
                                Table(orders[S:], ld, 1), None)
            d.e_self, d.e_supplier, d.cap_at_one = (-1 if P.e_self is None else P.e_self), P.e_supplier, int(not P.transshipment)
            d.orders, d.sums, d.ratio, d.scale = p(orders), p(self.sums[t]), p(self.ratio[t]), p(self.scale[t])
            d.store_out, d.wh_out, d.reward = p(nxt.store), p(nxt.wh), p(self.rewards[t])
        d.edge_scratch = p(self.edge_scratch) if self.edge_scratch is not None else None
        d._key = key
        self._pdesc[t] = d
        return d

    def _forward_period(self, t, prob, demand_soa, shift):
        P, M, B, ld, S = self.plan, self.mlp, prob.B, prob.ldb, prob.S
        st = self._views(self.states[t], prob)
        if self._period:
            d = self._period_desc(t, prob, demand_soa, shift)
            self._k("gnn_period_fwd", ops.gnn_period_fwd, d)
            if not d.fuse_env:
                self._alloc_env_fwd(t, prob, demand_soa, shift, st)
            return
        # node features: pipelines, padded to the longest one (:846-905)
        tp = t % self._Tp
        self.feat[tp].view(-1, ld).index_copy_(0, self.feat_rows, self.states[t])
        segs = self._segments(t)
        self._run_mlp("initial_node", t, segs, prob)
        self._run_mlp("initial_edge", t, segs, prob)
        edges0 = M["initial_edge"].Y[tp]
        ops.segment_sum(self.agg[tp], edges0, P.agg_off, P.agg_items, P.agg_scale)
        self._run_mlp("node_update", t, segs, prob, M["initial_node"].Y[tp], self.nodes1[tp])   # nodes1 = nodes0 + update
        self._run_mlp("edge_update", t, segs, prob, edges0, self.edges1[tp])                    # edges1 = edges0 + update
        self._run_mlp("output", t, segs, prob)
        self._alloc_env_fwd(t, prob, demand_soa, shift, st)

    def _alloc_env_fwd(self, t, prob, demand_soa, shift, st):
        P, M, B, ld, S = self.plan, self.mlp, prob.B, prob.ldb, prob.S
        out = M["output"].Y[t % self._Tp][0]                       # [E][ld] desired quantity per edge
        # proportional allocation of the warehouse's on-hand stock over its outgoing edges + self loop (:111-138, :1435-1492)
        orders = self.orders[t]
        if P.Wn == 1 and self.fuse_alloc_env:   # allocation head + env step in one launch (round 4, csrc/gnn_alloc_env.hip)
            self._k("alloc_env_fwd", ops.gnn_alloc_env_fwd, prob, st, Table(demand_soa[t + shift], ld, 1), out, orders, self.sums[t],
                    self.ratio[t], self.scale[t], P.e_self, P.e_supplier, not P.transshipment, self._views(self.states[t + 1], prob),
                    self.rewards[t])
            return
        if P.Wn == 1:
            ops.gnn_alloc_fwd(out, st.wh[0, 0], orders, self.sums[t], self.ratio[t], self.scale[t], S, P.e_self, P.e_supplier,
                              not P.transshipment, B)
        else:   # one lane per (scenario, warehouse); every warehouse's stock over its own edges + self loop
            ops.gnn_alloc_groups_fwd(out, st.wh, orders, self.sums[t], self.ratio[t], self.scale[t], P.groups, P.order_row,
                                     not P.transshipment, B)
        ts, tw = self._order_tables(orders, prob)
        # (zero_lead_orders="upstream": the launch itself adds the non-zero orders of zero-lead pairs where the reference's
        # flat-index put leaves them - round 6; before, a handful of torch ops per pair patched the state between launches)
        self._k("env_fwd", ops.env_step_fwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, None,
                out=self._views(self.states[t + 1], prob), reward=self.rewards[t], zero_lead_upstream=bool(self._zl_pairs))

    def _backward_period(self, t, prob, demand_soa, shift, g_next, g_cur):
        P, M, B, ld, S = self.plan, self.mlp, prob.B, prob.ldb, prob.S
        st = self._views(self.states[t], prob)
        ts, tw = self._order_tables(self.orders[t], prob)
        g_so, g_wo = self.g_orders[:S * P.Wn].view(S, P.Wn, -1), self.g_orders[S * P.Wn:]
        gc = self._views(g_cur, prob)
        fused = P.Wn == 1 and self.fuse_alloc_env
        if fused and self._period_bwd:
            # ... in the SAME launch as the MLP adjoints (round 6: the adjoint pair runs on the first wavefronts of nic_gnn_period_bwd)
            self._k("gnn_period_bwd", ops.gnn_period_bwd, self._period_bwd_desc(t, prob, g_cur, demand_soa, shift, g_next))
            return
        if fused:   # env-step adjoint + allocation adjoint in one launch
            self._k("alloc_env_bwd", ops.gnn_alloc_env_bwd, prob, st, Table(demand_soa[t + shift], ld, 1), M["output"].Y[t][0],
                    self.orders[t], self.sums[t], self.ratio[t], self.scale[t], P.e_self, P.e_supplier, not P.transshipment,
                    self._views(g_next, prob), Table(self.g_reward, 0, 1), gc, self.g_orders, self.d_out[0])
        else:
            self._k("env_bwd", ops.env_step_bwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, None,
                    self._views(g_next, prob), Table(self.g_reward, 0, 1), g_in=gc, g_orders=(g_so, g_wo, None),
                    zero_lead_upstream=bool(self._zl_pairs))   # (with the rule's adjoint: the order's gradient is the state
            #                                                      gradient of the element it was added to, where it was not 0)
        # allocation adjoint: alloc_e = out_e * min(1, on_hand / (sum + eps)) for the members, supplier edge passes through
        if fused:
            pass
        elif P.Wn == 1:
            ops.gnn_alloc_bwd(M["output"].Y[t][0], st.wh[0, 0], self.g_orders, self.sums[t], self.ratio[t], self.scale[t],
                              self.d_out[0], gc.wh[0, 0], S, P.e_self, P.e_supplier, not P.transshipment, B)
        else:
            ops.gnn_alloc_groups_bwd(M["output"].Y[t][0], st.wh, self.g_orders, self.sums[t], self.ratio[t], self.scale[t],
                                     self.d_out[0], gc.wh, P.groups, P.order_row, P.e_demand, S, not P.transshipment, B)
        if self._period_bwd:   # the five MLPs, every adjoint gather / aggregation and the row adds into g_cur: one launch
            self._k("gnn_period_bwd", ops.gnn_period_bwd, self._period_bwd_desc(t, prob, g_cur))
            return
        segs = self._segments(t)
        # output MLP -> edges1
        m = M["output"]
        self._mlp_bwd(m, t, segs, prob, self.d_out, dX=self.d_edges1)
        # edges1 = edges0 + edge_update(edges0, nodes1[src], nodes1[tgt]).  The adjoints of the residual connections, of the
        # endpoint gathers and of the message aggregation are sums of segment sums: one multi-term launch per destination.
        eu = M["edge_update"]
        self._mlp_bwd(eu, t, segs, prob, self.d_edges1)
        ops.segment_sum_terms(self.d_nodes1, [(eu.dX[32:64], *P.n_as_src_live, None), (eu.dX[64:96], *P.n_as_tgt_live, None)])
        # nodes1 = nodes0 + node_update(nodes0, incoming, outgoing)
        nu = M["node_update"]
        self._mlp_bwd(nu, t, segs, prob, self.d_nodes1)
        ops.segment_sum_terms(self.d_edges0, [(self.d_edges1, None, None, None), (eu.dX[:32], None, None, None),
                                              (nu.dX[32:64], *P.e_from_tgt, P.e_in_scale),
                                              (nu.dX[64:96], *P.e_from_src, P.e_out_scale)])
        # edges0 = initial_edge(nodes0[src], nodes0[tgt], lead)
        ie = M["initial_edge"]
        self._mlp_bwd(ie, t, segs, prob, self.d_edges0)
        ops.segment_sum_terms(self.d_nodes0, [(self.d_nodes1, None, None, None), (nu.dX[:32], None, None, None),
                                              (ie.dX[:32], *P.n_as_src, None), (ie.dX[32:64], *P.n_as_tgt, None)])
        # nodes0 = initial_node(features): the pipeline rows of the features are the state
        m = M["initial_node"]
        self._mlp_bwd(m, t, segs, prob, self.d_nodes0)
        gc.wh += m.dX[:prob.Ww, :P.Wn].permute(1, 0, 2)
        gc.store += m.dX[:prob.Ws, P.Wn:].permute(1, 0, 2)

    def _period_bwd_desc(self, t, prob, g_cur, demand_soa=None, shift=0, g_next=None):
        """`NicGnnPeriodBwd` of period t (cached per (period, state-gradient buffers, demand trace): every other buffer it names is
        engine-owned).  With `g_next` the env-step / allocation adjoint of the period runs inside the launch (one warehouse)."""
        key = (t, g_cur.data_ptr(), None if g_next is None else (g_next.data_ptr(), demand_soa.data_ptr(), shift, id(prob)))
        d = self._bdesc.get(key)
        if d is not None:
            return d
        P, ld = self.plan, prob.ldb
        d = _lib.NicGnnPeriodBwd()
        d.n_nodes, d.n_edges, d.n_live, d.n_scenarios, d.ldb, d.Dn, d.n_sub = P.n_nodes, P.n_edges, P.n_live, prob.B, ld, self.Dn, self._n_sub
        p = _lib.ptr
        d.src, d.tgt, d.lead, d.node_row0, d.node_slots, d.agg_scale = (p(P.src), p(P.tgt), p(P.lead), p(self.node_row0),
                                                                        p(self.node_slots), p(P.agg_scale))
        for k, (off, items) in enumerate((P.n_as_src_live, P.n_as_tgt_live, P.n_as_src, P.n_as_tgt)):
            d.list_off[k], d.list_items[k], d.n_items[k] = p(off), p(items), P.list_len[k]
        M = self.mlp
        d.feat, d.nodes0, d.nodes1 = p(self.feat[t]), p(M["initial_node"].Y[t]), p(self.nodes1[t])
        d.edges0, d.edges1, d.agg = p(M["initial_edge"].Y[t]), p(self.edges1[t]), p(self.agg[t])
        d.node_row_stride, d.edge_row_stride = P.n_nodes * ld, P.n_edges * ld
        d.d_out, d.g_state, d.scratch = p(self.d_out), p(g_cur), p(self.bscratch)
        for i, name in enumerate(MODULES):
            m, q = M[name], d.mlp[i]
            q.wpk_t, q.Y, q.H1, q.H2, q.row_stride = p(self.bpack[name].buf), p(m.Y[t]), p(m.hist(m.H1, t)), p(m.hist(m.H2, t)), m.n_ent * ld
            (q.slab1, q.lds1), (q.slab2, q.lds2), (q.slab3, q.lds3) = [(p(sl), sl.stride(1)) for sl in m.slabs]
        if g_next is not None:
            S = prob.S
            st, orders = self._views(self.states[t], prob), self.orders[t]
            gn, gc = self._views(g_next, prob), self._views(g_cur, prob)
            d.fuse_env = 1
            d.io = prob.make_io(st.store, st.wh, None, Table(demand_soa[t + shift], ld, 1), Table(orders[:S].view(S, 1, -1), ld, 1, ld),
                                Table(orders[S:], ld, 1), None)
            d.e_self, d.e_supplier, d.cap_at_one = (-1 if P.e_self is None else P.e_self), P.e_supplier, int(not P.transshipment)
            d.sums, d.ratio, d.scale = p(self.sums[t]), p(self.ratio[t]), p(self.scale[t])
            d.g_store_out, d.g_wh_out, d.g_reward = p(gn.store), p(gn.wh), Table(self.g_reward, 0, 1).t2()
            d.g_store_in, d.g_wh_in, d.g_orders = p(gc.store), p(gc.wh), p(self.g_orders)
            d._keep = (st, gn, gc, orders)
        self._bdesc[key] = d
        return d

    def _mlp_bwd(self, m, t, segs, prob, dY, dX=None):
        if m.mode == "hist":
            self._k("mlp3_bwd_" + m.name, ops.mlp3_bwd_hist, self._desc(m, segs[m.name], prob), dY, m.Y[t],
                    m.hist(m.X, t) if m.X is not None else None, m.hist(m.H1, t), m.hist(m.H2, t), dX if dX is not None else m.dX, m.slabs)
            return
        self._k("mlp3_bwd_" + m.name, ops.mlp3_bwd_fused, self._desc(m, segs[m.name], prob), dY, m.Y[t],
                dX if dX is not None else m.dX, m.slabs)

    # ---- inspection helpers used by the parity tests --------------------------------------------------------------------------
    def per_period_rewards(self):
        return self.rewards[:, :self.prob.B]

    def final_state(self):
        from .layout import ref_view
        st = self._views(self.states[-1], self.prob)
        return {"store_inventories": ref_view(st.store, self.prob.B), "warehouse_inventories": ref_view(st.wh, self.prob.B)}
