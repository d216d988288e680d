"""`FusedRollout`: the unrolled T-period rollout + backward of `Trainer.simulate_batch` / `do_one_epoch`
(trainer.py:160-216) for the MLP policies, without an autograd graph and without host round-trips.

Per period the engine enqueues, on one HIP stream, the policy GEMMs (csrc/linear_mfma.hip), the feasibility head
(csrc/policy_heads.hip) and the env step (csrc/env_step.hip); the backward sweep walks the periods in reverse with the
analytic kernels (env bwd -> head bwd -> wgrad / dgrad per layer), accumulating weight gradients in per-split slabs that
are reduced once at the end.  Everything a period needs later is kept RESIDENT in HBM in scenario-minor layout:

    states   [T+1][F][ldb]   F = S*Ws + Wn*Ww + E*We   — state_t IS the MLP input X_t (feature order = the reference's
                                                        cat(flatten(...)) order, neural_networks.py:207,328,380)
    hidden_l [T][N_l][ldb]   post-ELU activations (only when training; ELU' is recovered from them)
    logits   [T][N_out][ldb] ; orders [T][S*nsup+Wn+E][ldb] ; rewards [T][ldb] ; demand [T][S][ldb]

(BASELINE cfg3: 65,536 scenarios x T=100 x 3x512 hidden = 40 GB of activations — sized for 288 GB of HBM3E.)
Round 3: the real-data `data_driven` policy rides on the same engine (every period's block = [state rows | observation rows:
past-demand window, costs, days from christmas, lead times], head = ReLU + adjacency mask + proportional allocation), and the
many-warehouse vanilla head's logits layer runs on its LIVE rows only (pairs without an edge are never read upstream).
The host loop issues ~6 launches per period forward and ~11 backward and never synchronises; the only device->host
transfers are the two scalars the trainer reports.
"""
import torch

from . import _lib, ops
from . import horizon_rollout as hz
from . import small_rollout as sr
from .layout import demand_trace_soa, EnvProblem, ProblemCache, Table, pad_ld
from .ops import EnvState

_HEADS = {"vanilla_one_store": "softplus", "vanilla_warehouse": "warehouse", "vanilla_serial": "serial",
          "data_driven": "data_driven"}


def _pad32(n):
    return (n + 31) // 32 * 32


class _HorizonRefused(Exception):
    """nic_horizon_rollout_ok refused a shape the Python-side plan accepted (FusedRollout.run falls back to the per-period route)."""


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on (torch's current
    stream).  bench.py uses it to report the dominant kernel's average duration over the timed region."""

    def __init__(self, tags=None, stride=1, record_order=False):
        """stride: bracket every `stride`-th launch of each kernel class only (a pair of event records costs ~5 us of
        host + stream time; at ~1,700 launches per cfg3 step timing all of them slows the step by 4 %).
        record_order: keep the (tag, kernel) sequence of every launch - what joins a profiler's per-dispatch counter rows,
        which only carry kernel names, to the kernel classes (two MLPs may share one template)."""
        self.tags = set(tags) if tags is not None else None
        self.order = [] if record_order else None
        self.events = {}
        self.calls = {}
        self.names = {}  # tag -> name of the kernel the C ABI reported launching for it (nic_last_kernel)
        self.stride = max(1, int(stride))
        self.enabled = True

    def wants(self, tag):
        return self.enabled and (self.tags is None or tag in self.tags)

    def call(self, tag, fn, *args, **kw):
        if not self.wants(tag):
            return fn(*args, **kw)
        n = self.calls.get(tag, 0)
        self.calls[tag] = n + 1
        if self.order is not None:
            out = fn(*args, **kw)
            self.order.append((tag, (_lib.lib().nic_last_kernel() or b"").decode()))
            return out
        if n % self.stride:
            return fn(*args, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn(*args, **kw)
        b.record()
        self.events.setdefault(tag, []).append((a, b))
        if tag not in self.names:
            self.names[tag] = (_lib.lib().nic_last_kernel() or b"").decode()
        return out

    def summary(self):
        """tag -> (launches, mean ms over the launches that were bracketed).  Call after torch.cuda.synchronize()."""
        return {t: (self.calls[t], sum(a.elapsed_time(b) for a, b in ev) / len(ev)) for t, ev in self.events.items() if ev}

    def reset(self):
        self.events = {}
        self.calls = {}
        self.names = {}


class FusedRollout:
    @staticmethod
    def supports(model):
        name = getattr(model, "nn_args", {}).get("name") if hasattr(model, "nn_args") else None
        if name not in _HEADS or type(model).__name__ not in ("VanillaOneStore", "VanillaWarehouse", "VanillaSerial",
                                                              "DataDrivenNet"):
            return False
        a = model.nn_args
        if name == "data_driven":   # the real-data policy (data_driven_net.yml): ELU inside, ReLU on the output layer
            return (type(model).__name__ == "DataDrivenNet" and a["inner_layer_activations"]["master"] == "elu"
                    and a["output_layer_activation"]["master"] == "relu" and len(a["neurons_per_hidden_layer"]["master"]) >= 1)
        return (type(model).__name__ != "DataDrivenNet" and a["inner_layer_activations"]["master"] == "elu"
                and a["output_layer_activation"]["master"] is None and len(a["neurons_per_hidden_layer"]["master"]) >= 1)

    @staticmethod
    def observation_ok(model, observation_params, data=None):
        """Can this engine build the policy's observation?  The vanilla policies read the inventories only (no past-demand
        window, no time / sample features: trainer.py's generic loop otherwise); data_driven reads the past-demand window and
        `days_from_christmas` (neural_networks.py:452-470) and nothing else that moves with the period."""
        op = observation_params
        if op is None:
            return False
        past, tf, sf = (op["demand"] or {}).get("past_periods"), op["time_features"], op["sample_features"]
        past = past if isinstance(past, int) else 0   # (`past_periods: null` / a missing key: no window -> the generic route)
        if getattr(model, "nn_args", {}).get("name") == "data_driven":
            # (sample features, e.g. `store_nbr`, may be in the observation: DataDrivenNet.forward does not read them)
            return (isinstance(tf, (list, tuple)) and list(tf) == ["days_from_christmas"] and past >= 1
                    and (data is None or ("days_from_christmas" in data and "underage_costs" in data)))
        return past == 0 and not tf and not sf

    def __init__(self, model, problem_params, device):
        _lib.require_device()
        if not self.supports(model):
            raise ValueError("FusedRollout handles vanilla_one_store / vanilla_warehouse / vanilla_serial MLP policies")
        self.model = model
        self.problem_params = problem_params
        self.device = torch.device(device)
        self.head = _HEADS[model.nn_args["name"]]
        self._key = None
        self.timer = None  # KernelTimer or None
        # replay the launch sequence from a HIP graph (see _replay_or_capture): True / False, or "auto" = decided by MEASUREMENT
        # on the second training run of a shape: if the host needs longer to enqueue the ~13 launches per period than the GPU
        # needs to run them (small batches on a slow or busy host), the step is launch-bound and later runs are replayed
        self.use_graph = False
        self._auto_graph = None   # "auto": None = not measured yet, else the decision; `auto_graph_probe` = what was measured
        self.auto_graph_probe = None
        self.use_small = True   # whole-horizon kernels for the small one-store-chain policies (small_rollout.py)
        self.small_lane_scenarios = 0  # ... 16 or 32 scenarios per wavefront (0: 16 while 32 would leave SIMDs without a wavefront)
        self.small_wgrad_in_kernel = True  # ... with the weight gradients contracted inside the backward kernel (False: dZ history + one GEMM per layer)
        self.use_thin = True    # fused backward of thin (<= 32 rows) output layers (csrc/thin_layer.hip)
        self.batch_wgrad = True  # hidden-layer weight gradients contracted over all periods in one launch
        self.fuse_head_env = True  # vanilla_warehouse: head + env step (and their adjoints) in one launch each (csrc/head_env.hip)
        # ... and, where the shapes allow (<= 16 stores, <= 32 logits, <= 51 state rows: BASELINE cfg3 and the shipped one-warehouse
        # YAML), the whole per-period TAIL in one launch per direction (csrc/period_tail.hip): logits layer + head + env step + the
        # next period's first layer forward; first layer's input gradient + env / head adjoints + logits layer backward
        # "auto": up to `tail_max_scenarios` scenarios, where the per-period launches are latency-bound (measured crossover, see
        # DESIGN section 4); larger batches keep the separate, bandwidth-efficient launches
        self.fuse_tail = "auto"
        self.tail_max_scenarios = 16384
        # the BACKWARD tail launch ("auto"): from this batch size up, and below it whenever the launch sequence is NOT replayed from
        # a HIP graph.  At the reference's shipped batch of 1,024 its dependent chain (28.7 us per launch, rocprof) is longer than
        # the three launches it replaces (9.5 + 6.2 + 8.5 us): replayed, the separate launches win (5.17 vs 5.39 ms per batch); launched
        # eagerly the step is launch-bound and two launches fewer per period win (5.46 vs 6.65 ms at T = 100).  The forward one wins
        # either way (15.7 against 4.9 + 6.0 + 7.9 us); at 8,192 both win
        self.tail_bwd_min_scenarios = 2048
        # ... and, for 512-wide hidden layers (BASELINE cfg3), ALL periods in one launch per direction - an EXPERIMENT that is not in
        # the default library (tools/experiments/wide_rollout.hip, include/nic_experiments.h; NIC_BUILD_EXPERIMENTS=1 builds it): a
        # workgroup carries a block of 32 scenarios through the whole horizon, weights streamed from L2 as pre-packed MFMA fragments.
        # OFF by default: measured it matches the per-period launches at an 8-GPU shard (29.2 vs 28.2 ms at 8,192
        # scenarios) and loses at the full batch (231 vs 180 ms) - its H x H layers run at the tiled GEMM's rate (36-40 us per 8,192
        # columns) while the head / env / first-layer stages of a block serialise behind them on one wavefront per SIMD
        self.use_wide = False
        # data_driven on small batches (what the reference trains it on: 72 products): all periods in ONE forward and ONE backward
        # launch (csrc/horizon_rollout.hip).  Measured against the per-period kernels on the real-data shape (tools/
        # horizon_crossover.py, profiles/r04_horizon_crossover.json): 2.3 vs 8.7 ms (replayed) at 72 scenarios, 3.4 vs 10.2 at 4,096,
        # 7.2 vs 12.2 at 8,192 (one workgroup of 16 scenarios per CU: 512 workgroups are two rounds), 14.2 vs 16.0 at 16,384,
        # 29.2 vs 23.1 at 32,768
        self.use_horizon = True
        self.horizon_max_scenarios = 16384
        self.horizon = None     # HorizonPlan when the current shapes take that route
        self.eval_history = None  # evaluation keeps per-period states/orders/logits: None = while small, True / False = forced
        self.small = None       # SmallRolloutPlan when the current shapes take that route
        self._prob_cache = ProblemCache()
        self._prob = None

    def _k(self, tag, fn, *args, **kw):
        if self.timer is None:
            return fn(*args, **kw)
        return self.timer.call(tag, fn, *args, **kw)

    def _fused_head_env(self, prob):
        """vanilla_warehouse settings the fused head + env-step kernels take (csrc/head_env.hip): up to 64 stores, no extra
        echelons, no order rounding between head and env step (discrete allocation keeps the three separate launches)."""
        return (self.fuse_head_env and self.head == "warehouse" and prob.S <= 64 and prob.E == 0 and prob.Wn >= 1
                and not self._round)

    def _use_wide(self):
        """the whole-horizon forward kernel of the wide policy (csrc/wide_rollout.hip) runs for the current shapes and options"""
        return getattr(self, "_wide_shapes", False) and not self._round

    def _use_tail(self):
        """the fused per-period tail launches (csrc/period_tail.hip) run for the current shapes and options"""
        return self._tail_shapes and not self._round

    def _use_tail_bwd(self):
        """the backward tail launch: forced on -> always; "auto" -> from tail_bwd_min_scenarios up, or when launched eagerly"""
        return (self._use_tail() and getattr(self, "_tail_slots", 0) > 0
                and (self.fuse_tail is True or self.prob.B >= self.tail_bwd_min_scenarios or not self._graph_on()))

    def _graph_on(self):
        return self.use_graph is True or (self.use_graph == "auto" and self._auto_graph is True)

    # ---- buffers ------------------------------------------------------------------------------------------------
    def _linears(self):
        lins = self.model.master_linears()
        if any(isinstance(m.weight, torch.nn.parameter.UninitializedParameter) for m in lins):
            raise RuntimeError("policy has un-materialised LazyLinear layers; call materialize() or run one forward")
        return lins

    def materialize(self, in_features):
        """Materialises LazyLinear layers without a forward pass (same default init as the reference's first call)."""
        k = in_features
        for m in self.model.master_linears():
            fresh = isinstance(m.weight, torch.nn.parameter.UninitializedParameter)
            if fresh:
                m.in_features = k
                m.weight.materialize((m.out_features, k))
                if m.bias is not None:
                    m.bias.materialize((m.out_features,))
                m.reset_parameters()
            elif hasattr(m, "cls_to_become") and m.cls_to_become is not None:
                # a lazy layer whose parameters came from load_state_dict (checkpoint loaded into a fresh model): the
                # tensors exist, but the module still reports in_features = 0 and is still the lazy class
                m.in_features = m.weight.shape[1]
            if fresh or (hasattr(m, "cls_to_become") and m.cls_to_become is not None):
                for hook in ("_initialize_hook", "_load_hook"):  # what LazyModuleMixin._infer_parameters does
                    if hasattr(m, hook):
                        getattr(m, hook).remove()
                        delattr(m, hook)
                m.__class__ = m.cls_to_become
            k = m.out_features

    def _setup(self, prob, T, train, extra_rows=0):
        key = (prob.B, T, bool(train), prob.S, prob.Wn, prob.E, prob.Ws, prob.Ww, prob.We, self.batch_wgrad, self.use_thin,
               self.eval_history, self.small_wgrad_in_kernel, extra_rows, self.small_lane_scenarios, self.use_horizon,
               self.horizon_max_scenarios, getattr(self, "_shift_hint", 0), self.fuse_tail, self.tail_max_scenarios,
               self.tail_bwd_min_scenarios, self.fuse_head_env, self.use_wide)
        if self._key == key:
            return
        dev, ld = self.device, prob.ldb
        self._prob = None
        self._auto_graph = None   # (a new shape is measured afresh)
        for name in ("states", "orders", "logits", "hidden", "dZhist", "dZlast_hist", "dH", "slabs", "sr_states", "sr_hidden",
                     "sr_logits", "sr_dzh", "sr_dzo", "sr_slab", "sr_grad", "hz_X", "hz_z1", "hz_hist", "hz_dz"):
            setattr(self, name, None)  # release the previous shapes' buffers before sizing the new ones
        self.F_store, self.F_wh, self.F_ech = prob.S * prob.Ws, prob.Wn * prob.Ww, prob.E * prob.We
        F = self.F_store + self.F_wh + self.F_ech
        if self.head == "softplus":
            F = self.F_store
        # rows of the MLP input that are the state (what the first layer's input gradient is needed for); data_driven appends
        # `extra_rows` observation rows to every period's block: [past-demand window | costs | days from christmas | lead times]
        self.F_dyn = F
        F += extra_rows
        self.F = F
        self.materialize(F)
        lins = self._linears()
        dims = [F] + [m.out_features for m in lins]
        assert lins[0].in_features == F, (lins[0].in_features, F)
        self.dims = dims
        L = len(lins)
        z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731

        def zpad(*shape):
            """A history buffer every live column of which is written before it is read (hidden activations, pre-activation
            gradients): uninitialised storage with only the padding columns [B, ldb) zeroed.  (As torch.zeros the histories of
            BASELINE cfg3 were 80 GB of fill at set-up - 53 launches of 2^31 bytes that a profile of a few steps shows as its
            largest `FillFunctor` rows although no step issues them.)"""
            t_ = torch.empty(*shape, device=dev)
            if shape[-1] > prob.B:
                t_[..., prob.B:] = 0.0
            return t_
        self.small = None
        if self.use_small and all(m.bias is not None for m in lins) and sr.SmallRolloutPlan.supports(prob, self.head, dims):
            # ---- whole-horizon route: one forward kernel, one backward kernel, one wgrad GEMM per layer ------------------
            plan = self.small = sr.SmallRolloutPlan(prob, self.head, dims)
            nh, no = plan.n_hidden, plan.n_out
            self.rewards = z(T, ld)
            self.sr_final = z(F, ld)
            self.sr_state0, self._copied = z(F, ld), {}
            self.sr_weights = z(sr.packed_weight_count(F, nh, no))
            if train:
                # (rows padded as the 16-wide kernels keep them - NIC_SR16_STATE_ROWS / NIC_SR16_LOGIT_ROWS; the 32-wide form
                # uses the first F / n_out rows in [row][T][ld] order)
                self.sr_states, self.sr_hidden = z((F + 3) // 4 * 4, T, ld), z(nh * sr.H, T, ld)
                self.sr_logits = z(no if no == 1 else (no + 3) // 4 * 4, T, ld)
                self.g_reward, self._g_reward_key = z(ld), None
                if self.small_wgrad_in_kernel:
                    # weight gradients contracted inside the backward kernel: one partial gradient per wavefront in the
                    # packed-weight layout, summed once; the per-layer gradients are views of that sum
                    n_packed = sr.packed_weight_count(F, nh, no)
                    self.sr_slab = z(sr.small_rollout_bwd_wgrad_slots(prob.B), (n_packed + 3) // 4 * 4)
                    self.sr_grad = z(self.sr_slab.shape[1])
                    self.sr_scratch = None
                    sl = sr.layer_slices(F, nh, no)
                    self.gw = [self.sr_grad[o:o + n * k].view(n, k) for o, n, k, _ in sl]
                    self.gb = [self.sr_grad[bo:bo + n] for _, n, _, bo in sl]
                else:
                    self.sr_dzh, self.sr_dzo = z(nh * sr.H, T, ld), z(no, T, ld)
                    self.splits = [ops.wgrad_num_splits(dims[i + 1], dims[i], T * ld) for i in range(L)]
                    self.slabs = [z(self.splits[i], dims[i + 1], (dims[i] + 1 + 3) // 4 * 4) for i in range(L)]
                    self.gw = [torch.zeros_like(m.weight) for m in lins]
                    self.gb = [torch.zeros_like(m.bias) for m in lins]
            self._key = key
            return
        self.horizon = None
        # (one-store settings - the reference's one_store_real_data YAML trains batches of 8,192 and evaluates 32,768 - keep winning
        # four times as far: 0.68 vs 2.3 ms at 8,200 scenarios x T=12, 1.46 vs 3.24 at 32,800)
        hz_limit = self.horizon_max_scenarios * (4 if prob.S * prob.nsup <= 8 else 1)
        if (self.use_horizon and extra_rows > 0 and prob.B <= hz_limit and all(m.bias is not None for m in lins)
                and hz.HorizonPlan.supports(prob, self.head, dims)
                and hz.offsets_ok(prob, T, T + getattr(self, "_shift_hint", 0), hz.MAX_HIDDEN)):
            # ---- whole-horizon route for data_driven: one forward kernel, one backward kernel, four GEMMs over (period x scenario)
            # columns (the observation rows' share of the first layer; one weight gradient per layer).  Histories [row][T][ld].
            plan = self.horizon = hz.HorizonPlan(prob, dims)
            FD, Fo, n_cols = self.F_dyn, F - self.F_dyn, T * ld
            self.rewards = z(T, ld)
            self.hz_state0, self.hz_final = z(FD, ld), z(FD, ld)
            self.hz_X = z(F, T, ld)               # rows [0, FD): state before period t (written by the kernel); then the observation rows
            self.hz_z1 = z(dims[1], T, ld)
            self.hz_W1obs = z(dims[1], _pad32(Fo))   # (rows padded to 32 floats: the GEMM's A-tile loads are float4)
            if prob.Wn:
                conn = self.problem_params["warehouse_store_adjacency"]
                self.edge_mask = torch.tensor(conn, dtype=torch.float32, device=dev).t().contiguous()   # [S][Wn]
            else:
                self.edge_mask = None
            if train:
                self.hz_hist = [z(dims[1], T, ld), z(dims[2], T, ld), z(dims[3], T, ld), z(plan.n_ord + prob.Wn, T, ld)]  # h1, h2, logits, orders (+ shipped)
                self.hz_dz = [z(dims[i + 1], T, ld) for i in range(L)]   # (padding columns stay zero: the kernel writes live ones only)
                self.g_reward, self._g_reward_key = z(ld), None
                self.splits = [ops.wgrad_num_splits(dims[i + 1], dims[i], n_cols) for i in range(L)]
                self.slabs = [z(self.splits[i], dims[i + 1], (dims[i] + 1 + 3) // 4 * 4) for i in range(L)]
                self.gw = [torch.zeros_like(m.weight) for m in lins]
                self.gb = [torch.zeros_like(m.bias) for m in lins]
            self.demand_buf, self._graphs, self._eager_runs = None, {}, 0
            self._key = key
            return
        # Many-warehouse vanilla head: the logits of (store, warehouse) pairs without an edge are never read upstream
        # (`store_intermediate_outputs[:, connected_stores, w_idx]`, neural_networks.py:403-417: zero gradient, no effect), so
        # the logits layer - forward, input gradient, weight gradient - runs on the LIVE rows only: compact weights, a compact
        # logits block scattered into the head's [S * Wn + Wn] layout by one row copy per period, and the adjoint gather.
        self.live_rows, gd = None, list(dims)
        if self.head == "warehouse" and prob.Wn > 1:
            conn = self.problem_params["warehouse_store_adjacency"]
            live = [s_ * prob.Wn + w for s_ in range(prob.S) for w in range(prob.Wn) if conn[w][s_]] + \
                   [prob.S * prob.Wn + w for w in range(prob.Wn)]
            if len(live) <= 0.85 * dims[-1]:
                self.live_rows = torch.tensor(live, dtype=torch.long, device=dev)
                gd[-1] = len(live)
                # the fused head + env launches read / write the compact rows themselves (nic_head_env_*_rows): row of pair
                # (w, s) in the compact block (0 for a pair without an edge: loaded, never used), then the warehouses' own rows
                pos = {r: i for i, r in enumerate(live)}
                self.zrow = torch.tensor([[pos.get(s_ * prob.Wn + w, 0) for s_ in range(prob.S)] for w in range(prob.Wn)],
                                         dtype=torch.int32, device=dev)
                self.first_wh_row = len(live) - prob.Wn
        self.gd = gd   # layer widths of the GEMMs (= dims unless the logits layer is compacted)
        # the fused per-period tail (csrc/period_tail.hip): the shapes decide here (slab slots below), discrete allocation per run
        want_tail = self.fuse_tail is True or (self.fuse_tail == "auto" and prob.B <= self.tail_max_scenarios)
        self._tail_shapes = bool(want_tail and self.fuse_head_env and self.head == "warehouse" and self.live_rows is None
                                 and extra_rows == 0 and L >= 2 and self.use_thin and all(m.bias is not None for m in lins)
                                 and ops.period_tail_ok(prob, dims[-1], dims[-2], dims[1]))
        n_ord = prob.S * prob.nsup + prob.Wn + prob.E
        f_tot = self.F_store + self.F_wh + self.F_ech + extra_rows
        # evaluation keeps the state / order / logit history only while it is small (tests and short horizons read it);
        # a long-horizon evaluation (test periods: 5000) rolls through two state blocks and one order / logit block
        free_now = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                    if dev.type == "cuda" else 0)
        auto = 4 * (T + 1) * ld * (f_tot + n_ord + dims[-1]) <= 0.1 * free_now
        self._hist = bool(train) or extra_rows > 0 or (auto if self.eval_history is None else bool(self.eval_history))
        # (+ one row of ONES behind every period's block: with the bias as row K of the transposed first-layer weights the bias is one
        # more term of the first layer's contraction - the streamed first-layer forward then issues no bias loads, see _launch_forward)
        self.states = z(T + 1 if self._hist else 2, f_tot + 1, ld)
        self.states[:, f_tot] = 1.0
        self.orders = z(T if self._hist else 1, n_ord, ld)
        self.rewards = z(T, ld)
        self.logits = z(T if self._hist else 1, dims[-1], ld)
        keep = T if train else 1
        self.hidden = [zpad(keep, dims[i + 1], ld) for i in range(L - 1)]
        # engine copies of the weights: rows padded to a multiple of 32 floats so every A-tile load is a float4
        self.Wp = [z(gd[i + 1], _pad32(gd[i])) for i in range(L)]
        self.Wt = [z(gd[i] + (1 if i == 0 else 0), _pad32(gd[i + 1])) for i in range(L)]   # (layer 0: + the bias row, see above)
        # whole-horizon forward of the wide policy: every hidden layer 512 wide, history kept, shapes in the kernel's range
        if self.use_wide and not _lib.has_experiments():
            raise ValueError("use_wide: the whole-horizon kernels of the wide policy are an experiment outside the default library "
                             "(build with NIC_BUILD_EXPERIMENTS=1; include/nic_experiments.h)")
        self._wide_shapes = bool(self.use_wide and self.head == "warehouse" and self.live_rows is None and extra_rows == 0 and L >= 3
                                 and self._hist and all(m.bias is not None for m in lins) and len(set(dims[1:-1])) == 1
                                 and ops.wide_rollout_ok(prob, dims[-1], dims[1], L - 1))
        if self._wide_shapes:
            Hh = dims[1]
            self.Wpk = [z(Hh // 32, Hh // 8, 64, 4) for _ in range(L - 2)]
            self.Wq = z(Hh // 32, 16, 64)
            self._wq_index = ops.wide_pack_out_index(dims[-1], Hh, dev)
            self._wq_pad = z(32, Hh)
            if train:   # the backward kernel's packed operands (transposed hidden layers, first layer, logits layer transposed)
                self.WpkT = [z(Hh // 32, Hh // 8, 64, 4) for _ in range(L - 2)]
                self._win_index = ops.wide_pack_in_index(self.F, Hh, dev)
                self._win_pad = z(Hh, 64)
                self._wot_index = ops.wide_pack_out_t_index(dims[-1], Hh, dev)
                self._wot_pad = z(2 * ops.wide_ns(dims[-1]), Hh)
            self._tail_shapes = False   # (one route per shape: the whole-horizon kernels replace the per-period tail launches)
        self.Zc = z(gd[-1], ld) if self.live_rows is not None else None
        self.bias_c = z(gd[-1]) if self.live_rows is not None else None
        if train:
            self.g_state = [z(self.states.shape[1], ld), z(self.states.shape[1], ld)]
            self.g_orders = z(n_ord, ld)
            self.dZ = z(dims[-1], ld)
            wmax = max(dims[1:-1]) if L > 1 else 1
            self.dH = [z(wmax, ld), z(wmax, ld)]
            # pre-activation gradients of the hidden layers for EVERY period ([T][N_l][ldb], 13.4 GB per 512-wide layer at
            # BASELINE cfg3 — HBM is sized for it): their weight gradients are contracted once per training step over
            # (period x scenario) instead of once per period (see _launch_backward)
            hist_bytes = 4 * T * ld * sum(gd[1:])
            # free HBM = what the driver reports + what torch's caching allocator holds but is not using
            free_bytes = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                          if dev.type == "cuda" else 0)
            # not enough HBM left for the gradient history -> accumulate weight gradients period by period
            batch = self.batch_wgrad and hist_bytes <= 0.6 * free_bytes
            self.dZhist = [zpad(T, dims[i + 1], ld) for i in range(L - 1)] if batch else None
            # ... and the logits gradient too, unless the logits layer takes the fused thin-layer backward (per period)
            thin_last = L > 1 and self.use_thin and ops.linear_bwd_thin_ok(gd[L], gd[L - 1])
            # (the whole-horizon backward kernel leaves the logits gradient history: its layer's weight gradient is contracted over
            # all periods like the hidden layers')
            self._wide_bwd = bool(self._wide_shapes and batch)
            if self._wide_bwd:
                thin_last = False
            self.dZlast_hist = z(T, gd[-1], ld) if batch and not thin_last else None
            self.dZc = z(gd[-1], ld) if self.live_rows is not None else None
            # slab slots per layer: layers contracted over ALL periods in one launch split the (period x scenario) range into
            # (period groups x scenario chunks) so that a batch of 1,024 or 8,192 scenarios still gives every CU a workgroup
            by_periods = [batch and (self._wide_bwd or not (i > 0 and self.use_thin and ops.linear_bwd_thin_ok(gd[i + 1], gd[i])))
                          and not (i == L - 1 and self.dZlast_hist is None) for i in range(L)]   # (as _launch_backward decides)
            self.splits = [ops.wgrad_periods_num_splits(gd[i + 1], gd[i], prob.B, T) if by_periods[i]
                           else ops.wgrad_num_splits(gd[i + 1], gd[i], prob.B) for i in range(L)]
            # the logits layer's slab serves either backward: one slot per workgroup of the fused tail, or the thin layer's splits
            self._tail_slots = ops.period_tail_bwd_slots(prob.B) if self._tail_shapes else 0
            self.splits[L - 1] = max(self.splits[L - 1], self._tail_slots)
            self.slabs = [z(self.splits[i], gd[i + 1], (gd[i] + 1 + 3) // 4 * 4) for i in range(L)]
            self.g_reward, self._g_reward_key = z(ld), None
            self.gw = [torch.zeros_like(m.weight) for m in lins]
            self.gb = [torch.zeros_like(m.bias) if m.bias is not None else None for m in lins]
            if self.live_rows is not None:   # the compact logits layer reduces into these; scattered into gw / gb afterwards
                self.gw_c = z(gd[-1], gd[-2])
                self.gb_c = z(gd[-1]) if lins[-1].bias is not None else None
        self.demand_buf = None
        self._graphs = {}      # "fwd"/"bwd" -> torch.cuda.CUDAGraph (HIP graph) of the launch sequence
        self._eager_runs = 0
        if self.head == "warehouse":
            self.adj = self.model.adjacency(prob.S, prob.Wn, dev)
        if self.head == "data_driven" and prob.Wn:
            conn = self.problem_params["warehouse_store_adjacency"]
            self.edge_mask = torch.tensor(conn, dtype=torch.float32, device=dev).t().contiguous()   # [S][Wn]
        self._key = key

    def _views(self, block, prob):
        """(store, wh, ech) SoA views of one [F][ldb] state block."""
        a, b = self.F_store, self.F_store + self.F_wh
        store = block[:a].view(prob.S, prob.Ws, -1)
        wh = block[a:b].view(prob.Wn, prob.Ww, -1) if prob.Wn else None
        ech = block[b:b + self.F_ech].view(prob.E, prob.We, -1) if prob.E else None
        return EnvState(store, wh, ech)

    def _order_views(self, block, prob):
        a, b = prob.S * prob.nsup, prob.S * prob.nsup + prob.Wn
        return (block[:a].view(prob.S, prob.nsup, -1), block[a:b] if prob.Wn else None, block[b:] if prob.E else None)

    def _order_tables(self, block, prob):
        so, wo, eo = self._order_views(block, prob)
        ld = prob.ldb
        return (Table(so, prob.nsup * ld, 1, ld), Table(wo, ld, 1) if wo is not None else None,
                Table(eo, ld, 1) if eo is not None else None)

    def _ub(self):
        """Scalar order upper bound of the policy (read back from the device once, not per call / per capture)."""
        ub = self.model.warehouse_upper_bound
        if not torch.is_tensor(ub):
            return float(ub)
        key = (ub.data_ptr(), ub._version)
        if getattr(self, "_ub_cache", (None, None))[0] != key:
            self._ub_cache = (key, float(ub.reshape(-1)[0]))
        return self._ub_cache[1]

    # ---- one batch ----------------------------------------------------------------------------------------------
    def run(self, data, periods, ignore_periods=0, train=True, observation_params=None, demand_soa=None,
            grad_scale=None, accumulate_grads=False, discrete_allocation=False, assign_grads=True):
        """Rollout of one batch (and, if `train`, d(mean_loss)/d(theta) into `param.grad`).

        data: the batch dict `Simulator.reset` takes (device tensors).  demand_soa: optional [T][S][ldb] trace already
        in kernel layout (e.g. from Scenario(sampler='hip')) — skips the transpose of data['demands'].
        grad_scale: d(loss)/d(reward[b,t]); default 1/(B*T*S) = trainer.py:169.  Multi-GPU callers pass the GLOBAL B.
        discrete_allocation: orders rounded half-to-even between head and env step (trainer.py:201-202); evaluation only
        (torch.round has zero gradient, so a training step with it goes through the generic route).
        assign_grads=False leaves `param.grad` alone; the gradients are then read with `param_grads()`.
        Returns (total, reported) as 0-d device tensors = simulate_batch's return values (trainer.py:216).
        """
        if discrete_allocation and train:
            raise ValueError("discrete_allocation is an evaluation-time option of the fused rollout")
        self._round = bool(discrete_allocation)
        self._T_last = periods
        dev = self.device
        prob = self._problem_for(data)
        T, B, ld = periods, prob.B, prob.ldb
        extra = 0
        if self.head == "data_driven":
            if not self.observation_ok(self.model, observation_params, data):
                raise ValueError("data_driven needs a past-demand window and the days_from_christmas time feature")
            P_ = observation_params["demand"]["past_periods"]
            extra = prob.S * P_ + 2 * prob.S + data["days_from_christmas"].shape[1] + prob.S * data["lead_times"].shape[2]
        self._shift_hint = observation_params["demand"]["period_shift"] if observation_params else 0
        self._setup(prob, T, train, extra)
        if self._graph_on():
            # a captured graph holds raw pointers: keep the first call's table tensors and refresh their CONTENTS
            if self._prob is not None and self._prob.same_layout(prob):
                self._prob.copy_tables_from(prob)
                prob = self._prob
            else:
                self._prob, self._graphs, self._eager_runs = prob, {}, 0
        self.prob = prob
        shift = observation_params["demand"]["period_shift"] if observation_params else 0
        if demand_soa is None:
            demand_soa = demand_trace_soa(data["demands"], ld, dev)
        if self._graph_on():
            if self.demand_buf is None or self.demand_buf.shape != demand_soa.shape:
                self.demand_buf, self._graphs, self._eager_runs = torch.empty_like(demand_soa), {}, 0
            if self.demand_buf.data_ptr() != demand_soa.data_ptr():
                self.demand_buf.copy_(demand_soa)
            demand_soa = self.demand_buf
        self.demand = demand_soa
        if demand_soa.shape[0] < T + shift:
            raise ValueError("Current period is greater than the number of periods in the data")

        if self.small is not None:
            return self._run_small(data, prob, T, B, ld, shift, demand_soa, ignore_periods, train, grad_scale,
                                   accumulate_grads, assign_grads)
        if self.horizon is not None:
            try:
                return self._run_horizon(data, prob, T, B, ld, shift, demand_soa, ignore_periods, train, grad_scale,
                                         accumulate_grads, assign_grads, observation_params)
            except _HorizonRefused:
                # the C side's own limits (LDS bytes, 32-bit history offsets: nic_horizon_rollout_ok) said no to a shape the
                # Python-side plan took: this engine keeps the per-period launches from here on (a fallback, not a launch error)
                self.use_horizon = False
                return self.run(data, periods, ignore_periods, train=train, observation_params=observation_params,
                                demand_soa=demand_soa, grad_scale=grad_scale, accumulate_grads=accumulate_grads,
                                discrete_allocation=discrete_allocation, assign_grads=assign_grads)

        # engine copies of the weights (tiny) — refreshed every call because the optimizer moves them
        lins = self._linears()
        L = len(lins)
        rows = self.live_rows
        for i, m in enumerate(lins):
            w = m.weight.detach() if (rows is None or i < L - 1) else m.weight.detach()[rows]
            self.Wp[i][:, :self.gd[i]].copy_(w)
            self.Wt[i][:self.gd[i], :self.gd[i + 1]].copy_(w.t())
        biases = [m.bias.detach() if m.bias is not None else None for m in lins]
        if biases[0] is not None:
            self.Wt[0][self.gd[0], :self.gd[1]].copy_(biases[0])
        if rows is not None and biases[-1] is not None:   # (an engine-owned buffer: a captured graph holds its address)
            torch.index_select(biases[-1], 0, rows, out=self.bias_c)
            biases[-1] = self.bias_c
        Wv = [self.Wp[i][:, :self.gd[i]] for i in range(L)]
        Wtv = [self.Wt[i][:self.gd[i], :self.gd[i + 1]] for i in range(L)]

        # initial state
        s0 = self._views(self.states[0], prob)
        s0.store[:, :, :B].copy_(data["initial_inventories"].permute(1, 2, 0))
        if prob.Wn:
            s0.wh[:, :, :B].copy_(data["initial_warehouse_inventories"].permute(1, 2, 0))
        if prob.E:
            s0.ech[:, :, :B].copy_(data["initial_echelon_inventories"].permute(1, 2, 0))

        if self.head == "data_driven":
            self._fill_observation_rows(data, prob, T, B, ld, shift, observation_params, demand_soa)
        self._ub_now = self._ub() if self.head not in ("softplus", "data_driven") else 0.0
        self._ctx = (prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa)
        probe = None
        if self.use_graph == "auto" and self._auto_graph is None and train and self._eager_runs >= 1 and self.timer is None:
            import time
            probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            torch.cuda.synchronize()
            probe[0].record()
            probe_t0 = time.perf_counter()
        self._replay_or_capture("fwd", self._launch_forward)
        total = self.rewards.sum()
        reported = self.rewards[ignore_periods:].sum() if ignore_periods else total
        if not train:
            self._eager_runs += 1
            return total, reported

        if grad_scale is None:
            grad_scale = 1.0 / (B * T * self.problem_params["n_stores"])
        self._set_g_reward(B, grad_scale)
        tail = self._use_tail_bwd()
        for i, sl in enumerate(self.slabs):
            if not (tail and i == len(self.slabs) - 1):
                sl.zero_()
            elif sl.shape[0] > self._tail_slots:   # (the fused tail's first launch of a sweep overwrites the slots it uses; the
                sl[self._tail_slots:].zero_()       # rest - the other backward's splits - must not carry an earlier run's sums)
        if not tail:   # (the fused tail reads no gradient of the state after the last period)
            self.g_state[0].zero_()
        # layers whose backward is ONE fused pass over their input (nic_linear_bwd_thin): thin output, not the first
        thin = [i > 0 and self.use_thin and ops.linear_bwd_thin_ok(self.gd[i + 1], self.gd[i]) for i in range(len(lins))]
        if thin != getattr(self, "_thin", None):
            self._thin = thin
            self._graphs.pop("bwd", None)  # a captured launch sequence no longer applies
        self._replay_or_capture("bwd", self._launch_backward)
        if probe is not None:   # host time to enqueue the step against GPU time to run it (one synchronisation, once per shape)
            host_ms = (time.perf_counter() - probe_t0) * 1e3
            probe[1].record()
            probe[1].synchronize()
            gpu_ms = probe[0].elapsed_time(probe[1])
            self._auto_graph = host_ms > 0.85 * gpu_ms
            self.auto_graph_probe = {"host_enqueue_ms": host_ms, "gpu_ms": gpu_ms, "replay": self._auto_graph}
        self._eager_runs += 1
        if assign_grads:
            self._assign_grads(accumulate_grads)
        return total, reported

    def input_rows(self, data, observation_params=None):
        """Rows of the policy's input for batches shaped like `data` (what `materialize` wants before the first run)."""
        n = lambda k: (data[k].shape[1] * data[k].shape[2]) if k in data else 0  # noqa: E731
        rows = n("initial_inventories")
        if self.head != "softplus":
            rows += n("initial_warehouse_inventories") + n("initial_echelon_inventories")
        if self.head == "data_driven":
            S = data["initial_inventories"].shape[1]
            rows += (S * observation_params["demand"]["past_periods"] + 2 * S + data["days_from_christmas"].shape[1]
                     + S * data["lead_times"].shape[2])
        return rows

    def _fill_observation_rows_by_row(self, X, data, prob, T, B, ld, shift, observation_params, demand_soa):
        """The same rows as `_fill_observation_rows` in the whole-horizon route's [row][T][ld] order (X: [F][T][ld])."""
        S, P_ = prob.S, observation_params["demand"]["past_periods"]
        o = self.F_dyn
        n_t = demand_soa.shape[0]
        if getattr(self, "_dpad", None) is None or self._dpad.shape != (P_ + n_t, S, ld):
            self._dpad = torch.zeros(P_ + n_t, S, ld, device=self.device)
        self._dpad[P_:].copy_(demand_soa)
        # row (s, p) of period t = demand of period t + shift - P + p = padded row t + shift + p
        win = self._dpad.as_strided((S, P_, T, ld), (ld, S * ld, S * ld, 1), shift * S * ld)
        X[o:o + S * P_].view(S, P_, T, ld).copy_(win)
        o += S * P_
        for k in ("underage_costs", "holding_costs"):
            X[o:o + S, :, :B] = data[k].t().unsqueeze(1)
            o += S
        dfc = data["days_from_christmas"]                      # [B][D][periods of the data]
        D = dfc.shape[1]
        idx = torch.clamp(torch.arange(T, device=self.device) + shift, max=dfc.shape[2] - 1)
        X[o:o + D, :, :B] = dfc[:, :, idx].permute(1, 2, 0)
        o += D
        lt = data["lead_times"]                                 # [B][S][W]
        X[o:o + lt.shape[1] * lt.shape[2], :, :B] = lt.reshape(B, -1).t().unsqueeze(1)
        assert o + lt.shape[1] * lt.shape[2] == self.F

    def _fill_observation_rows(self, data, prob, T, B, ld, shift, observation_params, demand_soa):
        """data_driven: the observation rows behind the state rows of every period's input block, in the reference's
        concatenation order (neural_networks.py:452-470): past demands [S][P] (the window demands[t+shift-P : t+shift], zeros to
        the left of period 0: environment.py:436-458), underage [S], holding [S], days_from_christmas [D] (column
        min(t+shift, last): :460-468), lead times [S][W].  A handful of strided copies for ALL periods, outside the period loop."""
        S, P_ = prob.S, observation_params["demand"]["past_periods"]
        X = self.states                       # [T+1][F][ld]
        o = self.F_dyn
        n_t = demand_soa.shape[0]
        if getattr(self, "_dpad", None) is None or self._dpad.shape != (P_ + n_t, S, ld):
            self._dpad = torch.zeros(P_ + n_t, S, ld, device=self.device)
        self._dpad[P_:].copy_(demand_soa)      # P zero periods in front: the window of period t is rows [t+shift, t+shift+P)
        win = self._dpad.as_strided((T, S, P_, ld), (S * ld, ld, S * ld, 1), shift * S * ld)
        X[:T, o:o + S * P_].view(T, S, P_, ld).copy_(win)
        o += S * P_
        for k in ("underage_costs", "holding_costs"):
            X[:T, o:o + S, :B] = data[k].t()
            o += S
        dfc = data["days_from_christmas"]                      # [B][D][periods of the data]
        D = dfc.shape[1]
        idx = torch.clamp(torch.arange(T, device=self.device) + shift, max=dfc.shape[2] - 1)
        X[:T, o:o + D, :B] = dfc[:, :, idx].permute(2, 1, 0)
        o += D
        lt = data["lead_times"]                                 # [B][S][W]
        X[:T, o:o + lt.shape[1] * lt.shape[2], :B] = lt.reshape(B, -1).t()
        assert o + lt.shape[1] * lt.shape[2] == self.F

    def param_grads(self):
        """[(parameter, gradient buffer of the last training run)] - engine-owned buffers, overwritten by the next run."""
        out = []
        for i, m in enumerate(self._linears()):
            out.append((m.weight, self.gw[i]))
            if m.bias is not None:
                out.append((m.bias, self.gb[i]))
        return out

    def _cost_totals(self, ignore_periods):
        """(total, reported) = sums of the per-period costs [T][ldb] (all periods / periods >= ignore_periods, trainer.py:207-210)
        by `nic_small_rollout_reduce`: two launches, fixed order, no semaphore pass."""
        need = sr.small_rollout_reduce_scratch(0, 0, self.rewards.numel())
        if getattr(self, "sr_scratch", None) is None or self.sr_scratch.numel() < need:
            self.sr_scratch, self.sr_totals = torch.empty(need, device=self.device), torch.zeros(2, device=self.device)
        sr.small_rollout_reduce(None, 0, None, self.rewards, ignore_periods, self.sr_totals, self.sr_scratch)
        tt = self.sr_totals.clone()   # (the caller's tensors must not change under a later step)
        return tt[0], tt[1]

    def _copy_if_changed(self, slot, dst, src):
        """dst <- src.permute(1, 2, 0); with `inputs_versioned` (opt-in: bench.py, whose batch nobody writes) the copy is skipped
        when `src` is the very tensor (same object, same in-place version) that was copied into the same destination last time.
        torch's version counter does not see every write - raw-pointer kernels (the HIP sampler), `.data` assignments, `set_()` -
        so by default every presented batch is copied: one small launch."""
        if not getattr(self, "inputs_versioned", False):
            dst.copy_(src.permute(1, 2, 0))
            return
        cache = self.__dict__.setdefault("_copied", {})
        key = (dst.data_ptr(), tuple(dst.shape), src._version)
        hit = cache.get(slot)
        if hit is not None and hit[0] is src and hit[1] == key:
            return
        dst.copy_(src.permute(1, 2, 0))
        cache[slot] = (src, key)

    def _set_g_reward(self, B, grad_scale):
        """g_reward[b] = d loss / d reward[b, t] for the live scenarios, 0 in the padding columns.  Two launches that a training
        loop repeats with the same numbers every step: skipped while the buffer, the batch size and the (host-side) scale are the
        ones of the previous call - the whole-horizon routes' steps are a handful of launches, these were two of them."""
        key = (self.g_reward.data_ptr(), B, grad_scale) if isinstance(grad_scale, (int, float)) else None
        if key is not None and key == getattr(self, "_g_reward_key", None):
            return
        self.g_reward.zero_()
        self.g_reward[:B] = grad_scale
        self._g_reward_key = key

    def _assign_grads(self, accumulate):
        for p, g in self.param_grads():
            if accumulate and p.grad is not None and p.grad is not g:
                p.grad.add_(g)
            else:
                p.grad = g

    def _problem_for(self, data):
        """EnvProblem of a batch (cached per presented tensors, see layout.ProblemCache)."""
        return self._prob_cache.get(self.problem_params, data, self.device)

    # ---- whole-horizon route for the small policies -------------------------------------------------------------------
    def _run_small(self, data, prob, T, B, ld, shift, demand_soa, ignore_periods, train, grad_scale, accumulate_grads,
                   assign_grads=True):
        plan, lins = self.small, self._linears()
        sr.pack_weights(lins, self.sr_weights)
        s0 = self._views(self.sr_state0, prob)
        self._copy_if_changed("s0_store", s0.store[:, :, :B], data["initial_inventories"])
        if prob.Wn:
            self._copy_if_changed("s0_wh", s0.wh[:, :, :B], data["initial_warehouse_inventories"])
        if prob.E:
            self._copy_if_changed("s0_ech", s0.ech[:, :, :B], data["initial_echelon_inventories"])
        ub = self._ub() if self.head != "softplus" else 0.0
        # Scenarios per wavefront.  Training: 16 (v_mfma_f32_16x16x4_f32, wave-native activation history) - measured against 32:
        # cfg1 0.55 -> 0.30 ms, cfg4 (16,384 scenarios: 32 leaves half the SIMDs without a wavefront) 0.83 -> 0.70 ms, cfg2
        # (32,768) 1.16 -> 1.07 ms, 65,536 scenarios 2.28 -> 1.87 ms.  Evaluation (no history): 16 while 32 would leave SIMDs
        # idle, else 32 (the per-lane head / env-step code is replicated in four lane groups instead of two).  The dz-history
        # sweep (small_wgrad_in_kernel = False) only exists in the 32-wide form.
        width = self.small_lane_scenarios or (16 if (train or B <= 16384) else 32)
        if train and not self.small_wgrad_in_kernel:
            width = 32
        desc = plan.desc(T, shift, self.sr_weights, demand_soa, self.sr_state0, ub, round_orders=self._round, prob=prob,
                         lane_scenarios=width)
        hist = (self.sr_states, self.sr_hidden, self.sr_logits) if train else (None, None, None)
        self._k("small_rollout_fwd", sr.small_rollout_fwd, desc, self.rewards, self.sr_final, *hist)
        if train and self.small_wgrad_in_kernel:
            total = reported = None   # (summed together with the partial gradients after the backward launch)
        else:   # the same reduction, costs only: an evaluation pass returns the very bits a training pass does
            total, reported = self._cost_totals(ignore_periods)
        if not train:
            return total, reported
        if grad_scale is None:
            grad_scale = 1.0 / (B * T * self.problem_params["n_stores"])
        self._set_g_reward(B, grad_scale)
        if self.small_wgrad_in_kernel:
            self._k("small_rollout_bwd", sr.small_rollout_bwd_wgrad, desc, *hist, Table(self.g_reward, 0, 1), self.sr_slab)
            # one partial gradient per wavefront: only the rows THIS width's launch wrote are summed (the slab is sized for the
            # 16-wide form; a 32-wide launch fills half of it and must not pick up an earlier 16-wide run's rows)
            # ... and the step's two sums - partial gradients over the wavefronts, costs over (period, scenario) - in two small
            # launches with a fixed order (csrc/small_reduce.hip).  As torch reductions these were four launches, 37-46 us of a
            # 1-ms step (the row sums end in a semaphore pass with its own memset).
            n_rows = (B + width - 1) // width
            need = sr.small_rollout_reduce_scratch(n_rows, self.sr_grad.numel(), self.rewards.numel())
            if getattr(self, "sr_scratch", None) is None or self.sr_scratch.numel() < need:
                self.sr_scratch, self.sr_totals = torch.empty(need, device=self.device), torch.zeros(2, device=self.device)
            self._k("small_rollout_reduce", sr.small_rollout_reduce, self.sr_slab, n_rows, self.sr_grad, self.rewards, ignore_periods,
                    self.sr_totals, self.sr_scratch)
            tt = self.sr_totals.clone()   # (the caller's tensors must not change under a later step)
            total, reported = tt[0], tt[1]
            if assign_grads:
                self._assign_grads(accumulate_grads)
            return total, reported
        self._k("small_rollout_bwd", sr.small_rollout_bwd, desc, *hist, Table(self.g_reward, 0, 1), self.sr_dzh, self.sr_dzo)
        # weight gradients: contraction over (period, scenario) = T*ld columns; padding columns of dZ are zero
        n_cols, nh = T * ld, plan.n_hidden
        inputs = [self.sr_states[:self.dims[0]]] + [self.sr_hidden[sr.H * l:sr.H * (l + 1)] for l in range(nh)]
        dzs = [self.sr_dzh[sr.H * l:sr.H * (l + 1)] for l in range(nh)] + [self.sr_dzo]
        for i, m in enumerate(lins):
            self.slabs[i].zero_()
            dy, x = dzs[i].reshape(dzs[i].shape[0], n_cols), inputs[i].reshape(inputs[i].shape[0], n_cols)
            self._k(f"wgrad_{self.dims[i + 1]}x{self.dims[i]}", ops.linear_wgrad, dy, x, self.slabs[i], n_cols)
            ops.wgrad_reduce(self.slabs[i], self.gw[i], self.gb[i], self.dims[i], 1.0)
        if assign_grads:
            self._assign_grads(accumulate_grads)
        return total, reported

    # ---- whole-horizon route for data_driven on small batches ------------------------------------------------------------
    def _run_horizon(self, data, prob, T, B, ld, shift, demand_soa, ignore_periods, train, grad_scale, accumulate_grads,
                     assign_grads, observation_params):
        plan, lins = self.horizon, self._linears()
        F, FD, dims = self.F, self.F_dyn, self.dims
        Fo, n_cols = F - FD, T * ld
        a = self.F_store
        self.hz_state0[:a].view(prob.S, prob.Ws, ld)[:, :, :B].copy_(data["initial_inventories"].permute(1, 2, 0))
        if prob.Wn:
            self.hz_state0[a:].view(prob.Wn, prob.Ww, ld)[:, :, :B].copy_(data["initial_warehouse_inventories"].permute(1, 2, 0))
        X = self.hz_X
        self._fill_observation_rows_by_row(X, data, prob, T, B, ld, shift, observation_params, demand_soa)
        # the observation rows' share of the first layer for every period at once (+ bias): independent of the rollout
        self.hz_W1obs[:, :Fo].copy_(lins[0].weight.detach()[:, FD:])
        self._k(f"fwdT_{dims[1]}x{Fo}", ops.linear_fwd, self.hz_W1obs[:, :Fo], lins[0].bias.detach(), X[FD:].view(Fo, n_cols),
                self.hz_z1.view(dims[1], n_cols), n_cols, _lib.NIC_ACT_NONE)
        desc = plan.desc(prob, T, shift, lins, self.edge_mask, demand_soa, n_cols, round_orders=self._round)
        if getattr(self, "_hz_checked", None) != self._key:   # once per shape: the library's own limits
            if not hz.horizon_ok(desc):
                raise _HorizonRefused()
            self._hz_checked = self._key
        hist = ([X] + self.hz_hist) if train else [None] * 5
        self._k("horizon_fwd", hz.horizon_fwd, desc, self.hz_z1, self.hz_state0, self.rewards, self.hz_final, *hist)
        total = self.rewards.sum()
        reported = self.rewards[ignore_periods:].sum() if ignore_periods else total
        if not train:
            return total, reported
        if grad_scale is None:
            grad_scale = 1.0 / (B * T * self.problem_params["n_stores"])
        self._set_g_reward(B, grad_scale)
        self._k("horizon_bwd", hz.horizon_bwd, desc, *hist, Table(self.g_reward, 0, 1), *self.hz_dz)
        # weight gradients: contractions over (period, scenario) = T * ld columns; padding columns of the dz histories are zero
        inputs = [X, self.hz_hist[0], self.hz_hist[1]]
        for i in range(len(lins)):
            self.slabs[i].zero_()
            dy, x = self.hz_dz[i].view(dims[i + 1], n_cols), inputs[i].view(dims[i], n_cols)
            self._k(f"wgradT_{dims[i + 1]}x{dims[i]}", ops.linear_wgrad, dy, x, self.slabs[i], n_cols)
            ops.wgrad_reduce(self.slabs[i], self.gw[i], self.gb[i], dims[i], 1.0)
        if assign_grads:
            self._assign_grads(accumulate_grads)
        return total, reported

    # ---- launch sequences (eager, or captured once into a HIP graph and replayed) ---------------------------------
    def _replay_or_capture(self, name, fn):
        """`use_graph`: the launch sequence of a rollout is identical from call to call (same buffers, same shapes), so it
        is captured once into a HIP graph and replayed — this removes the per-launch host cost that bounds the small
        (32-wide) policies, whose kernels run for a few microseconds each.  The first call always runs eagerly (module
        loading is not capturable); timers force eager mode."""
        if not self._graph_on() or self.timer is not None or self._eager_runs < 1:
            return fn()
        variant = (self._round, self._ctx[4], self.fuse_head_env, self.fuse_tail, self.use_wide, self._use_tail_bwd())  # options baked into the captured launch sequence
        if getattr(self, "_graph_variant", variant) != variant:
            self._graphs = {}
        self._graph_variant = variant
        g = self._graphs.get(name)
        if g is None:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                fn()
            self._graphs[name] = g
        g.replay()

    def _launch_forward(self):
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        ub = self._ub_now
        hist = self._hist
        self._thin_in = (self.use_thin and L > 1 and biases[0] is not None and ops.linear_fwd_thin_in_ok(self.gd[1], self.gd[0])
                         and 4 * self.gd[1] * ld < 2 ** 31)
        # the ones row sits right behind the MLP's input rows when the input IS the whole state block (every head of this engine)
        self._thin_in_aug = (self._thin_in and self.F + 1 == self.states.shape[1]
                             and ops.linear_fwd_thin_in_ok(self.gd[1], self.gd[0] + 1))
        if self._use_wide():
            return self._launch_forward_wide()
        if self._use_tail():
            return self._launch_forward_tail()
        for t in range(T):
            cur, nxt, row = (t, t + 1, t) if hist else (t & 1, (t + 1) & 1, 0)
            st = self._views(self.states[cur], prob)
            x = self.states[cur][:self.F]
            hs = t if train else 0
            for i in range(L - 1):
                y = self.hidden[i][hs]
                if i == 0 and self._thin_in and self._thin_in_aug:   # ... with the bias inside the contraction: K + 1 rows, no bias
                    self._k(f"fwd_{self.gd[1]}x{self.gd[0]}", ops.linear_fwd_thin_in, self.Wt[0][:self.F + 1, :self.gd[1]], None,
                            self.states[cur][:self.F + 1], y, B, _lib.NIC_ACT_ELU)
                elif i == 0 and self._thin_in:   # short contraction, many rows: the write-bound streamed forward (thin_layer.hip)
                    self._k(f"fwd_{self.gd[1]}x{self.gd[0]}", ops.linear_fwd_thin_in, Wtv[0], biases[0], x, y, B, _lib.NIC_ACT_ELU)
                else:
                    self._k(f"fwd_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_fwd, Wv[i], biases[i], x, y, B, _lib.NIC_ACT_ELU)
                x = y
            Z = self.logits[row]
            rows_kw = {}
            if self.live_rows is None:
                self._k(f"fwd_{self.gd[L]}x{self.gd[L - 1]}", ops.linear_fwd, Wv[L - 1], biases[L - 1], x, Z, B, _lib.NIC_ACT_NONE)
            elif self._fused_head_env(prob):
                # compact logits straight into the first rows of the period's logits block; the fused head + env launch reads
                # them through the row map (no scatter into the head's [S * Wn + Wn] layout: one launch and 38 MB per period at
                # cfg5)
                Z = Z[:self.gd[L]]
                self._k(f"fwd_{self.gd[L]}x{self.gd[L - 1]}", ops.linear_fwd, Wv[L - 1], biases[L - 1], x, Z, B, _lib.NIC_ACT_NONE)
                rows_kw = dict(logit_rows=self.zrow, first_wh_row=self.first_wh_row)
            else:   # compact logits, then one row copy into the head's layout (rows of pairs without an edge stay 0: never read)
                self._k(f"fwd_{self.gd[L]}x{self.gd[L - 1]}", ops.linear_fwd, Wv[L - 1], biases[L - 1], x, self.Zc, B,
                        _lib.NIC_ACT_NONE)
                Z.index_copy_(0, self.live_rows, self.Zc)
            so, wo, eo = self._order_views(self.orders[row], prob)
            if self._fused_head_env(prob):   # head + env step in one launch (the orders still land in self.orders[row])
                ts, tw, _ = self._order_tables(self.orders[row], prob)
                self._k("head_env_fwd", ops.head_env_fwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, Z, self.adj, ub,
                        bool(self.model.transshipment), self._views(self.states[nxt], prob), self.rewards[t], **rows_kw)
                continue
            if self.head == "warehouse":
                self._k("head_fwd", ops.head_warehouse_fwd, Z, st.wh, self.adj, ub, bool(self.model.transshipment), so, wo,
                        prob.S, prob.Wn, prob.Ww, B)
            elif self.head == "serial":
                self._k("head_fwd", ops.head_serial_fwd, Z, st.wh, st.ech, ub, so, wo, eo, prob.E, prob.Ww, prob.We, B)
            elif self.head == "data_driven":
                self._k("head_fwd", ops.head_data_driven_fwd, Z, st.wh, self.edge_mask if prob.Wn else None, so, wo, prob.S,
                        prob.Wn, prob.Ww, B)
            else:
                self._k("head_fwd", ops.head_softplus_fwd, Z, so.view(-1, ld), prob.S * prob.nsup, B)
            if self._round:
                ops.round_orders(self.orders[row], B)  # discrete allocation (trainer.py:201-202)
            ts, tw, te = self._order_tables(self.orders[row], prob)
            self._k("env_fwd", ops.env_step_fwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, te,
                    out=self._views(self.states[nxt], prob), reward=self.rewards[t])

    def _tail_desc(self, t, state_block, orders_block):
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        return ops.period_tail_desc(prob, state_block, Table(demand_soa[t + shift], ld, 1), orders_block, self.adj, self._ub_now,
                                    bool(self.model.transshipment), Wv[L - 1], biases[L - 1], self.Wt[0][:self.F + 1, :self.gd[1]])

    def _launch_forward_wide(self):
        """Forward sweep of the wide policy in ONE launch (csrc/wide_rollout.hip); the histories it leaves are the per-period
        route's (states, orders, logits, hidden activations, rewards), so the backward sweep is unchanged."""
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        lins = self._linears()
        for l in range(1, L - 1):
            ops.wide_pack_hidden(lins[l].weight, self.Wpk[l - 1])
        self._wq_pad[:self.dims[-1]].copy_(lins[L - 1].weight.detach())
        rows, cols = self._wq_index
        self.Wq.copy_(self._wq_pad[rows, cols])
        self._k("wide_fwd", ops.wide_rollout_fwd, self._wide_desc(train))

    def _wide_desc(self, with_hidden):
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        return ops.wide_rollout_desc(prob, T, self.adj, self._ub_now, bool(self.model.transshipment), demand_soa, shift, self.states,
                                     self.orders, self.logits, self.rewards, self.hidden if with_hidden else None,
                                     self.Wt[0][:self.F + 1, :self.gd[1]], self.Wpk, [biases[l] for l in range(1, L - 1)], self.Wq,
                                     biases[L - 1])

    def _launch_backward_wide(self):
        """Backward sweep of the wide policy: ONE launch walks the periods in reverse over the histories the forward kernel left and
        writes every layer's pre-activation gradient history; the weight gradients are then one (period x scenario) contraction
        per layer, as on the per-period route."""
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        lins = self._linears()
        for l in range(1, L - 1):
            ops.wide_pack_hidden(lins[l].weight.detach().t(), self.WpkT[l - 1])
        self._win_pad[:, :self.F].copy_(lins[0].weight.detach())
        unit, f = self._win_index
        Wq_in = self._win_pad[unit, f]
        self._wot_pad[:self.dims[-1]].copy_(lins[L - 1].weight.detach())
        row, col = self._wot_index
        Wo_t = self._wot_pad[row, col]
        self._wide_keep = (Wq_in, Wo_t)
        desc = self._wide_desc(True)
        self._k("wide_bwd", ops.wide_rollout_bwd, desc, Table(self.g_reward, 0, 1), self.dZhist, self.dZlast_hist, self.WpkT, Wq_in, Wo_t)
        for i in range(L):
            x_hist = self.hidden[i - 1] if i > 0 else self.states[:T, :self.F]
            dz_hist = self.dZhist[i] if i < L - 1 else self.dZlast_hist
            self._k(f"wgradT_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_wgrad_periods, dz_hist, x_hist, self.slabs[i], B)
        for i in range(L):
            ops.wgrad_reduce(self.slabs[i], self.gw[i], self.gb[i], self.gd[i], 1.0)

    def _launch_forward_tail(self):
        """Forward sweep with the fused tail: [first layer of period 0], then per period the hidden-layer GEMMs and ONE tail launch
        (logits + head + env step + first layer of period t+1) - L - 1 launches per period instead of L + 1."""
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        hist = self._hist
        if self._thin_in and self._thin_in_aug:   # first layer of period 0 (later periods': the tail launch of the period before)
            self._k(f"fwd_{self.gd[1]}x{self.gd[0]}", ops.linear_fwd_thin_in, self.Wt[0][:self.F + 1, :self.gd[1]], None,
                    self.states[0][:self.F + 1], self.hidden[0][0], B, _lib.NIC_ACT_ELU)
        else:
            self._k(f"fwd_{self.gd[1]}x{self.gd[0]}", ops.linear_fwd, Wv[0], biases[0], self.states[0][:self.F], self.hidden[0][0], B,
                    _lib.NIC_ACT_ELU)
        for t in range(T):
            cur, nxt, row = (t, t + 1, t) if hist else (t & 1, (t + 1) & 1, 0)
            hs, hs_next = (t, t + 1) if train else (0, 0)
            x = self.hidden[0][hs]
            for i in range(1, L - 1):
                y = self.hidden[i][hs]
                self._k(f"fwd_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_fwd, Wv[i], biases[i], x, y, B, _lib.NIC_ACT_ELU)
                x = y
            desc = self._tail_desc(t, self.states[cur], self.orders[row])
            self._k("tail_fwd", ops.period_tail_fwd, desc, x, self.logits[row], self.states[nxt], self.rewards[t],
                    self.hidden[0][hs_next] if t + 1 < T else None)

    def _launch_backward_tail(self):
        """Backward sweep with the fused tail: per period ONE tail launch (first layer's input gradient of period t+1 + env / head
        adjoints + logits layer backward with its weight gradient) and the hidden layers' input gradients."""
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        g_next, g_cur = self.g_state
        hist = self.dZhist
        g_rew = Table(self.g_reward, 0, 1)
        dz1_buf = lambda t: hist[0][t] if hist is not None else self.dH[1 & 1][:self.gd[1]]  # noqa: E731
        for t in range(T - 1, -1, -1):
            desc = self._tail_desc(t, self.states[t], self.orders[t])
            dx = hist[L - 2][t] if hist is not None else self.dH[(L - 1) & 1][:self.gd[L - 1]]
            last = t == T - 1
            self._k("tail_bwd", ops.period_tail_bwd, desc, self.logits[t], self.hidden[L - 2][t], None if last else dz1_buf(t + 1),
                    None if last else g_next, g_rew, g_cur, dx, self.slabs[L - 1], last)
            d = dx
            for i in range(L - 2, -1, -1):
                x_in = self.hidden[i - 1][t] if i > 0 else self.states[t][:self.F]
                if hist is None:
                    self._k(f"wgrad_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_wgrad, d, x_in, self.slabs[i], B)
                if i > 0:
                    dxi = hist[i - 1][t] if hist is not None else self.dH[i & 1][:self.gd[i]]
                    self._k(f"dgrad_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_dgrad, Wtv[i], d, x_in, dxi, B, _lib.NIC_ACT_ELU, False)
                    d = dxi
                # (i == 0: the first layer's input gradient is the first stage of the NEXT tail launch; period 0's is not needed)
            g_next, g_cur = g_cur, g_next
        if hist is not None:
            for i in range(L - 1):
                x_hist = self.hidden[i - 1] if i > 0 else self.states[:T, :self.F]
                self._k(f"wgradT_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_wgrad_periods, hist[i], x_hist, self.slabs[i], B)
        for i in range(L):
            ops.wgrad_reduce(self.slabs[i], self.gw[i], self.gb[i], self.gd[i], 1.0)

    def _launch_backward(self):
        prob, T, B, ld, shift, train, Wv, Wtv, biases, L, demand_soa = self._ctx
        if self._use_wide() and getattr(self, "_wide_bwd", False):
            return self._launch_backward_wide()
        if self._use_tail_bwd():
            return self._launch_backward_tail()
        ub = self._ub_now
        g_next, g_cur = self.g_state
        detached_input = self.head == "serial"  # the reference detaches VanillaSerial's MLP input (:329)
        for t in range(T - 1, -1, -1):
            st = self._views(self.states[t], prob)
            ts, tw, te = self._order_tables(self.orders[t], prob)
            gso, gwo, geo = self._order_views(self.g_orders, prob)
            fused = self._fused_head_env(prob)
            if not fused:
                self._k("env_bwd", ops.env_step_bwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, te,
                        self._views(g_next, prob), Table(self.g_reward, 0, 1), g_in=self._views(g_cur, prob),
                        g_orders=(gso, gwo, geo))
            Z = self.logits[t]
            gc = self._views(g_cur, prob)
            hist, last_hist = self.dZhist, self.dZlast_hist
            compact = self.live_rows is not None
            dZ = last_hist[t] if (last_hist is not None and not compact) else self.dZ
            if fused and compact:   # ... on the compact logits: the gradient of the live rows lands where the GEMMs read it
                dZ = last_hist[t] if last_hist is not None else self.dZc
                self._k("head_env_bwd", ops.head_env_bwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, Z[:self.gd[L]],
                        self.adj, ub, bool(self.model.transshipment), self._views(g_next, prob), Table(self.g_reward, 0, 1), gc,
                        (gso, gwo), dZ, logit_rows=self.zrow, first_wh_row=self.first_wh_row)
                compact = False   # (nothing left to gather)
            elif fused:   # env-step adjoint + head adjoint in one launch
                self._k("head_env_bwd", ops.head_env_bwd, prob, st, Table(demand_soa[t + shift], ld, 1), ts, tw, Z, self.adj, ub,
                        bool(self.model.transshipment), self._views(g_next, prob), Table(self.g_reward, 0, 1), gc, (gso, gwo), dZ)
            elif self.head == "warehouse":
                self._k("head_bwd", ops.head_warehouse_bwd, Z, st.wh, self.adj, ub, bool(self.model.transshipment), gso, gwo,
                        dZ, gc.wh, prob.S, prob.Wn, prob.Ww, B)
            elif self.head == "serial":
                self._k("head_bwd", ops.head_serial_bwd, Z, st.wh, st.ech, ub, gso, gwo, geo, dZ, gc.wh, gc.ech, prob.E,
                        prob.Ww, prob.We, B)
            elif self.head == "data_driven":
                self._k("head_bwd", ops.head_data_driven_bwd, Z, st.wh, self.edge_mask if prob.Wn else None, gso, gwo, dZ,
                        gc.wh, prob.S, prob.Wn, prob.Ww, B)
            else:
                self._k("head_bwd", ops.head_softplus_bwd, Z, gso.view(-1, ld), dZ, prob.S * prob.nsup, B)
            d = dZ
            if compact:   # the live rows of the head's logits gradient (the others are exact zeros)
                d = last_hist[t] if last_hist is not None else self.dZc
                torch.index_select(dZ, 0, self.live_rows, out=d)
            for i in range(L - 1, -1, -1):
                x_in = self.hidden[i - 1][t] if i > 0 else self.states[t][:self.F]
                # gradient wrt the previous layer's pre-activation output: kept for every period when its weight gradient
                # is contracted at the end of the sweep, else a ping-pong scratch buffer
                dx = (hist[i - 1][t] if hist is not None else self.dH[i & 1][:self.gd[i]]) if i > 0 else None
                if i > 0 and self._thin[i]:
                    # thin (logits) layer: weight gradient and input gradient in ONE pass over the layer's input
                    self._k(f"bwd_thin_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_bwd_thin, Wv[i], d, x_in, dx,
                            self.slabs[i], B, _lib.NIC_ACT_ELU)
                    d = dx
                    continue
                if hist is None or (i == L - 1 and last_hist is None):  # (else: contracted over all periods after the sweep)
                    self._k(f"wgrad_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_wgrad, d, x_in, self.slabs[i], B)
                if i > 0:
                    self._k(f"dgrad_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_dgrad, Wtv[i], d, x_in, dx, B,
                            _lib.NIC_ACT_ELU, False)
                    d = dx
                elif not detached_input:   # (only the state rows of the input carry a gradient back in time)
                    self._k(f"dgrad_{self.gd[1]}x{self.gd[0]}", ops.linear_dgrad, Wtv[0][:self.F_dyn], d, None,
                            g_cur[:self.F_dyn], B, _lib.NIC_ACT_NONE, True)
            g_next, g_cur = g_cur, g_next
        if self.dZhist is not None:
            # hidden layers: dW_i = sum over (period, scenario) of dZ_i X_i^T in one launch each
            for i in range(L):
                if (i > 0 and self._thin[i]) or (i == L - 1 and self.dZlast_hist is None):
                    continue  # its weight gradient was accumulated period by period (fused thin-layer backward / no history)
                x_hist = self.hidden[i - 1] if i > 0 else self.states[:T, :self.F]
                dz_hist = self.dZhist[i] if i < L - 1 else self.dZlast_hist
                self._k(f"wgradT_{self.gd[i + 1]}x{self.gd[i]}", ops.linear_wgrad_periods, dz_hist, x_hist,
                        self.slabs[i], B)
        for i in range(L):
            if i == L - 1 and self.live_rows is not None:   # compact logits layer: reduce, then scatter into the parameter's rows
                ops.wgrad_reduce(self.slabs[i], self.gw_c, self.gb_c, self.gd[i], 1.0)
                self.gw[i].zero_()
                self.gw[i].index_copy_(0, self.live_rows, self.gw_c)
                if self.gb[i] is not None:
                    self.gb[i].zero_()
                    self.gb[i].index_copy_(0, self.live_rows, self.gb_c)
                continue
            ops.wgrad_reduce(self.slabs[i], self.gw[i], self.gb[i], self.gd[i], 1.0)

    # ---- inspection helpers used by the parity tests --------------------------------------------------------------
    def per_period_rewards(self):
        return self.rewards[:, :self.prob.B]

    def final_state(self):
        from .layout import ref_view
        if self.small is not None:
            last = self.sr_final
        elif self.horizon is not None:
            last = self.hz_final
        else:  # history: block T; rolling evaluation: block T & 1
            last = self.states[-1] if self._hist else self.states[self._T_last & 1]
        st = self._views(last, self.prob)
        out = {"store_inventories": ref_view(st.store, self.prob.B)}
        if st.wh is not None:
            out["warehouse_inventories"] = ref_view(st.wh, self.prob.B)
        if st.ech is not None:
            out["echelon_inventories"] = ref_view(st.ech, self.prob.B)
        return out
