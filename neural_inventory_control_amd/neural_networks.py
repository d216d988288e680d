"""Policies of the hot path behind the reference's plugin contract (neural_networks.py:6-427, 1495-1574):
subclass `MyNeuralNetwork`, implement `forward(observation: dict) -> dict` of 3-D action tensors.

Same constructor arguments, same `state_dict` keys (`net.master.<i>.weight/bias`, so the reference's shipped
checkpoint loads), same attributes the trainer reads (`trainable`, `gradient_clipping_norm_value`,
`warehouse_upper_bound`).  What differs is what runs underneath: every Linear(+ELU) is `HipLinear`, an autograd
function over the FP32-MFMA kernels of csrc/linear_mfma.hip with feature-major activations, and the feasibility
heads (softmax share of warehouse stock, sigmoid x upstream on-hand, softplus) are the kernels of
csrc/policy_heads.hip.  `trainer.Trainer` additionally recognises these architectures and runs the whole horizon
through `rollout.FusedRollout` (no autograd graph at all); the modules below are the general path and the plugin base.

The closed-form policies (base_stock, capped_base_stock, echelon_stock) additionally expose `closed_form_levels()` for the
whole-horizon kernel of closed_form.py; `GNN` lives here too.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from .layout import pad_ld, ref_view, to_soa


def _as_feature_major(x):
    """logical (B, K) -> 2-D [K][ld] tensor the kernels can read (no copy if `x` already is a feature-major view)."""
    xt = x.t()
    if x.is_cuda and xt.stride(1) == 1 and xt.stride(0) % 4 == 0 and xt.stride(0) >= x.shape[0] \
            and xt.data_ptr() % 16 == 0 and x.dtype == torch.float32:
        return xt
    return to_soa(x.float())


class _LinearFunction(torch.autograd.Function):
    """y = act(x @ W^T + b) on the matrix cores; activations stay feature-major between layers."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        if not x.is_cuda:
            raise _lib.NicUnavailableError("HipLinear needs device tensors (no CPU fallback)")
        B = x.shape[0]
        X = _as_feature_major(x)
        N = weight.shape[0]
        Y = torch.empty(N, X.stride(0), device=x.device, dtype=torch.float32)
        if X.stride(0) > B:
            Y[:, B:].zero_()
        ops.linear_fwd(weight.detach().contiguous(), None if bias is None else bias.detach(), X, Y, B, act)
        ctx.act, ctx.B = act, B
        ctx.save_for_backward(X, Y, weight)
        ctx.has_bias = bias is not None
        return ref_view(Y, B)

    @staticmethod
    def backward(ctx, gy):
        X, Y, weight = ctx.saved_tensors
        B, act = ctx.B, ctx.act
        N, K = weight.shape
        ld = X.stride(0)
        if act == _lib.NIC_ACT_ELU:  # dZ = gY * elu'(z), recovered from the output (y > 0 ? 1 : y + 1)
            yv = ref_view(Y, B)
            gy = torch.where(yv > 0, gy, gy * (yv + 1))
        dZ = torch.zeros(N, ld, device=gy.device, dtype=torch.float32)
        dZ[:, :B] = gy.t()
        gx = gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            lds = (K + 4) // 4 * 4
            # the split count the kernels want for this shape (a GNN layer sees batch x edges = 10^5..10^6 columns: clamping
            # it left 4 workgroups on the chip), bounded only by 64 MB of slab
            splits = max(1, min(ops.wgrad_num_splits(N, K, B), (64 << 20) // (N * lds * 4)))
            slab = torch.zeros(splits, N, lds, device=gy.device)
            ops.linear_wgrad(dZ, X, slab, B)
            gw = torch.empty(N, K, device=gy.device)
            gb = torch.empty(N, device=gy.device)
            ops.wgrad_reduce(slab, gw, gb, K, 1.0)
            if not ctx.has_bias:
                gb = None
        if ctx.needs_input_grad[0]:
            dX = torch.zeros(K, ld, device=gy.device)
            ops.linear_dgrad(weight.detach().t().contiguous(), dZ, None, dX, B, _lib.NIC_ACT_NONE, False)
            gx = ref_view(dX, B)
        return gx, gw, gb, None


class _HipLinearForward:
    fused_act = _lib.NIC_ACT_NONE

    def forward(self, x):
        # while a compiler traces the policy the layer is the registered operator `nic::linear` (library.py: fake-tensor kernel +
        # autograd formula, so the traced graph survives); eager calls take the autograd.Function - same kernels, no dispatcher
        if torch.compiler.is_compiling():
            from . import library  # noqa: F401  (registers the operators)
            apply = lambda x2, w, b, act: torch.ops.nic.linear(x2, w, b, act)  # noqa: E731
        else:
            apply = _LinearFunction.apply
        if x.dim() == 1:  # closed-form policies feed a constant scalar input (neural_networks.py:228 in the reference)
            return apply(x.unsqueeze(0), self.weight, self.bias, self.fused_act).squeeze(0)
        if x.dim() > 2:  # (batch, nodes-or-edges, features): every leading index is one GEMM column (GNN policy)
            lead = x.shape[:-1]
            y = apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self.fused_act)
            return y.reshape(*lead, y.shape[-1])
        return apply(x, self.weight, self.bias, self.fused_act)


class HipLinear(_HipLinearForward, nn.Linear):
    """nn.Linear whose forward/backward run on csrc/linear_mfma.hip; optionally fuses the following ELU."""


class HipLazyLinear(_HipLinearForward, nn.LazyLinear):
    """LazyLinear (in_features inferred at the first forward, neural_networks.py:88) that becomes a HipLinear.
    torch binds `forward` before the materialising pre-hook swaps the class, so both classes share one forward."""
    cls_to_become = HipLinear


class _FusedELU(nn.ELU):
    """Keeps the reference's Sequential indices (activation modules sit at odd positions); the ELU itself was already
    applied in the epilogue of the preceding HipLinear."""

    def forward(self, x):
        return x


class MyNeuralNetwork(nn.Module):
    """Plugin base class.  Constructor contract: neural_networks.py:8-58 of the reference."""

    def __init__(self, args, device="cpu"):
        super().__init__()
        self.device = device
        self.trainable = True
        self.gradient_clipping_norm_value = args.get("gradient_clipping_norm_value", None)
        self.activation_functions = {
            "relu": nn.ReLU(), "elu": nn.ELU(), "tanh": nn.Tanh(), "softmax": nn.Softmax(dim=1),
            "softplus": nn.Softplus(), "sigmoid": nn.Sigmoid(),
        }
        self.warehouse_upper_bound = 0
        self.layers = {}
        self.nn_args = copy.deepcopy(args)
        self.net = self.create_module_dict(args)
        if args["initial_bias"] is not None:
            for key, val in args["initial_bias"].items():
                if val is not None:
                    pos = -2 if args["output_layer_activation"][key] else -1
                    self.initialize_bias(key, pos, val)

    def forward(self, observation):
        raise NotImplementedError

    def create_module_dict(self, args):
        return nn.ModuleDict({
            key: self.create_sequential_net(key, args["inner_layer_activations"][key], args["output_layer_activation"][key],
                                            args["neurons_per_hidden_layer"][key], args["output_sizes"][key])
            for key in args["output_sizes"]})

    def create_sequential_net(self, name, inner_layer_activations, output_layer_activation, neurons_per_hidden_layer,
                              output_size):
        """Sequential(LazyLinear, act, ..., Linear[, out_act]) exactly as neural_networks.py:80-106 lays it out."""
        layers = []
        fuse = inner_layer_activations == "elu"
        for width in neurons_per_hidden_layer:
            lin = HipLazyLinear(width)
            if fuse:
                lin.fused_act = _lib.NIC_ACT_ELU
                layers += [lin, _FusedELU()]
            else:
                layers += [lin, self.activation_functions[inner_layer_activations]]
        if len(neurons_per_hidden_layer) == 0:
            layers.append(HipLazyLinear(output_size))
        else:
            layers.append(HipLinear(neurons_per_hidden_layer[-1], output_size))
        if output_layer_activation is not None:
            layers.append(self.activation_functions[output_layer_activation])
        self.layers[name] = layers
        return nn.Sequential(*layers)

    def initialize_bias(self, key, pos, value):
        self.layers[key][pos].bias.data.fill_(value)

    # helpers kept for plugin authors (neural_networks.py:111-193)
    def apply_proportional_allocation(self, desired_allocations, available_inventory, transshipment=False):
        if available_inventory.dim() > 1:
            available_inventory = available_inventory.sum(dim=1)
        scaling = available_inventory / (desired_allocations.sum(dim=1) + 1e-10)
        if not transshipment:
            scaling = torch.clip(scaling, max=1.0)
        return desired_allocations * scaling[:, None]

    def flatten_then_concatenate_tensors(self, tensor_list, dim=1):
        return torch.cat([t.flatten(start_dim=dim) for t in tensor_list], dim=dim)

    def concatenate_signal_to_object_state_tensor(self, object_state, signal):
        return torch.cat((object_state, signal.unsqueeze(1).expand(-1, object_state.size(1), -1)), dim=2)

    def unpack_args(self, args, keys):
        return [args[k] for k in keys] if len(keys) > 1 else args[keys[0]]

    # ---- description consumed by rollout.FusedRollout -----------------------------------------------------------
    def master_linears(self):
        return [m for m in self.net["master"] if isinstance(m, nn.Linear)]


# ---- heads as autograd functions over csrc/policy_heads.hip --------------------------------------------------------

class _SoftplusHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        B, rows = z.shape
        Z = _as_feature_major(z)
        out = torch.zeros(rows, Z.stride(0), device=z.device)
        ops.head_softplus_fwd(Z, out, rows, B)
        ctx.save_for_backward(Z)
        ctx.B = B
        return ref_view(out, B)

    @staticmethod
    def backward(ctx, g):
        (Z,) = ctx.saved_tensors
        G = to_soa(g, Z.stride(0))
        dZ = torch.zeros_like(G)
        ops.head_softplus_bwd(Z, G, dZ, Z.shape[0], ctx.B)
        return ref_view(dZ, ctx.B)


class _WarehouseHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, wh_inv, adjacency, ub, transshipment, S, Wn):
        B = z.shape[0]
        Z = _as_feature_major(z)
        ld = Z.stride(0)
        Wh = to_soa(wh_inv, ld)
        Ww = wh_inv.shape[2]
        so = torch.zeros(S, Wn, ld, device=z.device)
        wo = torch.zeros(Wn, ld, device=z.device)
        ops.head_warehouse_fwd(Z, Wh, adjacency, ub, transshipment, so, wo, S, Wn, Ww, B)
        ctx.save_for_backward(Z, Wh, adjacency)
        ctx.meta = (B, ub, transshipment, S, Wn, Ww)
        return ref_view(so, B), ref_view(wo, B).unsqueeze(2)

    @staticmethod
    def backward(ctx, g_so, g_wo):
        Z, Wh, adjacency = ctx.saved_tensors
        B, ub, trans, S, Wn, Ww = ctx.meta
        ld = Z.stride(0)
        Gs, Gw = to_soa(g_so, ld), to_soa(g_wo[:, :, 0], ld)
        dZ = torch.zeros(S * Wn + Wn, ld, device=Z.device)
        g_wh = torch.zeros(Wn, Ww, ld, device=Z.device)
        ops.head_warehouse_bwd(Z, Wh, adjacency, ub, trans, Gs, Gw, dZ, g_wh, S, Wn, Ww, B)
        return ref_view(dZ, B), ref_view(g_wh, B), None, None, None, None, None


class _SerialHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, wh_inv, ech_inv, ub):
        B = z.shape[0]
        Z = _as_feature_major(z)
        ld = Z.stride(0)
        Wh, Ech = to_soa(wh_inv, ld), to_soa(ech_inv, ld)
        E, Ww, We = ech_inv.shape[1], wh_inv.shape[2], ech_inv.shape[2]
        so, wo, eo = (torch.zeros(1, 1, ld, device=z.device), torch.zeros(1, ld, device=z.device),
                      torch.zeros(E, ld, device=z.device))
        ops.head_serial_fwd(Z, Wh, Ech, ub, so, wo, eo, E, Ww, We, B)
        ctx.save_for_backward(Z, Wh, Ech)
        ctx.meta = (B, ub, E, Ww, We)
        return ref_view(so, B), ref_view(wo, B).unsqueeze(2), ref_view(eo, B).unsqueeze(2)

    @staticmethod
    def backward(ctx, g_so, g_wo, g_eo):
        Z, Wh, Ech = ctx.saved_tensors
        B, ub, E, Ww, We = ctx.meta
        ld = Z.stride(0)
        Gs, Gw, Ge = to_soa(g_so, ld), to_soa(g_wo[:, :, 0], ld), to_soa(g_eo[:, :, 0], ld)
        dZ = torch.zeros(E + 2, ld, device=Z.device)
        g_wh, g_ech = torch.zeros(1, Ww, ld, device=Z.device), torch.zeros(E, We, ld, device=Z.device)
        ops.head_serial_bwd(Z, Wh, Ech, ub, Gs, Gw, Ge, dZ, g_wh, g_ech, E, Ww, We, B)
        return ref_view(dZ, B), ref_view(g_wh, B), ref_view(g_ech, B), None


_SCALAR_CACHE = {}


def _scalar(x):
    """Host value of a (device) scalar such as `warehouse_upper_bound`, read back ONCE per tensor version: the heads take
    it as a launch argument every period, and a device-to-host copy per period would be a sync (and cannot be captured into a
    HIP graph)."""
    if not torch.is_tensor(x):
        return float(x)
    key = (x.data_ptr(), x._version, x.device)
    hit = _SCALAR_CACHE.get(id(x))
    if hit is None or hit[0] != key or hit[2]() is not x:
        import weakref
        if len(_SCALAR_CACHE) > 64:
            _SCALAR_CACHE.clear()
        hit = _SCALAR_CACHE[id(x)] = (key, float(x.reshape(-1)[0]), weakref.ref(x))
    return hit[1]


# ---- architectures of the hot path ----------------------------------------------------------------------------------

class VanillaOneStore(MyNeuralNetwork):
    """neural_networks.py:195-214: MLP over the store pipeline, softplus(x + 1)."""

    def forward(self, observation):
        x = observation["store_inventories"].flatten(start_dim=1)
        return {"stores": _SoftplusHead.apply(self.net["master"](x)).unsqueeze(2)}


class _ClosedFormPolicy(MyNeuralNetwork):
    """Policies whose `net` maps the constant 0 to a handful of scalar levels (neural_networks.py:228): the trainer runs them
    through the whole-horizon kernel of closed_form.py, which takes `closed_form_levels()`; `forward` is the per-period form
    for `Simulator.step` callers."""

    def _net_of_zero(self):
        dev = next(self.parameters()).device
        return self.net["master"](torch.zeros(1, device=dev))

    def closed_form_levels(self):
        return self._net_of_zero()


class BaseStock(_ClosedFormPolicy):
    """neural_networks.py:216-229."""

    def forward(self, observation):
        x = observation["store_inventories"]
        inv_pos = x.sum(dim=2)
        level = self.closed_form_levels()
        return {"stores": torch.clip(level - inv_pos, min=0).unsqueeze(2)}


class CappedBaseStock(_ClosedFormPolicy):
    """neural_networks.py:296-311."""

    def forward(self, observation):
        x = observation["store_inventories"]
        inv_pos = x.sum(dim=2)
        out = self.closed_form_levels()
        return {"stores": torch.clip(out[0] - inv_pos, min=torch.zeros(1, device=x.device), max=out[1]).unsqueeze(2)}


class EchelonStock(_ClosedFormPolicy):
    """neural_networks.py:231-294 (locations ordered upstream -> downstream; cumulative softplus base levels)."""

    def closed_form_levels(self):
        x = self.activation_functions["softplus"](self._net_of_zero() + 10.0)
        return torch.cumsum(x, dim=0).flip(dims=[0])

    def forward(self, observation):
        s_inv, w_inv, e_inv = (observation[k] for k in ("store_inventories", "warehouse_inventories",
                                                        "echelon_inventories"))
        E = e_inv.size(1)
        levels = self.closed_form_levels()
        pos = torch.concat((e_inv.sum(dim=2), w_inv.sum(dim=2), s_inv.sum(dim=2)), dim=1)
        upstream = torch.concat((1000000 * torch.ones_like(w_inv[:, :, 0]), e_inv[:, :, 0], w_inv[:, :, 0]), dim=1)
        want = torch.clip(torch.stack([levels[k] - pos[:, k:].sum(dim=1) for k in range(2 + E)], dim=1), min=0)
        alloc = torch.minimum(want, upstream)
        return {"stores": alloc[:, -1:].unsqueeze(2), "warehouses": alloc[:, -2:-1].unsqueeze(2),
                "echelons": alloc[:, :E].unsqueeze(2)}


class VanillaSerial(MyNeuralNetwork):
    """neural_networks.py:314-355.  The reference wraps the MLP input in torch.tensor(...) (:329), i.e. DETACHES it;
    replicated so that gradients match."""

    def forward(self, observation):
        s_inv, w_inv, e_inv = (observation[k] for k in ("store_inventories", "warehouse_inventories",
                                                        "echelon_inventories"))
        x = self.flatten_then_concatenate_tensors([s_inv, w_inv, e_inv]).detach()
        z = self.net["master"](x)
        so, wo, eo = _SerialHead.apply(z, w_inv, e_inv, _scalar(self.warehouse_upper_bound))
        return {"stores": so, "warehouses": wo, "echelons": eo}


class VanillaWarehouse(MyNeuralNetwork):
    """neural_networks.py:358-427: one MLP over all pipelines; per warehouse a softmax over its connected stores (+ a
    constant-1 'keep' logit) times its on-hand stock; sigmoid x upper bound for the warehouses' own orders."""

    def __init__(self, args, scenario=None, device="cpu"):
        super().__init__(args, device)
        self.scenario = scenario
        self.transshipment = args.get("transshipment", False)
        self._adj_cache = {}

    def adjacency(self, n_stores, n_warehouses, device):
        key = (n_stores, n_warehouses, str(device))
        if key not in self._adj_cache:
            if n_warehouses == 1:
                adj = torch.ones(1, n_stores, dtype=torch.int32)
            else:
                raw = self.scenario.problem_params.get("warehouse_store_adjacency", None) if self.scenario else None
                if raw is None:
                    raise ValueError(f"warehouse_store_adjacency matrix required for n_warehouses={n_warehouses}")
                adj = (torch.tensor(raw, dtype=torch.float32) != 0).to(torch.int32)
            self._adj_cache[key] = adj.contiguous().to(device)
        return self._adj_cache[key]

    def forward(self, observation):
        s_inv, w_inv = observation["store_inventories"], observation["warehouse_inventories"]
        S, Wn = s_inv.size(1), w_inv.size(1)
        x = torch.cat((s_inv.flatten(start_dim=1), w_inv.flatten(start_dim=1)), dim=1)
        z = self.net["master"](x)
        if torch.compiler.is_compiling():   # the registered operator (library.py) while a compiler traces; same kernel
            from . import library  # noqa: F401  (registers the operators)
            ub = self.warehouse_upper_bound
            ub = ub if torch.is_tensor(ub) else torch.tensor([float(ub)])
            so, wo = torch.ops.nic.softmax_alloc(z, w_inv, self.adjacency(S, Wn, z.device), ub, bool(self.transshipment), S, Wn)
        else:
            so, wo = _WarehouseHead.apply(z, w_inv, self.adjacency(S, Wn, z.device), _scalar(self.warehouse_upper_bound),
                                          bool(self.transshipment), S, Wn)
        return {"stores": so, "warehouses": wo}



class SymmetryAware(MyNeuralNetwork):
    """The "symmetry-aware policy net" BASELINE.json's cfg3 names.  It is NOT in the reference's current source (no class, no
    YAML; `get_architecture` registers 13 names without it); SURVEY §2.2 recovered its forward from the stale bytecode, and the
    pieces it was built from are still upstream: `apply_proportional_allocation` (neural_networks.py:111-138),
    `concatenate_signal_to_object_state_tensor` (:178-187), the `store` / `warehouse` / `context` defaults of
    `set_default_output_size` (:1505-1509).  Weight sharing across stores: ONE context net over the whole state, ONE warehouse
    net applied to (warehouse pipeline | context), ONE store net applied to every (store pipeline | mean, std, underage cost,
    lead time | context); store orders are scaled down proportionally when they exceed the warehouse's on-hand stock, the
    warehouse orders its net's output times the upper bound.  Actions are 3-D as today's simulator needs them (the bytecode
    returned 2-D ones).  One-warehouse settings with the demand moments in the observation.
    PARITY UNPINNED: there is no reference to compare with; the HIP path is checked against this repository's own
    CPU restatement (oracle.symmetry_aware_act)."""

    def forward(self, observation):
        s_inv, w_inv = observation["store_inventories"], observation["warehouse_inventories"]
        if w_inv.size(1) != 1:
            raise ValueError("symmetry_aware: one-warehouse settings only")
        context = self.net["context"](torch.cat((s_inv.flatten(start_dim=1), w_inv.flatten(start_dim=1)), dim=1))
        warehouse_out = self.net["warehouse"](self.concatenate_signal_to_object_state_tensor(w_inv, context))[:, :, 0]
        store_params = torch.stack([observation["mean"], observation["std"], observation["underage_costs"],
                                    observation["lead_times"][:, :, 0]], dim=2)
        store_in = self.concatenate_signal_to_object_state_tensor(torch.cat((s_inv, store_params), dim=2), context)
        store_out = self.net["store"](store_in)[:, :, 0]
        stores = self.apply_proportional_allocation(store_out, w_inv[:, :, 0])
        ub = self.warehouse_upper_bound
        ub = ub.to(warehouse_out.device) if torch.is_tensor(ub) else torch.tensor([float(ub)], device=warehouse_out.device)
        return {"stores": stores.unsqueeze(2), "warehouses": (warehouse_out * ub.reshape(1, -1)).unsqueeze(2)}


class DataDrivenNet(MyNeuralNetwork):
    """neural_networks.py:430-515: one MLP over every observed feature of the real-data settings (pipelines, the past-demand
    window, costs, days from christmas, lead-time matrix); with warehouses the outputs are [Wn warehouse orders | S x Wn store
    orders], masked by the adjacency and scaled down proportionally when a warehouse's pipeline total is short."""

    def __init__(self, args, scenario=None, device="cpu"):
        super().__init__(args, device)
        self.scenario = scenario

    def forward(self, observation):
        n_warehouses = observation["warehouse_inventories"].size(1) if "warehouse_inventories" in observation else 0
        n_stores = observation["store_inventories"].size(1)
        feats = [observation["store_inventories"]]
        if n_warehouses > 0:
            feats.append(observation["warehouse_inventories"])
        feats += [observation["past_demands"], observation["underage_costs"], observation["holding_costs"],
                  observation["days_from_christmas"], observation["lead_times"]]
        outputs = self.net["master"](self.flatten_then_concatenate_tensors(feats))
        if n_warehouses == 0:
            return {"stores": outputs.unsqueeze(2)}
        adjacency = self.scenario.problem_params["warehouse_store_adjacency"]
        # The adjacency is a static list of the setting: its device copy is built once per device and the "does warehouse w
        # serve anyone" test (:499 upstream, a device tensor compared on the host) is answered from the list - the upstream form
        # costs one host->device copy and Wn device->host syncs per PERIOD (1.3 ms per period on the real-data batch) and cannot
        # be captured into a HIP graph.
        # (keyed on the list's CONTENT: `id()` of a list that was replaced can be recycled)
        mkey = (tuple(tuple(row) for row in adjacency), str(outputs.device))
        if getattr(self, "_edge_mask_key", None) != mkey:
            self._edge_mask = torch.tensor(adjacency, dtype=torch.float32, device=outputs.device).transpose(0, 1)  # [S, Wn]
            self._serves = [sum(adjacency[w]) > 0 for w in range(len(adjacency))]
            self._edge_mask_key = mkey
        edge_mask = self._edge_mask
        warehouse_outputs, store_flat = outputs[:, :n_warehouses], outputs[:, n_warehouses:]
        store_allocation = store_flat.reshape(outputs.size(0), n_stores, n_warehouses) * edge_mask.unsqueeze(0)
        final = torch.zeros_like(store_allocation)
        for w in range(n_warehouses):
            if self._serves[w]:
                final[:, :, w] = self.apply_proportional_allocation(store_allocation[:, :, w],
                                                                    observation["warehouse_inventories"][:, w])
        return {"stores": final, "warehouses": warehouse_outputs.unsqueeze(2)}


class QuantilePolicy(MyNeuralNetwork):
    """neural_networks.py:517-588: map features to a desired quantile per store, invert it with the frozen quantile forecaster
    to a base-stock level, order up to it."""

    def __init__(self, args, device="cpu"):
        super().__init__(args=args, device=device)
        self.fixed_nets = {"quantile_forecaster": self.load_forecaster(args, requires_grad=False)}
        self.allow_back_orders = False

    def load_forecaster(self, nn_params, requires_grad=True):
        from .quantile_forecaster import FullyConnectedForecaster
        import numpy as np
        fc = FullyConnectedForecaster([128, 128], lead_times=nn_params["forecaster_lead_times"], qs=np.arange(0.05, 1, 0.05),
                                      device=self.device)
        # (the shipped file was saved from CUDA tensors; map it to wherever this policy lives)
        fc.load_state_dict(torch.load(f"{nn_params['forecaster_location']}", map_location="cpu"))
        for p in fc.parameters():
            p.requires_grad_(requires_grad)
        return fc.to(self.device)

    def forecast_base_stock_allocation(self, past_demands, days_from_christmas, store_inventories, lead_times, quantiles,
                                       allow_back_orders=False):
        x = torch.cat([past_demands, days_from_christmas.unsqueeze(1).expand(past_demands.shape[0], past_demands.shape[1], 1)],
                      dim=2)
        levels = self.fixed_nets["quantile_forecaster"].get_quantile(x, quantiles, lead_times)
        pos = store_inventories.sum(dim=2)
        alloc = levels - pos if allow_back_orders else torch.clip(levels - pos, min=0)
        return {"stores": alloc.unsqueeze(2)}

    def compute_desired_quantiles(self, args):
        raise NotImplementedError

    def forward(self, observation):
        lead_times = observation["lead_times"][:, :, 0]
        u, h, past, xmas, inv = [observation[k] for k in ("underage_costs", "holding_costs", "past_demands",
                                                          "days_from_christmas", "store_inventories")]
        q = self.compute_desired_quantiles({"underage_costs": u, "holding_costs": h})
        return self.forecast_base_stock_allocation(past, xmas, inv, lead_times, q, allow_back_orders=self.allow_back_orders)


class TransformedNV(QuantilePolicy):
    """neural_networks.py:590-597: a learned map of the newsvendor quantile u / (u + h)."""

    def compute_desired_quantiles(self, args):
        return self.net["master"](args["underage_costs"] / (args["underage_costs"] + args["holding_costs"]))


class QuantileNV(QuantilePolicy):
    """neural_networks.py:599-611: the newsvendor quantile itself (nothing to train)."""

    def __init__(self, args, device="cpu"):
        super().__init__(args=args, device=device)
        self.trainable = False

    def compute_desired_quantiles(self, args):
        return args["underage_costs"] / (args["underage_costs"] + args["holding_costs"])


class ReturnsNV(QuantileNV):
    """neural_networks.py:613-622: QuantileNV that may order negative quantities (non-admissible)."""

    def __init__(self, args, device="cpu"):
        super().__init__(args=args, device=device)
        self.trainable = False
        self.allow_back_orders = True


class FixedQuantile(QuantilePolicy):
    """neural_networks.py:624-631: one learned quantile for every store and period."""

    def compute_desired_quantiles(self, args):
        u = args["underage_costs"]
        return self.net["master"](torch.zeros(1, device=u.device)).unsqueeze(1).expand(u.shape[0], u.shape[1])


class JustInTime(MyNeuralNetwork):
    """neural_networks.py:634-739: the non-admissible oracle benchmark - reads FUTURE demand from the simulator's internal data
    and orders so that units arrive exactly when they are demanded (each store served by its fastest connected warehouse)."""

    def __init__(self, args, scenario=None, device="cpu"):
        super().__init__(args=args, device=device)
        self.scenario = scenario
        self.trainable = False

    def forward(self, observation):
        current_period = observation["current_period"]
        demands, period_shift = self.unpack_args(observation["internal_data"], ["demands", "period_shift"])
        dev = demands.device
        n, n_stores, n_periods = demands.shape
        cur = int(current_period.reshape(-1)[0]) + period_shift  # (host tensor: no device sync)
        rows = torch.arange(n, device=dev)
        n_warehouses = observation["warehouse_inventories"].size(1) if "warehouse_inventories" in observation else 0
        if n_warehouses == 0:
            lt = observation["lead_times"][:, :, 0]
            fut = torch.stack([demands[:, j][rows, torch.clip(cur + lt[:, j].long(), max=n_periods - 1)]
                               for j in range(n_stores)], dim=1)
            return {"stores": torch.clip(fut, min=0).unsqueeze(2)}
        lt, wlt = observation["lead_times"], observation["warehouse_lead_times"]
        adj = torch.tensor(self.scenario.problem_params["warehouse_store_adjacency"], dtype=torch.float32)
        # which connected warehouse has the shortest batch-mean lead time for each store: recomputed for every BATCH like
        # upstream (:695-699; per-sample lead times from file can move the argmin from batch to batch), but only once per
        # batch - the lead-time tensor of an observation is the same object for all periods of a rollout
        # (the cache HOLDS the tensor it was computed from and compares identity + version: an address alone can be handed to
        # the next batch's tensor by the caching allocator; and every rollout starts afresh at its first period)
        held = getattr(self, "_fastest_for", None)
        if held is None or held[0] is not lt or held[1] != lt._version or cur == period_shift:
            mean_lt = lt.mean(dim=0).cpu()
            self._fastest = [
                (int(conn[torch.argmin(mean_lt[st, conn])]) if len(conn) > 0 else None)
                for st, conn in ((st, adj[:, st].nonzero(as_tuple=True)[0]) for st in range(n_stores))]
            self._fastest_for = (lt, lt._version)
        fastest = self._fastest
        alloc = torch.zeros(n, n_stores, n_warehouses, device=dev)
        wh = torch.zeros(n, n_warehouses, device=dev)
        for st, w in enumerate(fastest):
            if w is None:
                continue
            alloc[:, st, w] = demands[:, st][rows, torch.clip(cur + lt[:, st, w].long(), max=n_periods - 1)]
            wh[:, w] += demands[:, st][rows, torch.clip(cur + wlt[:, w].long() + lt[:, st, w].long(), max=n_periods - 1)]
        return {"stores": torch.clip(alloc, min=0), "warehouses": torch.clip(wh, min=0).unsqueeze(2)}


class GNN(MyNeuralNetwork):
    """Message-passing policy over the supply graph (neural_networks.py:742-1492 of the reference): nodes =
    [echelons..., warehouses..., stores...], edges = [internal (source-major), supplier, demand, self-loops of supplying
    nodes]; five small MLPs (`gnn.yml`): initial_node, initial_edge, node_update, edge_update, output.

    Re-designed for the device rather than translated: the graph is static, so it is compiled ONCE into index tensors and
    two dense normalised incidence matrices; message aggregation, the proportional allocation and the scatter of edge
    outputs into the action tensors are then batched gathers / small matmuls over (batch, edge) instead of the reference's
    Python loops over edges and nodes, and every MLP runs on the matrix cores (`HipLinear`, all nodes / edges of all
    scenarios as GEMM columns).  Sums over a node's edges therefore associate differently from the reference's sequential
    `+=` (costs agree to ~1e-6 relative; the oracle keeps the reference's order bit for bit)."""

    def __init__(self, args, scenario, device="cpu"):
        super().__init__(args, device)
        self.scenario = scenario
        self.transshipment = args.get("transshipment", False)
        self._graph = None

    # ---- static structure ----------------------------------------------------------------------------------------
    def _compile_graph(self, observation):
        prob = self.scenario.problem_params
        S, Wn, E = prob.get("n_stores", 0), prob.get("n_warehouses", 0), prob.get("n_extra_echelons", 0)
        dev = observation["store_inventories"].device
        n_nodes = E + Wn + S
        internal, lead = [], []
        if E > 0:  # serial system: echelons -> warehouse -> store
            for i in range(E - 1):
                internal.append((i, i + 1))
                lead.append(float(observation["echelon_lead_times"][0, i + 1]))
            internal.append((E - 1, E))
            lead.append(float(observation["warehouse_lead_times"][0, 0]))
            internal.append((E, E + Wn))
            lead.append(float(observation["lead_times"][0, 0, 0]))
            suppliers, customers = [0], [E + Wn]
            sup_lead = [float(observation["echelon_lead_times"][0, 0])]
        else:
            if Wn == 1:
                conn = [[1] * S]
            else:
                conn = prob.get("warehouse_store_adjacency")
                if conn is None:
                    raise ValueError(f"Multiple warehouses ({Wn}) detected but no 'warehouse_store_adjacency' matrix found "
                                     "in problem_params. Please specify which stores connect to which warehouses.")
            lt0 = observation["lead_times"][0].tolist()  # sample 0's lead times stand for the batch, as upstream (:984)
            for w in range(Wn):
                for st in range(S):
                    if conn[w][st]:
                        internal.append((w, Wn + st))
                        lead.append(float(lt0[st][w]))
            suppliers, customers = list(range(Wn)), list(range(Wn, Wn + S))
            sup_lead = [float(observation["warehouse_lead_times"][0, w]) for w in range(Wn)]
        out_deg, in_deg = [0] * n_nodes, [0] * n_nodes
        for a, b in internal:
            out_deg[a] += 1
            in_deg[b] += 1
        supplying = [] if self.transshipment else [n for n in range(n_nodes) if out_deg[n] > 0]
        n_int, n_sup, n_dem, n_self = len(internal), len(suppliers), len(customers), len(supplying)
        n_edges = n_int + n_sup + n_dem + n_self
        # edge endpoints: index n_nodes = the virtual (all-zero) supplier / customer node
        src = [a for a, _ in internal] + [n_nodes] * n_sup + customers + supplying
        tgt = [b for _, b in internal] + suppliers + [n_nodes] * n_dem + supplying
        for n in suppliers:
            in_deg[n] += 1
        for n in customers:
            out_deg[n] += 1
        for n in supplying:
            in_deg[n] += 1
            out_deg[n] += 1
        a_in = torch.zeros(n_nodes, n_edges)
        a_out = torch.zeros(n_nodes, n_edges)
        for e in range(n_edges):
            if tgt[e] < n_nodes and not (n_int + n_sup <= e < n_int + n_sup + n_dem):
                a_in[tgt[e], e] = 1.0 / max(in_deg[tgt[e]], 1) ** 0.5
            if src[e] < n_nodes and not (n_int <= e < n_int + n_sup):
                a_out[src[e], e] = 1.0 / max(out_deg[src[e]], 1) ** 0.5
        # proportional allocation groups: a supplying node's outgoing internal edges + its self-loop
        group_of_edge = [-1] * n_edges
        group_nodes = []
        for n in range(n_nodes):
            members = [i for i, (a, _) in enumerate(internal) if a == n]
            if n in supplying:
                members.append(n_int + n_sup + n_dem + supplying.index(n))
            if members:
                for e in members:
                    group_of_edge[e] = len(group_nodes)
                group_nodes.append(n)
        member = torch.zeros(len(group_nodes), n_edges)
        for e, gi in enumerate(group_of_edge):
            if gi >= 0:
                member[gi, e] = 1.0
        # action layout: edge index per (row, column), n_edges = "no edge" (reads a zero column)
        if E > 0:
            mapping = {"stores": [[n_int - 1]], "warehouses": [[n_int - 2]],
                       "echelons": [[n_int]] + [[i - 1] for i in range(1, E)]}
        else:
            rows = [[] for _ in range(S)]
            for i, (_, b) in enumerate(internal):
                rows[b - Wn].append(i)  # the j-th CONNECTED edge of a store is its column j, as upstream (:1423-1428)
            mapping = {"stores": rows, "warehouses": [[n_int + w] for w in range(Wn)]}
        gather = {}
        for kind, rows in mapping.items():
            if rows:
                width = max(len(r) for r in rows)
                gather[kind] = torch.tensor([r + [n_edges] * (width - len(r)) for r in rows], dtype=torch.long, device=dev)
        self._graph = dict(
            n_nodes=n_nodes, n_edges=n_edges, E=E, Wn=Wn, S=S,
            src=torch.tensor(src, dtype=torch.long, device=dev), tgt=torch.tensor(tgt, dtype=torch.long, device=dev),
            lead=torch.tensor(lead + sup_lead + [0.0] * (n_dem + n_self), device=dev).view(1, -1, 1),
            a_in=a_in.to(dev), a_out=a_out.to(dev), member=member.to(dev),
            group_nodes=torch.tensor(group_nodes, dtype=torch.long, device=dev), gather=gather,
            steps=(E + 1) if E > 0 else 1, device=dev)
        return self._graph

    # ---- per-call features ---------------------------------------------------------------------------------------
    @staticmethod
    def _node_features(observation, E):
        feats, inv_lens = [], []
        if E > 0:
            feats.append(torch.cat([observation["echelon_inventories"],
                                    observation["echelon_holding_costs"].unsqueeze(-1)], dim=-1))
            inv_lens.append(observation["echelon_inventories"].size(-1))
        wl = [observation["warehouse_inventories"], observation["warehouse_holding_costs"].unsqueeze(-1)]
        if observation.get("warehouse_edge_costs") is not None:
            wl.append(observation["warehouse_edge_costs"].unsqueeze(-1))
        feats.append(torch.cat(wl, dim=-1))
        inv_lens.append(observation["warehouse_inventories"].size(-1))
        sl = [observation["store_inventories"], observation["holding_costs"].unsqueeze(-1),
              observation["underage_costs"].unsqueeze(-1)]
        if "past_demands" in observation:
            sl.append(observation["past_demands"])
            if "days_from_christmas" in observation:
                sl.append(observation["days_from_christmas"].unsqueeze(-1))
        else:
            sl += [observation["mean"].unsqueeze(-1), observation["std"].unsqueeze(-1)]  # KeyError like upstream (:888)
        feats.append(torch.cat(sl, dim=-1))
        inv_lens.append(observation["store_inventories"].size(-1))
        max_inv = max(inv_lens)
        max_st = max(f.size(-1) - n for f, n in zip(feats, inv_lens))
        padded = [torch.cat([F.pad(f[:, :, :n], (0, max_inv - n)), F.pad(f[:, :, n:], (0, max_st - (f.size(2) - n)))], dim=2)
                  for f, n in zip(feats, inv_lens)]
        return torch.cat(padded, dim=1)

    def forward(self, observation):
        g = self._graph
        if g is None or g["device"] != observation["store_inventories"].device:
            g = self._compile_graph(observation)
        B = observation["store_inventories"].size(0)
        nodes = self.net["initial_node"](self._node_features(observation, g["E"]))

        def endpoints(nd):  # (source, target) features of every edge; index n_nodes selects the zero row
            ext = torch.cat([nd, nd.new_zeros(B, 1, nd.size(-1))], dim=1)
            return ext[:, g["src"]], ext[:, g["tgt"]]

        s_f, t_f = endpoints(nodes)
        edges = self.net["initial_edge"](torch.cat([s_f, t_f, g["lead"].expand(B, -1, -1)], dim=-1))
        for _ in range(g["steps"]):
            incoming = torch.matmul(g["a_in"], edges)    # [n_nodes, n_edges] x [B, n_edges, D]
            outgoing = torch.matmul(g["a_out"], edges)
            nodes = nodes + self.net["node_update"](torch.cat([nodes, incoming, outgoing], dim=-1))
            s_f, t_f = endpoints(nodes)
            edges = edges + self.net["edge_update"](torch.cat([edges, s_f, t_f], dim=-1))
        out = self.net["output"](edges).squeeze(-1)  # desired quantity per edge
        # proportional allocation: scale a node's outgoing edges (and its self-loop) to its on-hand stock
        parts = []
        if g["E"] > 0:
            parts.append(observation["echelon_inventories"][:, :, 0])
        if g["Wn"] > 0:
            parts.append(observation["warehouse_inventories"][:, :, 0])
        parts.append(observation["store_inventories"][:, :, 0])
        on_hand = torch.cat(parts, dim=1)[:, g["group_nodes"]]                 # [B, groups]
        scale = on_hand / (torch.matmul(out, g["member"].t()) + 1e-10)
        if not self.transshipment:
            scale = torch.clip(scale, max=1.0)
        in_group = g["member"].sum(dim=0)                                      # 1 for edges that belong to a group
        alloc = out * (torch.matmul(scale, g["member"]) + (1.0 - in_group))
        ext = torch.cat([alloc, alloc.new_zeros(B, 1)], dim=1)
        return {kind: ext[:, idx] for kind, idx in g["gather"].items()}


class NeuralNetworkCreator:
    """neural_networks.py:1495-1574."""

    def set_default_output_size(self, module_name, problem_params):
        S, Wn = problem_params["n_stores"], problem_params["n_warehouses"]
        master = S * Wn + Wn if Wn > 1 else S + Wn
        return {"master": master, "store": 1, "warehouse": 1, "context": None}[module_name]

    def get_architecture(self, name):
        architectures = {
            "vanilla_one_store": VanillaOneStore, "base_stock": BaseStock, "capped_base_stock": CappedBaseStock,
            "echelon_stock": EchelonStock, "vanilla_serial": VanillaSerial, "vanilla_warehouse": VanillaWarehouse,
            "data_driven": DataDrivenNet, "transformed_nv": TransformedNV, "fixed_quantile": FixedQuantile,
            "quantile_nv": QuantileNV, "returns_nv": ReturnsNV, "just_in_time": JustInTime, "gnn": GNN,
            # not in the reference's current registry (:1521-1535): the policy BASELINE cfg3 names, recovered per SURVEY §2.2
            "symmetry_aware": SymmetryAware,
        }
        return architectures[name]  # KeyError for unknown names, like the reference (:1536)

    def get_warehouse_upper_bound(self, warehouse_upper_bound_mult, scenario, device="cpu"):
        mean = scenario.store_params["demand"]["mean"]
        if type(mean) == float:
            mean = [mean]
        return torch.tensor([warehouse_upper_bound_mult * sum(mean)]).float().to(device)

    def create_neural_network(self, scenario, nn_params, device="cpu"):
        p = copy.deepcopy(nn_params)
        for key, val in p["output_sizes"].items():
            if val is None:
                p["output_sizes"][key] = self.set_default_output_size(key, scenario.problem_params)
        cls = self.get_architecture(p["name"])
        if p["name"] == "symmetry_aware":
            import warnings
            warnings.warn("symmetry_aware is EXPERIMENTAL: the reference ships no source or config for it (its registry raises "
                          "KeyError for this name); this implementation follows a recovered description and is checked only "
                          "against this repository's own CPU restatement - results are not reference parity", stacklevel=2)
        if p["name"] in ("vanilla_warehouse", "gnn", "just_in_time", "data_driven"):
            model = cls(p, scenario, device=device)
        else:
            model = cls(p, device=device)
        if "warehouse_upper_bound_mult" in nn_params.keys():
            model.warehouse_upper_bound = self.get_warehouse_upper_bound(nn_params["warehouse_upper_bound_mult"], scenario,
                                                                         device)
        return model.to(device)
