"""Thin torch-facing wrappers over the C ABI (include/nic_rollout.h).

Every function takes/returns CUDA tensors in the scenario-minor layout of `layout.py`, enqueues exactly one HIP
kernel on torch's current stream and never synchronises.  There is no CPU path: CPU tensors raise.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr
from .layout import EnvProblem, Table


def _dev(t):
    if t is not None and not t.is_cuda:
        raise _lib.NicUnavailableError("nic ops need device tensors (no CPU fallback; see oracle/ for the CPU checker)")
    return t


def _ld(m):
    """leading dimension of a row-major matrix (torch reports an arbitrary stride for a dimension of size 1)"""
    return m.stride(0) if m.shape[0] > 1 else max(m.stride(0), m.shape[1])


class EnvState:
    """The three pipelines of one period in SoA layout: store [S][Ws][ldb], wh [Wn][Ww][ldb], ech [E][We][ldb]."""
    __slots__ = ("store", "wh", "ech")

    def __init__(self, store, wh=None, ech=None):
        self.store, self.wh, self.ech = store, wh, ech

    @staticmethod
    def zeros_like(prob: EnvProblem):
        dev, ld = prob.device, prob.ldb
        return EnvState(
            torch.zeros(prob.S, prob.Ws, ld, device=dev),
            torch.zeros(prob.Wn, prob.Ww, ld, device=dev) if prob.Wn else None,
            torch.zeros(prob.E, prob.We, ld, device=dev) if prob.E else None)


def env_step_fwd(prob: EnvProblem, state: EnvState, demand: Table, store_orders: Table, wh_orders, ech_orders,
                 out: EnvState = None, reward=None, zero_lead_upstream=False):
    """nic_env_step_fwd.  Returns (next_state, reward[ldb]).  zero_lead_upstream: the reference's treatment of non-zero orders with
    lead time 0 inside the launch (include/nic_rollout.h; default: dropped)."""
    _dev(state.store)
    out = out or EnvState.zeros_like(prob)
    if reward is None:
        reward = torch.zeros(prob.ldb, device=prob.device)
    io = prob.make_io(state.store, state.wh, state.ech, demand, store_orders, wh_orders, ech_orders)
    check(lib().nic_env_step_fwd(io, ptr(out.store), ptr(out.wh), ptr(out.ech), ptr(reward), int(bool(zero_lead_upstream)),
                                 current_stream()))
    return out, reward


def env_step_bwd(prob: EnvProblem, state: EnvState, demand: Table, store_orders: Table, wh_orders, ech_orders,
                 g_out: EnvState, g_reward: Table, g_in: EnvState = None, g_orders=None, zero_lead_upstream=False):
    """nic_env_step_bwd.  g_out members may be None (zeros).  Returns (g_state_in, (g_store_orders [S][nsup][ldb],
    g_wh_orders [Wn][ldb] | None, g_ech_orders [E][ldb] | None))."""
    _dev(state.store)
    dev, ld = prob.device, prob.ldb
    g_in = g_in or EnvState.zeros_like(prob)
    if g_orders is None:
        g_orders = (torch.zeros(prob.S, prob.nsup, ld, device=dev),
                    torch.zeros(prob.Wn, ld, device=dev) if prob.Wn else None,
                    torch.zeros(prob.E, ld, device=dev) if prob.E else None)
    io = prob.make_io(state.store, state.wh, state.ech, demand, store_orders, wh_orders, ech_orders)
    check(lib().nic_env_step_bwd(io, ptr(g_out.store), ptr(g_out.wh), ptr(g_out.ech), g_reward.t2(), ptr(g_in.store),
                                 ptr(g_in.wh), ptr(g_in.ech), ptr(g_orders[0]), ptr(g_orders[1]), ptr(g_orders[2]),
                                 int(bool(zero_lead_upstream)), current_stream()))
    return g_in, g_orders


# ---- policy MLP layers (feature-major: X is [K][ldb]) ----------------------------------------------------------

def linear_fwd(W, bias, X, Y, n_scenarios, act):
    """Y[N][ldb] = act(W[N][K(ldw)] @ X[K][ldb] + bias)."""
    _dev(X)
    N, K = W.shape[0], X.shape[0]
    check(lib().nic_linear_fwd(ptr(W), _ld(W), ptr(bias), ptr(X), ptr(Y), N, K, n_scenarios, X.stride(0), act,
                               current_stream()))
    return Y


def linear_fwd_thin_in_ok(N, K):
    return bool(lib().nic_linear_fwd_thin_in_ok(int(N), int(K)))


def linear_fwd_thin_in(Wt, bias, X, Y, n_scenarios, act):
    """Y[N][ldb] = act(Wt[K][N(ldwt)]^T @ X[K][ldb] + bias) for a short contraction (K <= 52) and many output rows."""
    _dev(X)
    K, N = X.shape[0], Y.shape[0]
    check(lib().nic_linear_fwd_thin_in(ptr(Wt), _ld(Wt), ptr(bias), ptr(X), ptr(Y), N, K, n_scenarios, X.stride(0), act,
                                       current_stream()))
    return Y


def linear_dgrad(Wt, dY, Hprev, dX, n_scenarios, act_prev, accumulate):
    """dX[K][ldb] (+)= (Wt[K][N(ldwt)] @ dY[N][ldb]) * act'(Hprev)."""
    _dev(dY)
    K, N = dX.shape[0], dY.shape[0]
    check(lib().nic_linear_dgrad(ptr(Wt), _ld(Wt), ptr(dY), ptr(Hprev), ptr(dX), N, K, n_scenarios, dY.stride(0),
                                 act_prev, int(accumulate), current_stream()))
    return dX


def wgrad_num_splits(N, K, n_scenarios):
    return lib().nic_wgrad_num_splits(N, K, n_scenarios)


def wgrad_periods_num_splits(N, K, n_scenarios, n_periods):
    """Slab slots of `linear_wgrad_periods` (scenario splits x period groups) that give every CU a workgroup."""
    return lib().nic_wgrad_periods_num_splits(N, K, n_scenarios, n_periods)


def linear_wgrad(dY, X, slab, n_scenarios):
    """slab[split][N][lds] += per-split sum_b dY[N][b] X[K][b] (column K = bias gradient)."""
    _dev(dY)
    N, K = dY.shape[0], X.shape[0]
    check(lib().nic_linear_wgrad(ptr(dY), ptr(X), ptr(slab), slab.stride(1), N, K, n_scenarios, dY.stride(0),
                                 slab.shape[0], current_stream()))
    return slab


def linear_wgrad_periods(dY, X, slab, n_scenarios):
    """dY [T][N][ldb], X [T'][K'][ldb] views (period = leading dimension; only the strides and the first K rows of X's
    periods matter): slab += sum_t per-split dY[t] X[t]^T in one launch."""
    _dev(dY)
    T, N, K = dY.shape[0], dY.shape[1], X.shape[1]
    assert X.shape[0] == T and dY.stride(1) == X.stride(1) and dY.stride(2) == 1 and X.stride(2) == 1
    check(lib().nic_linear_wgrad_periods(ptr(dY), ptr(X), ptr(slab), slab.stride(1), N, K, n_scenarios, dY.stride(1),
                                         slab.shape[0], T, dY.stride(0), X.stride(0), current_stream()))
    return slab


def linear_bwd_thin_ok(N, K):
    """shapes nic_linear_bwd_thin takes (the logits layer of the policy MLPs)"""
    return N <= _lib.NIC_THIN_MAX_ROWS and K >= 32 and K % 32 == 0


def linear_bwd_thin(W, dY, X, dX, slab, n_scenarios, act_prev):
    """One pass over X: dX = act'(X) * W^T dY  and  slab[split] += dY X^T (column K = bias gradient)."""
    _dev(dY)
    N, K = dY.shape[0], X.shape[0]
    check(lib().nic_linear_bwd_thin(ptr(W), _ld(W), ptr(dY), ptr(X), ptr(dX), ptr(slab), slab.stride(1), N, K,
                                    n_scenarios, dY.stride(0), int(act_prev), slab.shape[0], current_stream()))
    return dX


def wgrad_reduce(slab, dW, db, K, scale=1.0):
    N = dW.shape[0]
    check(lib().nic_wgrad_reduce(ptr(slab), slab.stride(1), slab.shape[0], ptr(dW), _ld(dW), ptr(db), N, K,
                                 float(scale), current_stream()))


# ---- policy heads ----------------------------------------------------------------------------------------------

def round_orders(block, B):
    """in-place torch.round (half to even) of a [rows][ldb] order block: discrete allocation (trainer.py:201-202)"""
    _dev(block)
    check(lib().nic_round_orders(ptr(block), block.shape[0], B, block.stride(0), current_stream()))


def head_warehouse_fwd(Z, wh_inv, adjacency, ub, transshipment, store_orders, wh_orders, S, Wn, Ww, B):
    _dev(Z)
    check(lib().nic_head_warehouse_fwd(ptr(Z), ptr(wh_inv), ptr(adjacency), float(ub), int(transshipment),
                                       ptr(store_orders), ptr(wh_orders), S, Wn, Ww, B, Z.stride(0), current_stream()))


def head_warehouse_bwd(Z, wh_inv, adjacency, ub, transshipment, g_store_orders, g_wh_orders, dZ, g_wh_inv, S, Wn, Ww, B):
    _dev(Z)
    check(lib().nic_head_warehouse_bwd(ptr(Z), ptr(wh_inv), ptr(adjacency), float(ub), int(transshipment),
                                       ptr(g_store_orders), ptr(g_wh_orders), ptr(dZ), ptr(g_wh_inv), S, Wn, Ww, B,
                                       Z.stride(0), current_stream()))


def head_env_fwd(prob: EnvProblem, state: EnvState, demand: Table, store_orders: Table, wh_orders: Table, Z, adjacency, ub,
                 transshipment, out: EnvState, reward, logit_rows=None, first_wh_row=-1):
    """nic_head_env_fwd: vanilla_warehouse head + env step in one launch.  `store_orders` / `wh_orders` are the dense
    [S][Wn][ldb] / [Wn][ldb] blocks the orders are WRITTEN to (and consumed from).  logit_rows / first_wh_row: compact logits."""
    _dev(Z)
    io = prob.make_io(state.store, state.wh, None, demand, store_orders, wh_orders, None)
    check(lib().nic_head_env_fwd(io, ptr(Z), ptr(adjacency), ptr(logit_rows), int(first_wh_row), float(ub), int(transshipment),
                                      ptr(out.store), ptr(out.wh), ptr(reward), current_stream()))
    return out, reward


def head_env_bwd(prob: EnvProblem, state: EnvState, demand: Table, store_orders: Table, wh_orders: Table, Z, adjacency, ub,
                 transshipment, g_out: EnvState, g_reward: Table, g_in: EnvState, g_orders, dZ, logit_rows=None, first_wh_row=-1):
    """nic_head_env_bwd: env-step adjoint + head adjoint in one launch (g_orders: scratch (store, warehouse) blocks).
    logit_rows / first_wh_row: compact logits."""
    _dev(Z)
    io = prob.make_io(state.store, state.wh, None, demand, store_orders, wh_orders, None)
    check(lib().nic_head_env_bwd(io, ptr(Z), ptr(adjacency), ptr(logit_rows), int(first_wh_row), float(ub), int(transshipment),
                                      ptr(g_out.store), ptr(g_out.wh), g_reward.t2(), ptr(g_in.store), ptr(g_in.wh),
                                      ptr(g_orders[0]), ptr(g_orders[1]), ptr(dZ), current_stream()))
    return g_in, dZ


def period_tail_ok(prob: EnvProblem, n_out, K, N1):
    """shapes the fused per-period tail takes (nic_period_tail_ok)"""
    return bool(lib().nic_period_tail_ok(prob.dims(), int(n_out), int(K), int(N1)))


def period_tail_bwd_slots(n_scenarios):
    return lib().nic_period_tail_bwd_slots(int(n_scenarios))


def period_tail_desc(prob: EnvProblem, state_block, demand: Table, orders_block, adjacency, ub, transshipment, W_out, b_out, Wt_in):
    """NicPeriodTail of one period.  state_block: [F (+ ...)][ldb] rows [store | warehouse]; orders_block: [S Wn + Wn][ldb];
    W_out [n_out][ldw] (rows padded to 16 bytes); Wt_in [F + 1][ldwt] = the first layer transposed with its bias as row F."""
    S, Wn, ld = prob.S, prob.Wn, prob.ldb
    F_store = S * prob.Ws
    store = state_block[:F_store]
    wh = state_block[F_store:F_store + Wn * prob.Ww]
    t = _lib.NicPeriodTail()
    t.io = prob.make_io(store, wh, None, demand, Table(orders_block[:S * Wn], Wn * ld, 1, ld), Table(orders_block[S * Wn:], ld, 1), None)
    t.adjacency, t.upper_bound, t.transshipment = ptr(adjacency), float(ub), int(transshipment)
    t.W_out, t.ldw_out, t.b_out = ptr(W_out), _ld(W_out), ptr(b_out)
    t.n_out, t.K = W_out.shape[0], W_out.shape[1]
    t.Wt_in, t.ldwt_in, t.N1 = ptr(Wt_in), _ld(Wt_in), Wt_in.shape[1]
    t._keep = (state_block, demand, orders_block, adjacency, W_out, b_out, Wt_in)
    return t


def period_tail_fwd(desc, H_last, Z, state_out, reward, H_first_next):
    """nic_period_tail_fwd: logits layer + head + env step + next period's first layer (ELU) in one launch."""
    _dev(H_last)
    check(lib().nic_period_tail_fwd(desc, ptr(H_last), ptr(Z), ptr(state_out), ptr(reward), ptr(H_first_next), current_stream()))


def period_tail_bwd(desc, Z, H_last, dZ_first_next, g_state_next, g_reward: Table, g_state_out, dH_last, slab, first):
    """nic_period_tail_bwd: first layer's input gradient of period t+1 + env / head adjoints + logits layer backward."""
    _dev(Z)
    check(lib().nic_period_tail_bwd(desc, ptr(Z), ptr(H_last), ptr(dZ_first_next), ptr(g_state_next), g_reward.t2(), ptr(g_state_out),
                                    ptr(dH_last), ptr(slab), slab.stride(1), slab.shape[0], int(bool(first)), current_stream()))


def wide_rollout_ok(prob: EnvProblem, n_out, H, n_hidden):
    """shapes the whole-horizon kernels of the wide vanilla_warehouse policy take (nic_wide_rollout_ok).  An EXPERIMENT
    (include/nic_experiments.h, tools/experiments/wide_rollout.hip): it ties or loses against the tiled launches, so the default
    library does not carry it - False unless the library was built with NIC_BUILD_EXPERIMENTS=1."""
    if not _lib.has_experiments():
        return False
    return bool(lib().nic_wide_rollout_ok(prob.dims(), int(n_out), int(H), int(n_hidden)))


def wide_pack_hidden(W, out=None):
    """[N][K] weight of an H x H layer -> MFMA A fragments [N/32][K/8][64 lanes][4]: lane (r, h) of (tile, group g) holds
    W[32 tile + r][8 g + 2 j + h], j = 0..3 (NicWideRollout.Wp_hidden)."""
    N, K = W.shape
    src = W.detach().reshape(N // 32, 32, K // 8, 4, 2).permute(0, 2, 4, 1, 3)   # [tile][group][h][r][j]
    if out is None:
        return src.contiguous()
    out.view(N // 32, K // 8, 2, 32, 4).copy_(src)
    return out


def wide_pack_out_index(n_out, H, device):
    """(row, column) gather indices of NicWideRollout.Wq_out: [H/32][16][64]; lane (n, h) of (tile, r) holds
    W_out[n][32 tile + (r & 3) + 8 (r >> 2) + 4 h] (rows >= n_out come from a zero row)."""
    lane = torch.arange(64, device=device)
    r = torch.arange(16, device=device)
    tile = torch.arange(H // 32, device=device)
    col = 32 * tile[:, None, None] + ((r & 3) + 8 * (r >> 2))[None, :, None] + 4 * (lane >> 5)[None, None, :]
    row = (lane & 31)[None, None, :].expand_as(col)
    return row.contiguous(), col.contiguous()


def wide_rollout_desc(prob: EnvProblem, T, adjacency, ub, transshipment, demand_soa, shift, states, orders, logits, rewards, hidden,
                      Wt_in, Wp_hidden, b_hidden, Wq_out, b_out):
    """NicWideRollout over the rollout engine's history buffers (states [T+1][F+1][ld], orders [T][.][ld], logits [T][.][ld],
    rewards [T][ld], hidden[l] [T][H][ld] or None); demand_soa [T'][S][ld], first period `shift`."""
    w = _lib.NicWideRollout()
    w.io = prob.make_io(None, None, None, Table.null(), Table(None, 0, 0, 0), None, None)
    w.adjacency, w.upper_bound, w.transshipment = ptr(adjacency), float(ub), int(transshipment)
    w.T, w.H, w.n_hidden, w.n_out = int(T), Wt_in.shape[1], len(Wp_hidden) + 1, logits.shape[1]
    d = demand_soa[shift:]
    w.demand, w.ps_demand, w.ld_demand = ptr(d), demand_soa.stride(0), demand_soa.stride(1)
    w.states, w.orders, w.logits, w.rewards = ptr(states), ptr(orders), ptr(logits), ptr(rewards)
    w.ps_state, w.ps_orders, w.ps_logits = states.stride(0), orders.stride(0), logits.stride(0)
    for l in range(4):
        h = hidden[l] if hidden is not None and l < len(hidden) else None
        w.hidden[l] = ptr(h)
        if h is not None:
            w.ps_hidden = h.stride(0)
    w.Wt_in, w.ldwt_in = ptr(Wt_in), _ld(Wt_in)
    for l in range(1, w.n_hidden):
        w.Wp_hidden[l], w.b_hidden[l] = ptr(Wp_hidden[l - 1]), ptr(b_hidden[l - 1])
    w.Wq_out, w.b_out = ptr(Wq_out), ptr(b_out)
    w._keep = (adjacency, demand_soa, d, states, orders, logits, rewards, hidden, Wt_in, Wp_hidden, b_hidden, Wq_out, b_out)
    return w


def wide_rollout_fwd(desc):
    check(lib().nic_wide_rollout_fwd(desc, current_stream()))


def wide_pack_in_index(F, H, device):
    """gather indices of nic_wide_rollout_bwd's Wq_in ([2][H/32][16][64]) into the first layer's weight W_in [H][F] padded to 64
    input columns: lane (f, h) of (mt, tile, r) holds W_in[32 tile + (r & 3) + 8 (r >> 2) + 4 h][32 mt + f]."""
    lane = torch.arange(64, device=device)
    r = torch.arange(16, device=device)
    tile = torch.arange(H // 32, device=device)
    mt = torch.arange(2, device=device)
    unit = (32 * tile[:, None, None] + ((r & 3) + 8 * (r >> 2))[None, :, None] + 4 * (lane >> 5)[None, None, :])[None].expand(2, -1, -1, -1)
    f = (32 * mt[:, None, None, None] + (lane & 31)[None, None, None, :]).expand_as(unit)
    return unit.contiguous(), f.contiguous()


def wide_ns(n_out):
    ns = (n_out + 1) // 2
    return 4 if ns <= 4 else (9 if ns <= 9 else 16)


def wide_pack_out_t_index(n_out, H, device):
    """gather indices of Wo_t ([H/32][NS][64]) into W_out padded to 2 NS rows: lane (k, h) of (tile, s) holds W_out[2 s + h][32 tile + k]."""
    NS = wide_ns(n_out)
    lane = torch.arange(64, device=device)
    s_ = torch.arange(NS, device=device)
    tile = torch.arange(H // 32, device=device)
    row = (2 * s_[None, :, None] + (lane >> 5)[None, None, :]).expand(H // 32, -1, -1)
    col = (32 * tile[:, None, None] + (lane & 31)[None, None, :]).expand(-1, NS, -1)
    return row.contiguous(), col.contiguous()


def wide_rollout_bwd(desc, g_reward: Table, dZ_hidden, dZ_out, WpT_hidden, Wq_in, Wo_t):
    """nic_wide_rollout_bwd.  dZ_hidden: list of [T][H][ld] tensors (hidden layer 0 ..), dZ_out [T][n_out][ld]; WpT_hidden: packed
    transposes of hidden layers 1 .. (list, index 0 = layer 1)."""
    dz = (ctypes.c_void_p * 4)()
    wt = (ctypes.c_void_p * 4)()
    for l, t_ in enumerate(dZ_hidden):
        dz[l] = ptr(t_)
    for l, t_ in enumerate(WpT_hidden):
        wt[l + 1] = ptr(t_)
    check(lib().nic_wide_rollout_bwd(desc, g_reward.t2(), dz, dZ_hidden[0].stride(0), ptr(dZ_out), dZ_out.stride(0), wt, ptr(Wq_in),
                                     ptr(Wo_t), current_stream()))


def head_data_driven_fwd(Z, wh, mask, store_orders, wh_orders, S, Wn, Ww, B):
    """nic_head_data_driven_fwd: ReLU, adjacency mask and proportional allocation of the data_driven policy (one launch)."""
    _dev(Z)
    check(lib().nic_head_data_driven_fwd(ptr(Z), ptr(wh), ptr(mask), ptr(store_orders), ptr(wh_orders), S, Wn, Ww, B, Z.stride(0),
                                         current_stream()))


def head_data_driven_bwd(Z, wh, mask, g_store_orders, g_wh_orders, dZ, g_wh, S, Wn, Ww, B):
    _dev(Z)
    check(lib().nic_head_data_driven_bwd(ptr(Z), ptr(wh), ptr(mask), ptr(g_store_orders), ptr(g_wh_orders), ptr(dZ), ptr(g_wh),
                                         S, Wn, Ww, B, Z.stride(0), current_stream()))


def head_softplus_fwd(Z, orders, rows, B):
    _dev(Z)
    check(lib().nic_head_softplus_fwd(ptr(Z), ptr(orders), rows, B, Z.stride(0), current_stream()))


def head_softplus_bwd(Z, g_orders, dZ, rows, B):
    _dev(Z)
    check(lib().nic_head_softplus_bwd(ptr(Z), ptr(g_orders), ptr(dZ), rows, B, Z.stride(0), current_stream()))


def head_serial_fwd(Z, wh_inv, ech_inv, ub, store_orders, wh_orders, ech_orders, E, Ww, We, B):
    _dev(Z)
    check(lib().nic_head_serial_fwd(ptr(Z), ptr(wh_inv), ptr(ech_inv), float(ub), ptr(store_orders), ptr(wh_orders),
                                    ptr(ech_orders), E, Ww, We, B, Z.stride(0), current_stream()))


def head_serial_bwd(Z, wh_inv, ech_inv, ub, g_store_orders, g_wh_orders, g_ech_orders, dZ, g_wh_inv, g_ech_inv, E, Ww,
                    We, B):
    _dev(Z)
    check(lib().nic_head_serial_bwd(ptr(Z), ptr(wh_inv), ptr(ech_inv), float(ub), ptr(g_store_orders), ptr(g_wh_orders),
                                    ptr(g_ech_orders), ptr(dZ), ptr(g_wh_inv), ptr(g_ech_inv), E, Ww, We, B, Z.stride(0),
                                    current_stream()))


# ---- fused 3-layer MLP over gathered inputs + segment sums (graph policies) -----------------------------------------

class Mlp3Segment:
    """One group of input rows of an MLP3 launch: rows of `tensor` ([rows][n_src_entities][ldb], or [rows][n_src] for a
    per-entity constant with per_scenario=False) gathered through `index` (int32 device tensor [n_entities], -1 = zeros; None =
    identity)."""

    def __init__(self, tensor, index=None, per_scenario=True):
        self.tensor, self.index, self.per_scenario = tensor, index, per_scenario


def mlp3_pack_transposed(linears_or_tensors, n_out):
    """[W1^T][b1][W2^T][b2][W3^T padded to 32 columns][b3 padded to 32] (see NicMlp3Desc.weights_t)."""
    (w1, b1), (w2, b2), (w3, b3) = linears_or_tensors
    w3t = torch.zeros(32, 32, device=w3.device)
    w3t[:, :n_out] = w3.detach().t()
    b3p = torch.zeros(32, device=w3.device)
    b3p[:n_out] = b3.detach()
    return torch.cat([w1.detach().t().reshape(-1), b1.detach(), w2.detach().t().reshape(-1), b2.detach(), w3t.reshape(-1), b3p])


def mlp3_desc(segments, weights, n_entities, n_scenarios, ldb, n_out, out_act, hist_row_stride=0, weights_t=None,
              hist_native=False, ent_row_stride=0):
    d = _lib.NicMlp3Desc()
    d.hist_row_stride = int(hist_row_stride)
    d.hist_native = int(bool(hist_native))
    d.ent_row_stride = int(ent_row_stride)
    d.weights_t = weights_t.data_ptr() if weights_t is not None else None
    d.n_entities, d.n_scenarios, d.ldb = n_entities, n_scenarios, ldb
    d.n_out, d.out_act, d.n_segs = n_out, out_act, len(segments)
    k = 0
    for i, sgm in enumerate(segments):
        t = sgm.tensor
        s = d.seg[i]
        s.base, s.map = t.data_ptr(), (sgm.index.data_ptr() if sgm.index is not None else None)
        s.row_stride = t.stride(0)
        s.ent_stride = t.stride(1) if t.shape[1] > 1 else 0
        s.scn_stride = 1 if sgm.per_scenario else 0
        s.n_rows = t.shape[0]
        k += t.shape[0]
    d.K = k
    d.weights = weights.data_ptr()
    d._keep = (segments, weights, weights_t)
    return d


def mlp3_fwd(desc, Y, X_hist=None, H1=None, H2=None, residual=None, Ysum=None):
    """Y = MLP(gathered inputs); with `residual` / `Ysum` also Ysum = residual + Y in the same launch."""
    _dev(Y)
    check(lib().nic_mlp3_fwd_residual(desc, ptr(Y), ptr(X_hist), ptr(H1), ptr(H2), ptr(residual), ptr(Ysum), current_stream()))
    return Y


def mlp3_bwd_fused_slots():
    return lib().nic_mlp3_bwd_fused_slots()


def mlp3_bwd_fused(desc, dY, Y, dX, slabs):
    """History-free backward: recompute + in-kernel weight gradients into slabs [slots][N][lds] (3 layers)."""
    _dev(dY)
    s1, s2, s3 = slabs
    check(lib().nic_mlp3_bwd_fused(desc, ptr(dY), ptr(Y), ptr(dX), ptr(s1), s1.stride(1), ptr(s2), s2.stride(1), ptr(s3),
                                   s3.stride(1), current_stream()))


def mlp3_bwd_hist_slots():
    return lib().nic_mlp3_bwd_hist_slots()


def mlp3_bwd_hist(desc, dY, Y, X_hist, H1, H2, dX, slabs):
    """Backward over the stored activations with in-kernel weight gradients into slabs [slots][N][lds] (3 layers)."""
    _dev(dY)
    s1, s2, s3 = slabs
    check(lib().nic_mlp3_bwd_hist(desc, ptr(dY), ptr(Y), ptr(X_hist), ptr(H1), ptr(H2), ptr(dX), ptr(s1), s1.stride(1), ptr(s2),
                                  s2.stride(1), ptr(s3), s3.stride(1), current_stream()))


# ---- the GNN policy's period in one launch (csrc/gnn_period.hip) -------------------------------------------------------------

GNN_PERIOD_MLPS = ("initial_node", "initial_edge", "node_update", "edge_update", "output")


def gnn_period_ok(n_nodes, n_edges, Dn):
    """1 if a graph's embeddings and the staged weights fit in a workgroup's LDS, 2 if the node embeddings do and the edge tiles go
    to a scratch area in global memory (`gnn_period_edge_scratch_floats`), 0 if the period kernel cannot take the graph."""
    return int(lib().nic_gnn_period_ok(int(n_nodes), int(n_edges), int(Dn)))


def gnn_period_edge_scratch_floats(n_edges, n_scenarios):
    """floats of `NicGnnPeriod.edge_scratch`: one 2-KB tile per (16-scenario block, edge)"""
    return ((int(n_scenarios) + 15) // 16) * int(n_edges) * 512


def _frag32(base):
    """contraction steps of a 32-row operand held as an embedding tile: step s, lane group g -> row base + 16 (s >> 2) + 4 g + (s & 3)"""
    return [[base + 16 * (s >> 2) + 4 * g + (s & 3) for g in range(4)] for s in range(8)]


def gnn_period_kmaps(Dn):
    """First-layer contraction order of the five MLPs (which input row lane group g feeds into step s; -1: none), each padded to
    groups of four steps - the layout `nic_gnn_period_fwd` contracts in (include/nic_rollout.h)."""
    pad = lambda steps: steps + [[-1] * 4] * (-len(steps) % 4)  # noqa: E731
    node = [[4 * s_ + g if 4 * s_ + g < Dn else -1 for g in range(4)] for s_ in range(4 * ((Dn + 15) // 16))]
    return {"initial_node": node,
            "initial_edge": pad(_frag32(0) + _frag32(32) + [[64, -1, -1, -1]]),
            "node_update": _frag32(0) + _frag32(32) + _frag32(64),
            "edge_update": _frag32(0) + _frag32(32) + _frag32(64),
            "output": _frag32(0)}


def gnn_period_pack_index(kmap, K, nrb, device):
    """Gather index into a [16 nrb][K + 1] weight matrix (last column zero) that lays a layer out as MFMA A fragments
    [rb][group of 4 steps][lane][4]: (rb, q, lane, j) -> W[16 rb + (lane & 15)][kmap[4 q + j][lane >> 4]]."""
    sq = len(kmap) // 4
    lane = torch.arange(64)
    km = torch.tensor(kmap, dtype=torch.long)                       # [steps][4]
    km = torch.where(km < 0, torch.full_like(km, K), km)
    col = km.view(sq, 4, 4)[:, :, lane >> 4].permute(0, 2, 1)       # [sq][lane][j]
    row = (16 * torch.arange(nrb).view(nrb, 1) + (lane & 15).view(1, 64))   # [nrb][lane]
    idx = row.view(nrb, 1, 64, 1) * (K + 1) + col.view(1, sq, 64, 4)
    return idx.reshape(-1).to(device)


class GnnPeriodPack:
    """Packed weights of one MLP for `nic_gnn_period_fwd`, refreshed from the live parameters by `pack()`."""

    def __init__(self, linears, kmap, n_out, device):
        self.linears = linears
        K = linears[0].weight.shape[1]
        self.K, self.n_out, self.nrb3 = K, n_out, (2 if n_out > 16 else 1)
        self.s1q = len(kmap) // 4
        self.idx = [gnn_period_pack_index(kmap, K, 2, device), gnn_period_pack_index(_frag32(0), 32, 2, device),
                    gnn_period_pack_index(_frag32(0), 32, self.nrb3, device)]
        n = lib().nic_gnn_period_pack_size(self.s1q, n_out)
        self.buf = torch.zeros(n, device=device)
        self._pad = [torch.zeros(32, K + 1, device=device), torch.zeros(32, 33, device=device),
                     torch.zeros(16 * self.nrb3, 33, device=device)]
        assert n == sum(i.numel() for i in self.idx) + 96

    def pack(self):
        o = 0
        for i, lin in enumerate(self.linears):
            w, b = lin.weight.detach(), lin.bias.detach()
            pad = self._pad[i]
            pad[:w.shape[0], :w.shape[1]] = w
            n = self.idx[i].numel()
            torch.index_select(pad.view(-1), 0, self.idx[i], out=self.buf[o:o + n])
            o += n
            self.buf[o:o + b.numel()] = b
            o += 32
        return self.buf


def gnn_period_fwd(desc):
    _lib.require_device()
    check(lib().nic_gnn_period_fwd(desc, current_stream()))


class GnnPeriodBwdPack:
    """TRANSPOSED weights of one MLP as MFMA A fragments for `nic_gnn_period_bwd` (include/nic_rollout.h): [L3^T][L2^T][L1^T of
    every 32-row input segment] - the input gradient of a layer is W^T dz, contracted over the layer's 32 outputs in the order an
    embedding tile holds them.  `segments` = number of 32-row blocks of the first layer's inputs that get an input gradient (the
    edge MLP's lead-time row gets none; node features of fewer than 32 rows are one zero-padded block)."""

    def __init__(self, linears, n_out, segments, device):
        self.linears, self.n_out, self.segments = linears, n_out, segments
        K1 = linears[0].weight.shape[1]
        one = [[0, -1, -1, -1]] + [[-1] * 4] * 3
        self.idx3 = gnn_period_pack_index(_frag32(0), 32, 2, device) if n_out == 32 else gnn_period_pack_index(one, 1, 2, device)
        self.idx = gnn_period_pack_index(_frag32(0), 32, 2, device)
        n = self.idx3.numel() + (1 + segments) * 1024   # (include/nic_rollout.h: (n_out == 1 ? 512 : 1024) + 1024 (1 + segments))
        self.buf = torch.zeros(n, device=device)
        self._pad3 = torch.zeros(32, (32 if n_out == 32 else 1) + 1, device=device)
        self._pad = torch.zeros(32, 33, device=device)
        self.seg_rows = [min(32, K1 - 32 * s) for s in range(segments)]

    def pack(self):
        w1, w2, w3 = (lin.weight.detach() for lin in self.linears)
        self._pad3[:, :self.n_out] = w3.t()
        o = self.idx3.numel()
        torch.index_select(self._pad3.view(-1), 0, self.idx3, out=self.buf[:o])
        self._pad[:, :32] = w2.t()
        torch.index_select(self._pad.view(-1), 0, self.idx, out=self.buf[o:o + 1024])
        o += 1024
        for s, rows in enumerate(self.seg_rows):
            self._pad.zero_()
            self._pad[:rows, :32] = w1[:, 32 * s:32 * s + rows].t()
            torch.index_select(self._pad.view(-1), 0, self.idx, out=self.buf[o:o + 1024])
            o += 1024
        return self.buf


class WeightPackPlan:
    """Every packed-weight buffer of an engine refreshed from the live parameters by TWO launches (one concatenation of the parameters,
    one gather) instead of the five-to-ten small launches each packer's own `pack()` makes (the GNN step had ~150 of them, 0.75 ms of
    a 21 ms step).  Built by running each packer once on stand-in parameters that hold their own position in the concatenation: what
    lands in the packer's buffers is then the gather index (0 = a padding zero).  `items`: (packer, names of its buffers); a packer
    has `.linears` and `.pack()`; its buffers become views of one allocation (256-byte aligned)."""

    def __init__(self, items, device):
        import types
        lins, seen = [], set()
        for obj, _ in items:
            for lin in obj.linears:
                if id(lin) not in seen:
                    seen.add(id(lin))
                    lins.append(lin)
        self.lins = lins   # (the layers, not their parameters: a parameter replaced on a layer later is still the one that is read)
        sizes, stand_in, o = [], {}, 1   # (position 0 of the concatenation is the zero every padding element reads)
        for lin in lins:
            w, b = lin.weight, lin.bias
            fw = torch.arange(o, o + w.numel(), dtype=torch.float32, device=device).view_as(w)
            o += w.numel()
            fb = torch.arange(o, o + b.numel(), dtype=torch.float32, device=device).view_as(b)
            o += b.numel()
            stand_in[id(lin)] = types.SimpleNamespace(weight=fw, bias=fb)
            sizes += [tuple(w.shape), tuple(b.shape)]
        self._sizes = sizes
        assert o < 2 ** 24   # (positions travel through the packers as float32)
        self.zero = torch.zeros(1, device=device)
        parts, places, n = [], [], 0
        for obj, names in items:
            real = obj.linears
            obj.linears = [stand_in[id(lin)] for lin in real]
            try:
                obj.pack()
            finally:
                obj.linears = real
            for name in names:
                t = getattr(obj, name)
                idx = t.detach().round().long().reshape(-1)
                pad = (-idx.numel()) % 64
                parts.append(torch.cat([idx, idx.new_zeros(pad)]))
                places.append((obj, name, n, t.shape))
                n += idx.numel() + pad
        self.index = torch.cat(parts)
        self.buf = torch.zeros(n, device=device)
        for obj, name, at, shape in places:
            setattr(obj, name, self.buf[at:at + math.prod(shape)].view(shape))

    def pack(self):
        params = [p for lin in self.lins for p in (lin.weight, lin.bias)]
        if [tuple(p.shape) for p in params] != self._sizes:
            raise RuntimeError("WeightPackPlan: a layer's shape changed since the plan was built")
        flat = torch.cat([self.zero] + [p.detach().reshape(-1) for p in params])
        torch.index_select(flat, 0, self.index, out=self.buf)
        return self.buf


def gnn_period_bwd_scratch_floats(n_nodes, n_edges, n_live, n_scenarios, n_sub):
    return int(lib().nic_gnn_period_bwd_scratch_floats(int(n_nodes), int(n_edges), int(n_live), int(n_scenarios), int(n_sub)))


def gnn_period_bwd(desc):
    _lib.require_device()
    check(lib().nic_gnn_period_bwd(desc, current_stream()))


def segment_sum_terms(dst, terms, accumulate=False):
    """dst [R][n_dst][ldb] = (accumulate ? dst : 0) + sum over terms, in order.  A term is (src, offsets, items, scale) - a segment
    sum of src's [R][.][ldb] rows - or (src, None, None, scale): src's own row n.  One launch (nic_segment_sum_terms)."""
    _dev(dst)
    R, n_dst, ldb = dst.shape
    arr = (_lib.NicSegTerm * len(terms))()
    for j, (src, offsets, items, scale) in enumerate(terms):
        arr[j].src, arr[j].src_row_stride = ptr(src), src.stride(0)
        arr[j].offsets, arr[j].items, arr[j].scale = ptr(offsets), ptr(items), ptr(scale)
    check(lib().nic_segment_sum_terms(ptr(dst), dst.stride(0), arr, len(terms), R, n_dst, ldb, ldb, int(accumulate), current_stream()))
    return dst


def gnn_alloc_fwd(out, on_hand, orders, sums, ratio, scale, S, e_self, e_supplier, cap_at_one, n_scenarios):
    """Proportional allocation head of the GNN policy (one launch): see nic_gnn_alloc_fwd."""
    _dev(out)
    check(lib().nic_gnn_alloc_fwd(ptr(out), ptr(on_hand), ptr(orders), ptr(sums), ptr(ratio), ptr(scale), S,
                                  -1 if e_self is None else e_self, e_supplier, int(cap_at_one), n_scenarios, out.stride(0),
                                  current_stream()))


def gnn_alloc_bwd(out, on_hand, g_orders, sums, ratio, scale, d_out, g_on_hand, S, e_self, e_supplier, cap_at_one, n_scenarios):
    _dev(out)
    check(lib().nic_gnn_alloc_bwd(ptr(out), ptr(on_hand), ptr(g_orders), ptr(sums), ptr(ratio), ptr(scale), ptr(d_out),
                                  ptr(g_on_hand), S, d_out.shape[0], -1 if e_self is None else e_self, e_supplier, int(cap_at_one),
                                  n_scenarios, out.stride(0), current_stream()))


def gnn_alloc_env_fwd(prob, state, demand: Table, out, orders, sums, ratio, scale, e_self, e_supplier, cap_at_one, next_state, reward):
    """nic_gnn_alloc_env_fwd: allocation head + env step in one launch (one warehouse).  orders [S + 1][ldb]."""
    _dev(out)
    S, ld = prob.S, prob.ldb
    io = prob.make_io(state.store, state.wh, None, demand, Table(orders[:S].view(S, 1, -1), ld, 1, ld), Table(orders[S:], ld, 1), None)
    check(lib().nic_gnn_alloc_env_fwd(io, ptr(out), ptr(orders), ptr(sums), ptr(ratio), ptr(scale), -1 if e_self is None else e_self,
                                      e_supplier, int(cap_at_one), ptr(next_state.store), ptr(next_state.wh), ptr(reward),
                                      current_stream()))


def gnn_alloc_env_bwd(prob, state, demand: Table, out, orders, sums, ratio, scale, e_self, e_supplier, cap_at_one, g_out, g_reward: Table,
                      g_in, g_orders, d_out):
    """nic_gnn_alloc_env_bwd: env-step adjoint + allocation adjoint in one launch.  g_out / g_in: EnvState of gradients."""
    _dev(out)
    S, ld = prob.S, prob.ldb
    io = prob.make_io(state.store, state.wh, None, demand, Table(orders[:S].view(S, 1, -1), ld, 1, ld), Table(orders[S:], ld, 1), None)
    check(lib().nic_gnn_alloc_env_bwd(io, ptr(out), ptr(sums), ptr(ratio), ptr(scale), d_out.shape[0], -1 if e_self is None else e_self,
                                      e_supplier, int(cap_at_one), ptr(g_out.store), ptr(g_out.wh), g_reward.t2(), ptr(g_in.store),
                                      ptr(g_in.wh), ptr(g_orders), ptr(d_out), current_stream()))


def gnn_alloc_groups_fwd(out, on_hand, orders, sums, ratio, scale, groups, order_row, cap_at_one, n_scenarios):
    """Proportional allocation of several warehouses' stock, one launch (nic_gnn_alloc_groups_fwd).  on_hand [G][Ww][ldb]:
    slot 0 of every group's pipeline; sums / ratio / scale [G][ldb]."""
    _dev(out)
    check(lib().nic_gnn_alloc_groups_fwd(ptr(out), ptr(on_hand), on_hand.stride(0), ptr(orders), ptr(sums), ptr(ratio), ptr(scale),
                                         ptr(groups), ptr(order_row), groups.shape[0], int(cap_at_one), n_scenarios, out.stride(0),
                                         current_stream()))


def gnn_alloc_groups_bwd(out, on_hand, g_orders, sums, ratio, scale, d_out, g_on_hand, groups, order_row, zero_first, zero_count,
                         cap_at_one, n_scenarios):
    _dev(out)
    assert g_on_hand.stride(0) == on_hand.stride(0)
    check(lib().nic_gnn_alloc_groups_bwd(ptr(out), ptr(on_hand), on_hand.stride(0), ptr(g_orders), ptr(sums), ptr(ratio), ptr(scale),
                                         ptr(d_out), ptr(g_on_hand), ptr(groups), ptr(order_row), groups.shape[0], zero_first,
                                         zero_count, int(cap_at_one), n_scenarios, out.stride(0), current_stream()))


def segment_sum(dst, src, offsets, items, dst_scale=None, accumulate=False):
    """dst [R][n_dst][ldb] (+)= dst_scale[n] * sum over items[offsets[n]:offsets[n+1]] of src[R][.][ldb] rows."""
    _dev(dst)
    R, n_dst, ldb = dst.shape
    check(lib().nic_segment_sum(ptr(dst), dst.stride(0), ptr(src), src.stride(0), ptr(offsets), ptr(items), ptr(dst_scale), R,
                                n_dst, ldb, ldb, int(accumulate), current_stream()))
    return dst


# ---- sampler / utilities ---------------------------------------------------------------------------------------

def sample_demand(out, T, S, n_scenarios, scenario_offset, seed, kind, mean, chol, clip):
    """out: [T][S][ldb] device tensor; kind 0 normal (chol: [S][S] lower factor), 1 poisson."""
    _dev(out)
    check(lib().nic_sample_demand(ptr(out), T, S, n_scenarios, out.stride(1), int(scenario_offset), int(seed), kind,
                                  ptr(mean), ptr(chol), int(clip), current_stream()))
    return out


def sample_demand_equicorrelated(out, T, S, n_scenarios, scenario_offset, seed, mean, std, rho, clip):
    """out: [T][S][ldb]; normal demand with cov_ij = rho std_i std_j off the diagonal (the reference's covariance)."""
    _dev(out)
    check(lib().nic_sample_demand_equicorrelated(ptr(out), T, S, n_scenarios, out.stride(1), int(scenario_offset), int(seed),
                                                 ptr(mean), ptr(std), float(rho), int(clip), current_stream()))
    return out
