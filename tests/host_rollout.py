"""TEST INFRASTRUCTURE: the per-period rollout + analytic backward sweep of `FusedRollout` (rollout.py) re-played on the
CPU with the HOST build of the kernel bodies (tests/hostsim: env step + policy heads, the same NIC_HD code the HIP
kernels run) and plain torch matmuls in place of the MFMA GEMMs.

It exists so that the *composition* of the kernels — the order of the backward sweep, which gradient buffer each kernel
accumulates into, the engine's zero-order / clamp tie rules along a whole trajectory — is checked against the reference's
golden gradients on the CPU container too, and so that gradient tolerances can be studied without a GPU (knife-edge
scenarios, fp64 referee).  The product never imports it.
"""
import torch

from neural_inventory_control_amd import layout
from neural_inventory_control_amd.layout import EnvProblem, Table, pad_ld, to_soa

P = lambda x: x.data_ptr() if x is not None else None  # noqa: E731


def _views(block, prob, F_store, F_wh):
    a, b = F_store, F_store + F_wh
    store = block[:a].view(prob.S, prob.Ws, -1)
    wh = block[a:b].view(prob.Wn, prob.Ww, -1) if prob.Wn else None
    ech = block[b:].view(prob.E, prob.We, -1) if prob.E else None
    return store, wh, ech


def _order_views(block, prob):
    a, b = prob.S * prob.nsup, prob.S * prob.nsup + prob.Wn
    return (block[:a].view(prob.S, prob.nsup, -1), block[a:b] if prob.Wn else None, block[b:] if prob.E else None)


def _order_tables(block, prob):
    so, wo, eo = _order_views(block, prob)
    ld = prob.ldb
    return (Table(so, prob.nsup * ld, 1, ld), Table(wo, ld, 1) if wo is not None else None,
            Table(eo, ld, 1) if eo is not None else None)


def run(be, problem_params, data, layers, head, periods, ignore=0, ub=0.0, adjacency=None, transshipment=False,
        grad_scale=None, period_shift=0):
    """layers: [(W, b)] float32 CPU tensors of the master MLP (ELU between layers, none after the last).
    head: 'warehouse' | 'serial' | 'softplus'.  Returns dict(total, reported, rewards (T,B), grads [(dW, db)], final)."""
    dev = be.device
    data = {k: v.to(dev) for k, v in data.items()}
    layers = [(W.to(dev), b.to(dev)) for W, b in layers]
    prob = EnvProblem(problem_params, data, dev)
    B, ld, T = prob.B, prob.ldb, periods
    F_store, F_wh, F_ech = prob.S * prob.Ws, prob.Wn * prob.Ww, prob.E * prob.We
    f_tot = F_store + F_wh + F_ech
    F = F_store if head == "softplus" else f_tot
    n_ord = prob.S * prob.nsup + prob.Wn + prob.E
    L = len(layers)
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    states, orders, rewards = z(T + 1, f_tot, ld), z(T, n_ord, ld), z(T, ld)
    hidden = [z(T, layers[i][0].shape[0], ld) for i in range(L - 1)]
    logits = z(T, layers[-1][0].shape[0], ld)
    d = data["demands"]
    demand = z(d.shape[2], d.shape[1], ld)
    demand[:, :, :B] = d.permute(2, 1, 0)
    s0 = _views(states[0], prob, F_store, F_wh)
    s0[0][:, :, :B] = data["initial_inventories"].permute(1, 2, 0)
    if prob.Wn:
        s0[1][:, :, :B] = data["initial_warehouse_inventories"].permute(1, 2, 0)
    if prob.E:
        s0[2][:, :, :B] = data["initial_echelon_inventories"].permute(1, 2, 0)
    adj = None
    if head == "warehouse":
        adj = (torch.ones(1, prob.S) if prob.Wn == 1 else torch.tensor(adjacency, dtype=torch.float32) != 0)
        adj = adj.to(torch.int32).contiguous().to(dev)
    mask = z(ld)
    mask[:B] = 1.0

    def head_fwd(Z, st, row):
        so, wo, eo = _order_views(orders[row], prob)
        if head == "warehouse":
            be.head_warehouse_fwd(Z, st[1], adj, float(ub), int(transshipment), so, wo, prob.S, prob.Wn, prob.Ww, B, ld)
        elif head == "serial":
            be.head_serial_fwd(Z, st[1], st[2], float(ub), so, wo, eo, prob.E, prob.Ww, prob.We, B, ld)
        else:
            be.head_softplus_fwd(Z, so.view(-1, ld), prob.S * prob.nsup, B, ld)

    for t in range(T):
        st = _views(states[t], prob, F_store, F_wh)
        x = states[t][:F]
        for i, (W, b) in enumerate(layers):
            y = (W @ x + b[:, None]) * mask  # padding columns stay zero, like the kernels' epilogues
            if i < L - 1:
                y = torch.nn.functional.elu(y)
                hidden[i][t] = y
            x = y
        logits[t] = x
        head_fwd(logits[t], st, t)
        ts, tw, te = _order_tables(orders[t], prob)
        io = prob.make_io(st[0], st[1], st[2], Table(demand[t + period_shift], ld, 1), ts, tw, te)
        nx = _views(states[t + 1], prob, F_store, F_wh)
        be.env_fwd(io, nx[0], nx[1], nx[2], rewards[t])
    be.sync()
    total = rewards.sum()
    reported = rewards[ignore:].sum()

    # ---- backward sweep (rollout.py::_launch_backward) ---------------------------------------------------------------
    if grad_scale is None:
        grad_scale = 1.0 / (B * T * problem_params["n_stores"])
    g_reward = z(ld)
    g_reward[:B] = grad_scale
    g_next, g_cur = z(f_tot, ld), z(f_tot, ld)
    g_orders = z(n_ord, ld)
    gW = [torch.zeros_like(W, dtype=torch.float64) for W, _ in layers]
    gb = [torch.zeros_like(b, dtype=torch.float64) for _, b in layers]
    detached_input = head == "serial"
    for t in range(T - 1, -1, -1):
        st = _views(states[t], prob, F_store, F_wh)
        ts, tw, te = _order_tables(orders[t], prob)
        io = prob.make_io(st[0], st[1], st[2], Table(demand[t + period_shift], ld, 1), ts, tw, te)
        gn, gc = _views(g_next, prob, F_store, F_wh), _views(g_cur, prob, F_store, F_wh)
        gso, gwo, geo = _order_views(g_orders, prob)
        be.env_bwd(io, gn[0], gn[1], gn[2], layout.Table(g_reward, 0, 1).t2(), gc[0], gc[1], gc[2], gso, gwo, geo)
        Z = logits[t]
        dZ = z(*Z.shape)
        if head == "warehouse":
            be.head_warehouse_bwd(Z, st[1], adj, float(ub), int(transshipment), gso, gwo, dZ, gc[1], prob.S, prob.Wn,
                                  prob.Ww, B, ld)
        elif head == "serial":
            be.head_serial_bwd(Z, st[1], st[2], float(ub), gso, gwo, geo, dZ, gc[1], gc[2], prob.E, prob.Ww, prob.We, B, ld)
        else:
            be.head_softplus_bwd(Z, gso.view(-1, ld), dZ, prob.S * prob.nsup, B, ld)
        dcur = dZ
        for i in range(L - 1, -1, -1):
            x_in = hidden[i - 1][t] if i > 0 else states[t][:F]
            gW[i] += dcur.double() @ x_in.double().t()
            gb[i] += dcur.double().sum(dim=1)
            if i > 0:
                dx = layers[i][0].t() @ dcur
                dcur = torch.where(x_in > 0, dx, dx * (x_in + 1))  # elu' recovered from the output
            elif not detached_input:
                g_cur[:F] += layers[0][0].t() @ dcur
        g_next, g_cur = g_cur, g_next
    be.sync()
    fin = _views(states[T], prob, F_store, F_wh)
    final = {"store_inventories": layout.ref_view(fin[0], B)}
    if prob.Wn:
        final["warehouse_inventories"] = layout.ref_view(fin[1], B)
    if prob.E:
        final["echelon_inventories"] = layout.ref_view(fin[2], B)
    return dict(total=total, reported=reported, rewards=rewards[:, :B], final=final, states=states, orders=orders,
                logits=logits, hidden=hidden, prob=prob, grads=[t.float().cpu() for pair in zip(gW, gb) for t in pair])
