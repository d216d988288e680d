"""Policy-head bodies (policy_heads_body.h) on the host against the oracle's head arithmetic and its autograd."""
import pytest
import torch
import torch.nn.functional as F

import hostsim_util
from neural_inventory_control_amd.layout import to_soa, ref_view, pad_ld
from oracle import inventory_oracle as orc

P = lambda x: x.data_ptr() if x is not None else None  # noqa: E731


@pytest.mark.parametrize("S,Wn,adj", [
    (16, 1, None),
    (10, 2, [[0, 0, 1, 1, 1, 0, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 0, 1, 1, 1]]),
    (8, 3, [[1, 1, 0, 0, 1, 0, 1, 0], [0, 1, 1, 1, 0, 0, 1, 1], [1, 0, 0, 1, 0, 1, 0, 1]]),
    (4, 2, [[1, 1, 1, 1], [0, 0, 0, 0]]),  # a warehouse without any connected store
])
@pytest.mark.parametrize("trans", [False, True])
def test_warehouse_head(S, Wn, adj, trans):
    h = hostsim_util.load()
    B, Ww = 37, 3
    ldb = pad_ld(B)
    gen = torch.Generator().manual_seed(3)
    Z = (torch.randn(B, S * Wn + Wn, generator=gen) * 3).requires_grad_(True)
    wh = (torch.rand(B, Wn, Ww, generator=gen) * 20)
    wh[0, :, 0] = 0.0  # empty warehouse
    wh.requires_grad_(True)
    ub = 123.5
    pol = orc.OraclePolicy("vanilla_warehouse", [], "elu", None, torch.tensor([ub]), adj, trans)
    adj_t = torch.ones(1, S) if Wn == 1 else torch.tensor(adj, dtype=torch.float32)
    # oracle arithmetic of neural_networks.py:393-426 on given logits
    store_logits = Z[:, :S * Wn].view(-1, S, Wn)
    alloc = torch.zeros_like(store_logits)
    for w in range(Wn):
        conn = adj_t[w].nonzero(as_tuple=True)[0]
        if len(conn) > 0:
            alloc[:, conn, w] = orc._softmax_share_of_stock(store_logits[:, conn, w], wh[:, w:w + 1], trans)
    wh_orders = torch.sigmoid(Z[:, S * Wn:]) * pol.warehouse_upper_bound
    g_so = torch.randn(B, S, Wn, generator=gen)
    g_wo = torch.randn(B, Wn, generator=gen)
    ((alloc * g_so).sum() + (wh_orders * g_wo).sum()).backward()

    Zs, whs = to_soa(Z.detach(), ldb), to_soa(wh.detach(), ldb)
    adj_i = adj_t.to(torch.int32).contiguous()
    so, wo = torch.zeros(S, Wn, ldb), torch.zeros(Wn, ldb)
    h.hostsim_head_warehouse_fwd(P(Zs), P(whs), P(adj_i), ub, int(trans), P(so), P(wo), S, Wn, Ww, B, ldb)
    torch.testing.assert_close(ref_view(so, B), alloc.detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(ref_view(wo, B), wh_orders.detach(), rtol=2e-6, atol=1e-6)
    # structurally-zero orders must be EXACT zeros (the env's `!= 0` filter depends on it)
    assert torch.equal(ref_view(so, B) == 0, alloc.detach() == 0)
    dZ = torch.zeros(S * Wn + Wn, ldb)
    gwi = torch.zeros(Wn, Ww, ldb)
    gso_s, gwo_s = to_soa(g_so, ldb), to_soa(g_wo, ldb)  # keep alive across the call
    h.hostsim_head_warehouse_bwd(P(Zs), P(whs), P(adj_i), ub, int(trans), P(gso_s), P(gwo_s),
                                 P(dZ), P(gwi), S, Wn, Ww, B, ldb)
    torch.testing.assert_close(ref_view(dZ, B), Z.grad, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(ref_view(gwi, B), wh.grad, rtol=2e-5, atol=2e-6)


def test_softplus_head():
    h = hostsim_util.load()
    B, ldb = 50, 64
    Z = torch.linspace(-30, 30, B).reshape(B, 1).clone().requires_grad_(True)
    y = F.softplus(Z + 1)
    g = torch.randn(B, 1)
    (y * g).sum().backward()
    out, dZ = torch.zeros(1, ldb), torch.zeros(1, ldb)
    Zs = to_soa(Z.detach(), ldb)
    h.hostsim_head_softplus_fwd(P(Zs), P(out), 1, B, ldb)
    gs = to_soa(g, ldb)
    h.hostsim_head_softplus_bwd(P(Zs), P(gs), P(dZ), 1, B, ldb)
    torch.testing.assert_close(ref_view(out, B), y.detach(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(ref_view(dZ, B), Z.grad, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("E", [2, 1, 3])
def test_serial_head(E):
    h = hostsim_util.load()
    B, Ww, We = 41, 3, 4
    ldb = pad_ld(B)
    gen = torch.Generator().manual_seed(5)
    Z = torch.randn(B, E + 2, generator=gen).requires_grad_(True)
    wh = (torch.rand(B, 1, Ww, generator=gen) * 9).requires_grad_(True)
    ech = (torch.rand(B, E, We, generator=gen) * 9).requires_grad_(True)
    ub = 20.0
    upstream = torch.concat((torch.tensor([ub]).unsqueeze(1).expand(B, -1), ech[:, :, 0], wh[:, :, 0]), dim=1)
    alloc = torch.sigmoid(Z) * upstream  # neural_networks.py:335-344
    g = torch.randn(B, E + 2, generator=gen)
    (alloc * g).sum().backward()
    Zs, whs, echs = to_soa(Z.detach(), ldb), to_soa(wh.detach(), ldb), to_soa(ech.detach(), ldb)
    so, wo, eo = torch.zeros(1, 1, ldb), torch.zeros(1, ldb), torch.zeros(E, ldb)
    h.hostsim_head_serial_fwd(P(Zs), P(whs), P(echs), ub, P(so), P(wo), P(eo), E, Ww, We, B, ldb)
    torch.testing.assert_close(ref_view(eo, B), alloc.detach()[:, :E], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(wo[0, :B], alloc.detach()[:, E], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(so[0, 0, :B], alloc.detach()[:, E + 1], rtol=2e-6, atol=1e-7)
    dZ, gwi, gei = torch.zeros(E + 2, ldb), torch.zeros(1, Ww, ldb), torch.zeros(E, We, ldb)
    gs = to_soa(g, ldb)
    g_store, g_wh, g_ech = gs[E + 1:], gs[E:E + 1], gs[:E]
    h.hostsim_head_serial_bwd(P(Zs), P(whs), P(echs), ub, P(g_store), P(g_wh), P(g_ech), P(dZ), P(gwi), P(gei),
                              E, Ww, We, B, ldb)
    torch.testing.assert_close(ref_view(dZ, B), Z.grad, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(ref_view(gwi, B), wh.grad, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(ref_view(gei, B), ech.grad, rtol=2e-5, atol=1e-6)
