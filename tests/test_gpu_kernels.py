"""Parity of every HIP kernel, called through the C ABI (libnic_hip.so), on a real MI355X.

env step / policy heads: the same checks as the host-sim build (golden vectors + oracle autograd);
policy GEMMs: against float64 torch on the same inputs (FP32 MFMA = exact f32 products, f32 accumulation);
sampler: moments + sharding invariance (statistical parity with numpy)."""
import numpy as np
import pytest
import torch

import kernel_checks as kc
from golden_io import case_names
from neural_inventory_control_amd import _lib, ops
from neural_inventory_control_amd.layout import pad_ld

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return kc.HipBackend()


def test_library_loaded_and_device_visible():
    assert _lib.library_built()
    assert _lib.lib().nic_device_count() >= 1


@pytest.mark.parametrize("name", case_names())
def test_env_forward_matches_golden(be, name):
    kc.check_env_forward(be, name)


@pytest.mark.parametrize("name", case_names())
@pytest.mark.parametrize("profit", [False, True])
def test_env_backward_matches_oracle_autograd(be, name, profit):
    kc.check_env_backward(be, name, profit)


@pytest.mark.parametrize("S,Wn,adj", kc.WAREHOUSE_HEAD_CASES)
@pytest.mark.parametrize("trans", [False, True])
def test_warehouse_head(be, S, Wn, adj, trans):
    kc.check_warehouse_head(be, S, Wn, adj, trans)


def test_softplus_head(be):
    kc.check_softplus_head(be)


@pytest.mark.parametrize("E", [2, 1, 3])
def test_serial_head(be, E):
    kc.check_serial_head(be, E)


def test_zero_lead_orders_hand_computed_period(be):
    """Pins the "drop" semantics of the HIP env step (what ZERO_LEAD_CASES are compared against) with hand-computed numbers."""
    kc.check_zero_lead_micro(be)


@pytest.mark.parametrize("B,S,Wn", [(100, 5, 3), (64, 1, 2), (257, 9, 4), (1, 3, 2)])
def test_env_step_zero_lead_rule_inside_the_launch_equals_the_torch_patch(B, S, Wn):
    """nic_env_step_fwd / _bwd with zero_lead_upstream = 1 against the same launches with 0 followed by the torch formulation of
    the rule (`Simulator._add_zero_lead_orders` and its adjoint): several zero-lead suppliers per store, zero orders on some of them
    (filtered out, no gradient), B off the 64-scenario workgroup and the wrap from scenario B - 1 to scenario 0."""
    from neural_inventory_control_amd.layout import EnvProblem, Table
    from neural_inventory_control_amd.ops import EnvState
    DEV = "cuda"
    g = torch.Generator().manual_seed(B * 131 + S * 7 + Wn)
    Ws, Ww = 3, 4
    lead1 = torch.randint(0, 3, (S, Wn), generator=g).float()      # (0 on about a third of the pairs; the same for every scenario)
    lead1[0, 0] = 0.0
    problem = {"n_stores": S, "n_warehouses": Wn, "n_extra_echelons": 0, "lost_demand": True, "maximize_profit": False,
               "warehouse_store_adjacency": [[1] * S for _ in range(Wn)]}
    data = {"demands": torch.rand(B, S, 1, generator=g) * 5, "initial_inventories": torch.rand(B, S, Ws, generator=g) * 4,
            "underage_costs": torch.full((B, S), 9.0), "holding_costs": torch.full((B, S), 1.0),
            "lead_times": lead1.unsqueeze(0).expand(B, S, Wn).clone(),
            "initial_warehouse_inventories": torch.rand(B, Wn, Ww, generator=g) * 30,
            "warehouse_holding_costs": torch.full((B, Wn), 0.3), "warehouse_lead_times": torch.full((B, Wn), 3.0),
            "warehouse_edge_costs": torch.full((B, Wn), 0.5)}
    data = {k: v.to(DEV) for k, v in data.items()}
    prob = EnvProblem(problem, data, DEV)
    ld = prob.ldb
    a_store = torch.rand(B, S, Wn, generator=g) * 3
    a_store[torch.rand(B, S, Wn, generator=g) < 0.25] = 0.0
    a_store, a_wh = a_store.to(DEV), (torch.rand(B, Wn, generator=g) * 6).to(DEV)
    so = torch.zeros(S, Wn, ld, device=DEV)
    so[:, :, :B] = a_store.permute(1, 2, 0)
    wo = torch.zeros(Wn, ld, device=DEV)
    wo[:, :B] = a_wh.t()
    st = EnvState(torch.zeros(S, Ws, ld, device=DEV), torch.zeros(Wn, Ww, ld, device=DEV), None)
    st.store[:, :, :B] = data["initial_inventories"].permute(1, 2, 0)
    st.wh[:, :, :B] = data["initial_warehouse_inventories"].permute(1, 2, 0)
    dem = torch.zeros(S, ld, device=DEV)
    dem[:, :B] = data["demands"][:, :, 0].t()
    ts, tw = Table(so, Wn * ld, 1, ld), Table(wo, ld, 1)
    res = {}
    for zl in (False, True):
        out, reward = ops.env_step_fwd(prob, st, Table(dem, ld, 1), ts, tw, None, zero_lead_upstream=zl)
        res[zl] = (out.store.clone(), out.wh.clone(), reward.clone())
    hit = (data["lead_times"] == 0) & (a_store != 0)
    u = torch.zeros(B, S, device=DEV)                                # (B, S), summed over the suppliers left to right like the kernel
    for w in range(Wn):
        u = u + a_store[:, :, w] * hit[:, :, w]
    want = res[False][0].clone()
    if S > 1:
        want[:S - 1, Ws - 1, :B] += u[:, 1:].t()
    want[S - 1, Ws - 1, :B] += torch.roll(u[:, 0], -1)
    assert bool(hit.any())
    assert torch.equal(res[True][0], want) and torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    # backward: g_orders gains the gradient of the slot the order was added to, where the order is not 0
    g_next = EnvState(torch.zeros(S, Ws, ld, device=DEV), torch.zeros(Wn, Ww, ld, device=DEV), None)
    g_next.store[:, :, :B] = torch.randn(S, Ws, B, generator=g).to(DEV)
    g_next.wh[:, :, :B] = torch.randn(Wn, Ww, B, generator=g).to(DEV)
    g_reward = torch.zeros(ld, device=DEV)
    g_reward[:B] = 1.0 / B
    gres = {}
    for zl in (False, True):
        g_in, g_ord = ops.env_step_bwd(prob, st, Table(dem, ld, 1), ts, tw, None, g_next, Table(g_reward, 0, 1),
                                       zero_lead_upstream=zl)
        gres[zl] = (g_in.store.clone(), g_in.wh.clone(), g_ord[0].clone(), g_ord[1].clone())
    gt = torch.zeros(B, S, device=DEV)
    if S > 1:
        gt[:, 1:] = g_next.store[:S - 1, Ws - 1, :B].t()
    gt[:, 0] = torch.roll(g_next.store[S - 1, Ws - 1, :B], 1)
    want_g = gres[False][2].clone()
    want_g[:, :, :B] += (gt.unsqueeze(2) * hit).permute(1, 2, 0)
    assert torch.equal(gres[True][2], want_g)
    assert all(torch.equal(gres[True][i], gres[False][i]) for i in (0, 1, 3))


def test_env_rejects_bad_arguments(be):
    io = _lib.NicEnvStepIO()
    assert be.l.nic_env_step_fwd(io, None, None, None, None, 0, None) != 0
    assert b"n_scenarios" in be.l.nic_last_error() or b"null" in be.l.nic_last_error()


# ---- policy GEMMs -------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(512, 51, 1000), (512, 512, 2048 + 37), (17, 512, 777), (32, 4, 100), (1, 32, 130), (64, 64, 64),
               (200, 100, 300), (195, 393, 520), (4, 32, 4096),
               # multiples of 32 scenarios, wide layers: the 256 x 256 LDS-DMA kernels (8 waves, two-pass epilogue)
               (512, 512, 4096), (256, 200, 2048), (300, 512, 1024), (512, 256, 8192),
               # short contraction, >= 64 output rows: the streamed first-layer forward of thin_layer.hip
               (512, 51, 4096), (128, 7, 777), (96, 64, 2048), (64, 1, 100), (192, 33, 96),
               # thin outputs with up to 128 input features (GNN layers): multi-accumulator wgrad_small_kernel
               (32, 65, 4096), (17, 96, 1000), (32, 128, 2049), (5, 40, 300),
               # BASELINE cfg5's ragged layers at its per-GPU scenario count (+ a ragged tail): 393 rows in ONE 448-row block
               # (13 of 14 row tiles computed), 195 rows in a 256-row block (7 of 8), and the all-period weight gradients' tilings
               (512, 393, 32768 + 160), (195, 512, 32768 + 96), (195, 512, 4096), (512, 393, 4096),
               # the compacted logits layer of cfg5 (98 live rows of 195): 128 x 128 forward, 128 x 256 weight-gradient tiles
               (98, 512, 32768), (98, 512, 4096 + 64), (128, 512, 8192), (96, 200, 4096),
               # ... and its first layer on the live input rows (295 of 393): 128 x 320 / 128 x 384 weight-gradient tiles
               (512, 295, 32768), (512, 295, 4096 + 32), (512, 350, 8192), (256, 320, 4096),
               # round 4: 65..128 input rows under >= 192 output rows (the shipped many-warehouse setting's first layer, 512 x 66):
               # 256 x 128 weight-gradient tiles; 129..191 input rows: 256 x 256
               (512, 66, 1024), (512, 100, 2048 + 64), (256, 128, 4096), (300, 150, 1024)]


def _rand(shape, gen, dev, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dev)


def _close(got, want64, scale64, what):
    """|got - want| <= 4e-6 * sum|a||b| (f32 accumulation over K terms) elementwise."""
    err = (got.double().cpu() - want64).abs()
    bound = 4e-6 * scale64 + 1e-6
    bad = err > bound
    assert not bool(bad.any()), (what, float(err.max()), float(bound.min()), int(bad.sum()))


@pytest.mark.parametrize("N,K,B", [(512, 51, 4096), (512, 51, 1000), (128, 7, 777), (160, 50, 2049), (256, 1, 100), (512, 52, 64),
                                   (192, 33, 96)])
def test_linear_fwd_thin_in_equals_the_tiled_forward(N, K, B):
    """nic_linear_fwd_thin_in (short contraction, many output rows, transposed weights) against nic_linear_fwd: the same
    contraction order and the same ELU, so the same bits; float64 reference within the GEMM band; padding columns untouched."""
    dev = "cuda"
    gen = torch.Generator().manual_seed(N * 100 + K)
    ldb = pad_ld(B)
    X = _rand((K, ldb), gen, dev)
    W = _rand((N, K), gen, dev, 0.3)
    Wt = torch.zeros(K, (N + 31) // 32 * 32 + 32, device=dev)   # (a row stride larger than N)
    Wt[:, :N] = W.t()
    bias = _rand((N,), gen, dev)
    assert ops.linear_fwd_thin_in_ok(N, K) and not ops.linear_fwd_thin_in_ok(100, K) and not ops.linear_fwd_thin_in_ok(N, 53)
    for act in (_lib.NIC_ACT_ELU, _lib.NIC_ACT_NONE):
        for b_ in (bias, None):
            want = torch.full((N, ldb), float("nan"), device=dev)
            got = torch.full((N, ldb), float("nan"), device=dev)
            ops.linear_fwd(W, b_, X, want, B, act)
            ops.linear_fwd_thin_in(Wt[:, :N], b_, X, got, B, act)
            torch.cuda.synchronize()
            n4 = (B + 3) // 4 * 4
            if K <= 32:   # one k tile in the tiled kernel: the same order of additions
                assert torch.equal(got[:, :n4], want[:, :n4]), (act, float((got[:, :B] - want[:, :B]).abs().max()))
            else:
                torch.testing.assert_close(got[:, :n4], want[:, :n4], rtol=2e-6, atol=2e-6)
            assert bool(torch.isnan(got[:, n4:]).all())
    if K + 1 <= 52:
        # round 4, what the rollout engine does: the bias as row K of the transposed weights against a row of ONES behind the
        # input's K rows, no bias argument (no bias loads in the kernel: nothing younger than the previous block's stores)
        Xa = torch.cat([X, torch.ones(1, ldb, device=dev)], dim=0)
        Wta = torch.zeros(K + 1, Wt.shape[1], device=dev)
        Wta[:K] = Wt
        Wta[K, :N] = bias
        want = torch.full((N, ldb), float("nan"), device=dev)
        got = torch.full((N, ldb), float("nan"), device=dev)
        ops.linear_fwd_thin_in(Wt[:, :N], bias, X, want, B, _lib.NIC_ACT_ELU)
        ops.linear_fwd_thin_in(Wta[:, :N], None, Xa, got, B, _lib.NIC_ACT_ELU)
        torch.cuda.synchronize()
        n4 = (B + 3) // 4 * 4
        torch.testing.assert_close(got[:, :n4], want[:, :n4], rtol=2e-6, atol=2e-6)
        assert bool(torch.isnan(got[:, n4:]).all())
    W64, X64 = W.double().cpu(), X.double().cpu()[:, :B]
    pre = W64 @ X64 + bias.double().cpu()[:, None]
    Y = torch.zeros(N, ldb, device=dev)
    ops.linear_fwd_thin_in(Wt[:, :N], bias, X, Y, B, _lib.NIC_ACT_ELU)
    torch.cuda.synchronize()
    _close(Y[:, :B], torch.where(pre > 0, pre, torch.expm1(pre)), W64.abs() @ X64.abs() + bias.double().cpu().abs()[:, None], "thin_in")
    assert _lib.lib().nic_last_kernel().decode() == "thin_in_fwd_kernel<26>"


@pytest.mark.parametrize("N,K,B", GEMM_SHAPES)
@pytest.mark.parametrize("padded_w", [False, True])
def test_linear_fwd(N, K, B, padded_w):
    dev = "cuda"
    gen = torch.Generator().manual_seed(N * 1000 + K)
    ldb = pad_ld(B)
    X = _rand((K, ldb), gen, dev)
    if padded_w:
        Wfull = torch.zeros(N, (K + 31) // 32 * 32, device=dev)
        Wfull[:, :K] = _rand((N, K), gen, dev, 0.3)
        W = Wfull[:, :K]
    else:
        W = _rand((N, K), gen, dev, 0.3)
    bias = _rand((N,), gen, dev)
    for act in (_lib.NIC_ACT_ELU, _lib.NIC_ACT_NONE):
        Y = torch.full((N, ldb), float("nan"), device=dev)
        ops.linear_fwd(W, bias, X, Y, B, act)
        torch.cuda.synchronize()
        W64, X64 = W.double().cpu(), X.double().cpu()[:, :B]
        pre = W64 @ X64 + bias.double().cpu()[:, None]
        want = torch.where(pre > 0, pre, torch.expm1(pre)) if act == _lib.NIC_ACT_ELU else pre
        scale = W64.abs() @ X64.abs() + bias.double().cpu().abs()[:, None]
        _close(Y[:, :B], want, scale, ("fwd", act))
    Y2 = torch.zeros(N, ldb, device=dev)
    ops.linear_fwd(W, None, X, Y2, B, _lib.NIC_ACT_NONE)
    torch.cuda.synchronize()
    _close(Y2[:, :B], W.double().cpu() @ X.double().cpu()[:, :B], W.double().cpu().abs() @ X.double().cpu()[:, :B].abs(), "nobias")


@pytest.mark.parametrize("N,K,B", GEMM_SHAPES)
def test_linear_dgrad(N, K, B):
    dev = "cuda"
    gen = torch.Generator().manual_seed(N * 1000 + K + 1)
    ldb = pad_ld(B)
    W = _rand((N, K), gen, dev, 0.3)
    Wt = W.t().contiguous()
    dY = _rand((N, ldb), gen, dev)
    H = _rand((K, ldb), gen, dev)  # "post-activation" of the previous layer (values <= 0 have derivative H + 1)
    H = torch.where(H > 0, H, torch.expm1(H))
    base = _rand((K, ldb), gen, dev)
    W64, dY64, H64 = W.double().cpu(), dY.double().cpu()[:, :B], H.double().cpu()[:, :B]
    lin = W64.t() @ dY64
    scale = W64.t().abs() @ dY64.abs()
    # plain
    dX = torch.full((K, ldb), float("nan"), device=dev)
    ops.linear_dgrad(Wt, dY, None, dX, B, _lib.NIC_ACT_NONE, False)
    torch.cuda.synchronize()
    _close(dX[:, :B], lin, scale, "dgrad plain")
    # ELU' from the stored output, accumulate into an existing gradient
    dX = base.clone()
    ops.linear_dgrad(Wt, dY, H, dX, B, _lib.NIC_ACT_ELU, True)
    torch.cuda.synchronize()
    want = lin * torch.where(H64 > 0, torch.ones_like(H64), H64 + 1) + base.double().cpu()[:, :B]
    _close(dX[:, :B], want, scale + base.double().cpu()[:, :B].abs(), "dgrad elu acc")


@pytest.mark.parametrize("N,K,B", GEMM_SHAPES)
def test_linear_wgrad_accumulates_over_periods(N, K, B):
    dev = "cuda"
    gen = torch.Generator().manual_seed(N * 1000 + K + 2)
    ldb = pad_ld(B)
    splits = ops.wgrad_num_splits(N, K, B)
    assert splits >= 1
    lds = (K + 1 + 3) // 4 * 4
    slab = torch.zeros(splits, N, lds, device=dev)
    want_w = torch.zeros(N, K, dtype=torch.float64)
    want_b = torch.zeros(N, dtype=torch.float64)
    scale_w = torch.zeros(N, K, dtype=torch.float64)
    for period in range(3):
        dY = _rand((N, ldb), gen, dev)  # padding columns are NOT zero: the kernel must ignore b >= n_scenarios
        X = _rand((K, ldb), gen, dev)
        ops.linear_wgrad(dY, X, slab, B)
        dY64, X64 = dY.double().cpu()[:, :B], X.double().cpu()[:, :B]
        want_w += dY64 @ X64.t()
        want_b += dY64.sum(dim=1)
        scale_w += dY64.abs() @ X64.abs().t()
    dW = torch.full((N, K), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.wgrad_reduce(slab, dW, db, K, 0.5)
    torch.cuda.synchronize()
    _close(dW, 0.5 * want_w, scale_w, "wgrad")
    _close(db, 0.5 * want_b, torch.full((N,), 3.0 * B), "bgrad")
    assert float(slab[:, :, K + 1:].abs().sum()) == 0.0


THIN_SHAPES = [(17, 512, 777), (17, 512, 4096), (5, 128, 130), (32, 256, 1000), (1, 32, 64), (18, 96, 2049), (9, 64, 63)]


@pytest.mark.parametrize("N,K,B", THIN_SHAPES)
@pytest.mark.parametrize("splits", [None, 3])
def test_linear_bwd_thin_matches_dgrad_plus_wgrad(N, K, B, splits):
    """nic_linear_bwd_thin = nic_linear_dgrad (ELU' of the layer below) + nic_linear_wgrad in one pass over X."""
    dev = "cuda"
    assert ops.linear_bwd_thin_ok(N, K)
    gen = torch.Generator().manual_seed(N * 1000 + K + 7)
    ldb = pad_ld(B)
    W = torch.zeros(N, K + 8, device=dev)[:, :K]  # padded leading dimension, as HipLinear stores its weights
    W.copy_(_rand((N, K), gen, dev, 0.3))
    n_splits = splits or ops.wgrad_num_splits(N, K, B)
    lds = (K + 1 + 3) // 4 * 4
    slab = torch.zeros(n_splits, N, lds, device=dev)
    want_w = torch.zeros(N, K, dtype=torch.float64)
    want_b = torch.zeros(N, dtype=torch.float64)
    scale_w = torch.zeros(N, K, dtype=torch.float64)
    W64 = W.double().cpu()
    for period in range(3):
        act = _lib.NIC_ACT_ELU if period != 1 else _lib.NIC_ACT_NONE
        dY = _rand((N, ldb), gen, dev)  # padding columns are NOT zero: the kernel must ignore b >= n_scenarios
        H = _rand((K, ldb), gen, dev)
        H = torch.where(H > 0, H, torch.expm1(H))
        dX = torch.full((K, ldb), float("nan"), device=dev)
        ops.linear_bwd_thin(W, dY, H, dX, slab, B, act)
        torch.cuda.synchronize()
        dY64, H64 = dY.double().cpu()[:, :B], H.double().cpu()[:, :B]
        lin = W64.t() @ dY64
        if act == _lib.NIC_ACT_ELU:
            lin = lin * torch.where(H64 > 0, torch.ones_like(H64), H64 + 1)
        _close(dX[:, :B], lin, W64.t().abs() @ dY64.abs(), f"thin dgrad period {period}")
        want_w += dY64 @ H64.t()
        want_b += dY64.sum(dim=1)
        scale_w += dY64.abs() @ H64.abs().t()
    dW = torch.full((N, K), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.wgrad_reduce(slab, dW, db, K, 0.5)
    torch.cuda.synchronize()
    _close(dW, 0.5 * want_w, scale_w, "thin wgrad")
    _close(db, 0.5 * want_b, torch.full((N,), 3.0 * B), "thin bgrad")
    assert float(slab[:, :, K + 1:].abs().sum()) == 0.0


def test_linear_bwd_thin_rejects_bad_shapes():
    dev = "cuda"
    z = torch.zeros(64, 64, device=dev)
    slab = torch.zeros(1, 64, 68, device=dev)
    with pytest.raises(_lib.NicError):
        ops.linear_bwd_thin(z[:40, :48], z[:40], z[:48], z[:48].clone(), slab, 64, _lib.NIC_ACT_ELU)   # N > 32
    with pytest.raises(_lib.NicError):
        ops.linear_bwd_thin(z[:8, :48], z[:8], z[:48], z[:48].clone(), slab, 64, _lib.NIC_ACT_ELU)     # K % 32 != 0


@pytest.mark.parametrize("N,K,B,T", [(512, 512, 2048, 3), (256, 200, 1024, 4), (300, 512, 1056, 2), (64, 33, 100, 3)])
def test_linear_wgrad_periods_equals_per_period_launches(N, K, B, T):
    """One launch contracting over T operand pairs == T nic_linear_wgrad launches (LDS-DMA shapes and fallback shapes)."""
    dev = "cuda"
    gen = torch.Generator().manual_seed(N + K + T)
    ldb = pad_ld(B)
    splits = ops.wgrad_num_splits(N, K, B)
    lds = (K + 1 + 3) // 4 * 4
    dY = _rand((T, N, ldb), gen, dev)
    Xfull = _rand((T + 1, K + 5, ldb), gen, dev)  # periods are slices of a larger history block, as in the rollout
    X = Xfull[:T, :K]
    want_w = torch.zeros(N, K, dtype=torch.float64)
    want_b = torch.zeros(N, dtype=torch.float64)
    scale_w = torch.zeros(N, K, dtype=torch.float64)
    for t in range(T):
        dY64, X64 = dY[t].double().cpu()[:, :B], X[t].double().cpu()[:, :B]
        want_w += dY64 @ X64.t()
        want_b += dY64.sum(dim=1)
        scale_w += dY64.abs() @ X64.abs().t()
    slab = torch.zeros(splits, N, lds, device=dev)
    ops.linear_wgrad_periods(dY, X, slab, B)
    ops.linear_wgrad_periods(dY, X, slab, B)  # accumulates
    dW = torch.full((N, K), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.wgrad_reduce(slab, dW, db, K, 0.5)
    torch.cuda.synchronize()
    _close(dW, want_w, 2 * scale_w, "wgrad periods")
    _close(db, want_b, torch.full((N,), 2.0 * T * B), "bgrad periods")


@pytest.mark.parametrize("N,K,B,T,slots", [(512, 512, 1024, 7, None), (512, 512, 256, 9, 64), (512, 51, 1024, 6, None),
                                           (512, 51, 512, 5, 7), (98, 512, 512, 5, None), (512, 393, 384, 4, 6),
                                           (512, 512, 8192, 3, None), (512, 66, 1024, 6, None), (320, 150, 512, 5, None),
                                           # register-staged kernels (ragged scenario counts / narrow layers: the real-data
                                           # batches of 72 products x 95 weeks with the 64-wide data_driven net)
                                           (64, 597, 72, 19, None), (66, 64, 72, 10, None), (64, 64, 100, 7, 5), (200, 100, 300, 6, None)])
def test_linear_wgrad_periods_splits_the_horizon_into_period_groups(N, K, B, T, slots):
    """Round 4: with few scenarios the slab slots of the all-period contraction are (period group x scenario split) pairs
    (nic_wgrad_periods_num_splits), so that a batch of 1,024 still gives every CU a workgroup.  Any slot count is accepted:
    `None` = the count the library asks for, otherwise an arbitrary one (slots that are not used stay untouched)."""
    dev = "cuda"
    gen = torch.Generator().manual_seed(N + K + T + B)
    ldb = pad_ld(B)
    want_slots = ops.wgrad_periods_num_splits(N, K, B, T)
    assert want_slots >= ops.wgrad_num_splits(N, K, B) or B >= 8192
    if B <= 1024:
        assert want_slots > max(1, B // 128), "few scenarios: the horizon must be split as well"
    n_slots = slots or want_slots
    lds = (K + 1 + 3) // 4 * 4
    dY = _rand((T, N, ldb), gen, dev)
    X = _rand((T, K, ldb), gen, dev)
    want_w = torch.zeros(N, K, dtype=torch.float64)
    want_b = torch.zeros(N, dtype=torch.float64)
    scale_w = torch.zeros(N, K, dtype=torch.float64)
    for t in range(T):
        dY64, X64 = dY[t].double().cpu()[:, :B], X[t].double().cpu()[:, :B]
        want_w += dY64 @ X64.t()
        want_b += dY64.sum(dim=1)
        scale_w += dY64.abs() @ X64.abs().t()
    slab = torch.zeros(n_slots, N, lds, device=dev)
    ops.linear_wgrad_periods(dY, X, slab, B)
    dma = B % 32 == 0 and ((N >= 192 and K >= 65) or (N >= 384 and K <= 64) or (96 <= N <= 128 and K >= 192))   # wgrad_dma_shape
    assert _lib.lib().nic_last_kernel().decode().startswith("gemm_wgrad_dma_kernel" if dma else "gemm_wgrad_kernel")
    dW = torch.full((N, K), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.wgrad_reduce(slab, dW, db, K, 1.0)
    torch.cuda.synchronize()
    _close(dW, want_w, scale_w, "wgrad period groups")
    _close(db, want_b, torch.full((N,), 1.0 * T * B), "bgrad period groups")
    used = int((slab.abs().sum(dim=(1, 2)) > 0).sum())
    assert 1 <= used <= n_slots


def test_every_wx_tile_of_the_picker_is_exercised_and_correct():
    """Round 4: `pick_wx_tile` chooses among three tilings of the LDS-DMA kernel (co-resident workgroups per CU, padding to the
    block height).  A sweep over output rows x scenario counts - from the reference's shipped batch size to BASELINE cfg3's - checks
    forward and input gradient against float64 and records which kernel ran: every tiling must come up at least once."""
    dev = "cuda"
    seen = set()
    sweep = [(512, 512, b) for b in (512, 1024, 2048, 4096, 8192, 16384, 32768)] + \
            [(17, 512, b) for b in (1024, 4096, 16384, 65536)] + [(64, 512, b) for b in (1024, 8192, 65536)] + \
            [(128, 512, b) for b in (1024, 4096, 32768)] + [(393, 512, b) for b in (2048, 32768)] + \
            [(256, 256, b) for b in (1024, 4096, 16384, 65536)] + [(195, 512, 32768), (512, 512, 65536)]
    for N, K, B in sweep:
        gen = torch.Generator().manual_seed(N * 7 + K + B)
        ldb = pad_ld(B)
        W = torch.zeros(N, (K + 31) // 32 * 32, device=dev)
        W[:, :K] = _rand((N, K), gen, dev, 0.3)
        Wt = torch.zeros(K, (N + 31) // 32 * 32, device=dev)
        Wt[:, :N] = W[:, :K].t()
        bias = _rand((N,), gen, dev)
        X = _rand((K, ldb), gen, dev)
        Y = torch.full((N, ldb), float("nan"), device=dev)
        ops.linear_fwd(W[:, :K], bias, X, Y, B, _lib.NIC_ACT_ELU)
        seen.add(_lib.lib().nic_last_kernel().decode())
        # a sample of columns against float64 (the whole product at 65,536 columns would take the CPU a minute)
        cols = torch.unique(torch.cat([torch.arange(0, min(B, 300)), torch.randint(0, B, (300,), generator=gen),
                                       torch.arange(max(0, B - 300), B)]))
        W64, X64 = W[:, :K].double().cpu(), X.double().cpu()[:, cols]
        pre = W64 @ X64 + bias.double().cpu()[:, None]
        _close(Y[:, cols.to(dev)], torch.where(pre > 0, pre, torch.expm1(pre)), W64.abs() @ X64.abs() + bias.double().cpu().abs()[:, None],
               ("fwd", N, K, B))
        H = torch.where(X > 0, X, torch.expm1(X))
        dX = torch.full((K, ldb), float("nan"), device=dev)
        ops.linear_dgrad(Wt[:, :N], Y, H, dX, B, _lib.NIC_ACT_ELU, False)
        seen.add(_lib.lib().nic_last_kernel().decode())
        Y64, H64 = Y.double().cpu()[:, cols], H.double().cpu()[:, cols]
        want = (W64.t() @ Y64) * torch.where(H64 > 0, torch.ones_like(H64), H64 + 1)
        _close(dX[:, cols.to(dev)], want, W64.t().abs() @ Y64.abs(), ("dgrad", N, K, B))
        assert bool(torch.isnan(dX[:, (B + 3) // 4 * 4:]).all()) and bool(torch.isnan(Y[:, (B + 3) // 4 * 4:]).all())
    tiles = {k.split("<")[1].rsplit(",", 1)[0] for k in seen if k.startswith("gemm_wx_dma_kernel<")}
    every = {"2,4,2,1", "2,4,1,1", "1,4,1,1"}   # 128 x 128 and 64 x 128 (8 waves), 32 x 128 (4 waves)
    assert tiles == every, (sorted(tiles), sorted(seen))


# ---- sampler ------------------------------------------------------------------------------------------------------

def test_sampler_normal_moments_and_sharding_invariance():
    dev = "cuda"
    S, T, B = 5, 40, 4096
    mean = torch.tensor([5.0, 3.0, 7.0, 4.0, 6.0])
    std = torch.tensor([1.5, 1.0, 2.0, 0.8, 1.2])
    rho = 0.5
    cov = rho * std[:, None] * std[None, :]
    cov[range(S), range(S)] = std * std
    chol = torch.linalg.cholesky(cov.double()).float()
    ldb = pad_ld(B)
    out = torch.zeros(T, S, ldb, device=dev)
    ops.sample_demand(out, T, S, B, 0, 1234, 0, mean.to(dev), chol.to(dev).contiguous(), False)
    torch.cuda.synchronize()
    x = out[:, :, :B].permute(1, 0, 2).reshape(S, -1).double().cpu()  # S x (T*B)
    np.testing.assert_allclose(x.mean(dim=1).numpy(), mean.numpy(), atol=0.02)
    np.testing.assert_allclose(np.cov(x.numpy()), cov.numpy(), atol=0.05)
    # independent across periods and scenarios
    a, b = out[0, 0, :B].double().cpu(), out[1, 0, :B].double().cpu()
    assert abs(np.corrcoef(a.numpy(), b.numpy())[0, 1]) < 0.06
    # sharding invariance: two shards with global scenario offsets reproduce the single-GPU traces bit for bit
    half = B // 2
    o1 = torch.zeros(T, S, pad_ld(half), device=dev)
    o2 = torch.zeros(T, S, pad_ld(half), device=dev)
    ops.sample_demand(o1, T, S, half, 0, 1234, 0, mean.to(dev), chol.to(dev).contiguous(), False)
    ops.sample_demand(o2, T, S, half, half, 1234, 0, mean.to(dev), chol.to(dev).contiguous(), False)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o1[:, :, :half], o2[:, :, :half]], dim=2), out[:, :, :B])
    # clip
    oc = torch.zeros(T, S, ldb, device=dev)
    ops.sample_demand(oc, T, S, B, 0, 1234, 0, (mean * 0).to(dev), chol.to(dev).contiguous(), True)
    torch.cuda.synchronize()
    assert float(oc.min()) == 0.0


@pytest.mark.parametrize("S", [1, 5, 16, 64])
def test_sampler_equicorrelated_moments_covariance_and_sharding(S):
    """The reference's covariance (rho s_i s_j off the diagonal, data_handling.py:194-201) through the one-factor sampler:
    per-store mean / variance, every pairwise covariance, independence across periods, bit-exact sharding invariance."""
    dev = "cuda"
    T, B = 24, 8192
    gen = torch.Generator().manual_seed(S)
    mean = 2.5 + 5.0 * torch.rand(S, generator=gen)
    std = mean * (0.25 + 0.25 * torch.rand(S, generator=gen))
    rho = 0.5 if S > 1 else 0.0
    cov = rho * std[:, None] * std[None, :]
    cov[range(S), range(S)] = std * std
    ldb = pad_ld(B)
    out = torch.zeros(T, S, ldb, device=dev)
    ops.sample_demand_equicorrelated(out, T, S, B, 0, 4321, mean.to(dev), std.to(dev), rho, False)
    torch.cuda.synchronize()
    assert float(out[:, :, B:].abs().sum()) == 0.0
    x = out[:, :, :B].permute(1, 0, 2).reshape(S, -1).double().cpu()  # S x (T*B) = 196,608 draws per store
    np.testing.assert_allclose(x.mean(dim=1).numpy(), mean.numpy(), atol=0.03)
    got = np.atleast_2d(np.cov(x.numpy()))
    np.testing.assert_allclose(got, cov.numpy(), atol=0.06 * float(cov.max()))
    # normality of the marginals: skewness ~ 0, kurtosis ~ 3
    zs = (x - x.mean(dim=1, keepdim=True)) / x.std(dim=1, keepdim=True)
    assert float((zs ** 3).mean(dim=1).abs().max()) < 0.03 and float(((zs ** 4).mean(dim=1) - 3).abs().max()) < 0.08
    a, b = out[0, 0, :B].double().cpu(), out[1, 0, :B].double().cpu()
    assert abs(np.corrcoef(a.numpy(), b.numpy())[0, 1]) < 0.05
    half = B // 2
    o1, o2 = torch.zeros(T, S, pad_ld(half), device=dev), torch.zeros(T, S, pad_ld(half), device=dev)
    ops.sample_demand_equicorrelated(o1, T, S, half, 0, 4321, mean.to(dev), std.to(dev), rho, False)
    ops.sample_demand_equicorrelated(o2, T, S, half, half, 4321, mean.to(dev), std.to(dev), rho, False)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o1[:, :, :half], o2[:, :, :half]], dim=2), out[:, :, :B])
    oc = torch.zeros(T, S, ldb, device=dev)
    ops.sample_demand_equicorrelated(oc, T, S, B, 0, 4321, (mean * 0).to(dev), std.to(dev), rho, True)
    torch.cuda.synchronize()
    assert float(oc.min()) == 0.0


def test_sampler_equicorrelated_is_launch_geometry_independent():
    """Above ~4M (scenario, period) pairs the sampler gives each lane four periods, below it one: a big run and its two
    half-size shards (which take the other form) must agree bit for bit, ragged horizon included."""
    dev = "cuda"
    S, T, B = 16, 18, 262144
    mean, std = torch.full((S,), 5.0, device=dev), torch.linspace(1.0, 2.0, S).to(dev)
    full = torch.zeros(T, S, B, device=dev)
    ops.sample_demand_equicorrelated(full, T, S, B, 0, 9, mean, std, 0.5, True)
    assert _lib.lib().nic_last_kernel() == b"sample_equicorrelated_kernel<4>"
    half = B // 2
    parts = []
    for off in (0, half):
        o = torch.zeros(T, S, half, device=dev)
        ops.sample_demand_equicorrelated(o, T, S, half, off, 9, mean, std, 0.5, True)
        assert _lib.lib().nic_last_kernel() == b"sample_equicorrelated_kernel<1>"
        parts.append(o)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, dim=2), full)


def test_sampler_one_store_form_is_sharding_invariant_at_ragged_sizes():
    """Round 6: with one store a lane owns four consecutive scenarios x four consecutive periods (all four words of a Philox block
    used, 16-byte stores).  Shards that start at scenario indices which are not multiples of four, batch sizes that are not
    multiples of four and a horizon that is not a multiple of four must reproduce the single run bit for bit; padding columns stay
    zero; moments of N(mean, std^2) with clipping at zero as the oracle's generator has them."""
    dev = "cuda"
    T, B = 7, 2001
    mean, std = torch.tensor([5.0], device=dev), torch.tensor([1.6], device=dev)
    full = torch.zeros(T, 1, pad_ld(B), device=dev)
    ops.sample_demand_equicorrelated(full, T, 1, B, 0, 77, mean, std, 0.0, True)
    assert _lib.lib().nic_last_kernel() == b"sample_one_store_kernel"
    assert float(full[:, :, B:].abs().sum()) == 0.0
    parts, off = [], 0
    for n in (1001, 3, 997):
        o = torch.zeros(T, 1, pad_ld(n), device=dev)
        ops.sample_demand_equicorrelated(o, T, 1, n, off, 77, mean, std, 0.0, True)
        assert float(o[:, :, n:].abs().sum()) == 0.0
        parts.append(o[:, :, :n])
        off += n
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, dim=2), full[:, :, :B])
    big = torch.zeros(40, 1, pad_ld(65536), device=dev)
    ops.sample_demand_equicorrelated(big, 40, 1, 65536, 0, 78, mean, std, 0.5, False)   # (rho has nothing to act on with one store)
    x = big[:, 0, :65536].double().cpu()
    assert abs(float(x.mean()) - 5.0) < 0.01 and abs(float(x.std()) - 1.6) < 0.01
    z = (x - x.mean()) / x.std()
    assert abs(float((z ** 3).mean())) < 0.02 and abs(float((z ** 4).mean()) - 3.0) < 0.05
    # independence across periods and across neighbouring scenarios (the words of one block / of neighbouring blocks)
    assert abs(np.corrcoef(x[0].numpy(), x[1].numpy())[0, 1]) < 0.02 and abs(np.corrcoef(x[4].numpy(), x[3].numpy())[0, 1]) < 0.02
    assert abs(np.corrcoef(x[:, :-1].reshape(-1).numpy(), x[:, 1:].reshape(-1).numpy())[0, 1]) < 0.01


@pytest.mark.parametrize("S", [3, 16, 40, 64])
def test_sampler_general_covariance(S):
    """Arbitrary (here: negatively correlated blocks) covariance through the Cholesky sampler: z kept in registers."""
    dev = "cuda"
    T, B = 16, 8192
    gen = torch.Generator().manual_seed(100 + S)
    A = torch.randn(S, S, generator=gen, dtype=torch.float64) * 0.4
    cov = A @ A.t() + torch.eye(S, dtype=torch.float64)
    chol = torch.linalg.cholesky(cov).float()
    mean = torch.linspace(3.0, 9.0, S)
    out = torch.zeros(T, S, pad_ld(B), device=dev)
    ops.sample_demand(out, T, S, B, 0, 77, 0, mean.to(dev), chol.to(dev).contiguous(), False)
    torch.cuda.synchronize()
    x = out[:, :, :B].permute(1, 0, 2).reshape(S, -1).double().cpu()
    np.testing.assert_allclose(x.mean(dim=1).numpy(), mean.numpy(), atol=0.05)
    np.testing.assert_allclose(np.cov(x.numpy()), cov.numpy(), atol=0.05 * float(cov.max()))


def test_sampler_poisson():
    dev = "cuda"
    S, T, B = 2, 50, 8192
    lam = torch.tensor([5.0, 0.7])
    out = torch.zeros(T, S, pad_ld(B), device=dev)
    ops.sample_demand(out, T, S, B, 0, 99, 1, lam.to(dev), None, True)
    torch.cuda.synchronize()
    x = out[:, :, :B].permute(1, 0, 2).reshape(S, -1).double().cpu()
    assert torch.equal(x, x.round()) and float(x.min()) >= 0
    np.testing.assert_allclose(x.mean(dim=1).numpy(), lam.numpy(), rtol=0.01)
    np.testing.assert_allclose(x.var(dim=1).numpy(), lam.numpy(), rtol=0.03)
    # P(X = 0) = exp(-lambda)
    np.testing.assert_allclose((x == 0).double().mean(dim=1).numpy(), np.exp(-lam.numpy()), atol=0.003)
    # a mean whose mass reaches past the 64-entry CDF table: the search continues the recurrence
    big = torch.tensor([45.0])
    ob = torch.zeros(8, 1, pad_ld(B), device=dev)
    ops.sample_demand(ob, 8, 1, B, 0, 5, 1, big.to(dev), None, True)
    torch.cuda.synchronize()
    xb = ob[:, :, :B].double().cpu().reshape(-1)
    assert abs(float(xb.mean()) - 45.0) < 0.15 and abs(float(xb.var()) - 45.0) < 1.5 and float(xb.max()) > 63


# ---- fused 3-layer MLP over gathered inputs (csrc/mlp3.hip) ----------------------------------------------------------------

def _mlp3_reference(x, W, act):
    """x: (cols, K) float64; W: [(w1, b1), (w2, b2), (w3, b3)] float64 with requires_grad."""
    import torch.nn.functional as F
    h1 = F.elu(F.linear(x, *W[0]))
    h2 = F.elu(F.linear(h1, *W[1]))
    z = F.linear(h2, *W[2])
    y = F.elu(z) if act == 1 else (F.softplus(z) if act == 2 else z)
    return y, h1, h2


@pytest.mark.parametrize("K_rows,n_out,act,B", [((32, 32, 1), 32, 1, 100), ((32, 32, 32), 32, 1, 64), ((7,), 32, 1, 37),
                                                 ((32,), 1, 2, 70), ((20, 13), 5, 0, 33)])
def test_mlp3_fwd_bwd_with_gathered_segments(K_rows, n_out, act, B):
    """nic_mlp3_fwd / nic_mlp3_bwd_fused / nic_mlp3_bwd_hist against float64 torch autograd: inputs gathered through entity maps (with -1 = zero rows and
    a per-entity constant segment), every history buffer, dX, and the pre-activation gradients the weight gradients use."""
    dev = "cuda"
    gen = torch.Generator().manual_seed(sum(K_rows) + n_out)
    E, ld = 6, pad_ld(B, 32)
    n_src = 4
    K = sum(K_rows)
    segs, dense_parts = [], []
    for si, rows in enumerate(K_rows):
        if rows == 1:  # per-entity constant (an edge's lead time)
            t = torch.randn(1, E, generator=gen)
            segs.append(ops.Mlp3Segment(t.to(dev), None, per_scenario=False))
            dense_parts.append(t[:, :, None].expand(1, E, B))
        else:
            t = torch.zeros(rows, n_src, ld)
            t[:, :, :B] = torch.randn(rows, n_src, B, generator=gen)
            idx = torch.randint(-1, n_src, (E,), generator=gen, dtype=torch.int32)
            if si == 0:
                idx = None if len(K_rows) == 1 and rows != 20 else idx
            if idx is None:  # identity map needs as many source entities as outputs
                t = torch.zeros(rows, E, ld)
                t[:, :, :B] = torch.randn(rows, E, B, generator=gen)
                dense = t[:, :, :B]
            else:
                dense = torch.where((idx >= 0)[None, :, None], t[:, idx.clamp_min(0).long(), :B], torch.zeros(()))
            segs.append(ops.Mlp3Segment(t.to(dev), idx.to(dev) if idx is not None else None))
            dense_parts.append(dense)
    X = torch.cat(dense_parts, dim=0)  # [K][E][B]
    dims = [(32, K), (32, 32), (n_out, 32)]
    W = [(torch.randn(n, k, generator=gen) / k ** 0.5, torch.randn(n, generator=gen) * 0.3) for n, k in dims]
    packed = torch.cat([t.reshape(-1) for wb in W for t in wb]).to(dev)
    packed_t = ops.mlp3_pack_transposed([(w.to(dev), b_.to(dev)) for w, b_ in W], n_out)
    desc = ops.mlp3_desc(segs, packed, E, B, ld, n_out, act, 0, packed_t)
    z = lambda r: torch.zeros(r, E, ld, device=dev)  # noqa: E731
    Y, Xh, H1, H2 = z(n_out), z(K), z(32), z(32)
    ops.mlp3_fwd(desc, Y, Xh, H1, H2)
    torch.cuda.synchronize()
    W64 = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in W]
    x64 = X.permute(1, 2, 0).reshape(E * B, K).double().requires_grad_(True)
    y, h1, h2 = _mlp3_reference(x64, W64, act)
    as_rows = lambda t, r: t.reshape(E, B, r).permute(2, 0, 1)  # noqa: E731
    assert torch.equal(Xh[:, :, :B].cpu(), X)
    torch.testing.assert_close(H1[:, :, :B].cpu().double(), as_rows(h1.detach(), 32), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(H2[:, :, :B].cpu().double(), as_rows(h2.detach(), 32), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(Y[:, :, :B].cpu().double(), as_rows(y.detach(), n_out), rtol=2e-5, atol=2e-6)
    assert float(Y[:, :, B:].abs().sum()) == 0.0
    # backward
    gY = torch.zeros(n_out, E, ld)
    gY[:, :, :B] = torch.randn(n_out, E, B, generator=gen)
    (y * gY[:, :, :B].permute(1, 2, 0).reshape(E * B, n_out).double()).sum().backward()
    # (the round-2 backward that stored the pre-activation gradients for separate contractions was removed in round 6)
    # the history-free backward (re-gather, recompute, in-kernel weight gradients): same dX, same dW / db after the slab reduce;
    # slabs accumulate over launches (two launches = twice the gradient)
    slots = ops.mlp3_bwd_fused_slots()
    slabs = [torch.zeros(slots, n, (k + 1 + 3) // 4 * 4, device=dev) for n, k in dims]
    dX2 = z(K)
    for _ in range(2):
        ops.mlp3_bwd_fused(desc, gY.to(dev), Y, dX2, slabs)
    torch.cuda.synchronize()
    torch.testing.assert_close(dX2[:, :, :B].cpu().double(), as_rows(x64.grad, K), rtol=1e-4, atol=1e-5)
    for sl, (n, k), (w, b_) in zip(slabs, dims, W64):
        gw, gb = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
        ops.wgrad_reduce(sl, gw, gb, k, 0.5)
        torch.cuda.synchronize()
        torch.testing.assert_close(gw.cpu().double(), w.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(gb.cpu().double(), b_.grad, rtol=1e-4, atol=1e-5)


    # the backward over the stored activations with in-kernel weight gradients (nic_mlp3_bwd_hist): same again, one slot per workgroup
    # - with the stored inputs (X history) and with the inputs read again from the segments' sources (no X history)
    for x_hist in (Xh, None):
        slabs = [torch.zeros(ops.mlp3_bwd_hist_slots(), n, (k + 1 + 3) // 4 * 4, device=dev) for n, k in dims]
        dX3 = z(K)
        for _ in range(2):
            ops.mlp3_bwd_hist(desc, gY.to(dev), Y, x_hist, H1, H2, dX3, slabs)
        torch.cuda.synchronize()
        torch.testing.assert_close(dX3[:, :, :B].cpu().double(), as_rows(x64.grad, K), rtol=1e-4, atol=1e-5)
        for sl, (n, k), (w, b_) in zip(slabs, dims, W64):
            gw, gb = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
            ops.wgrad_reduce(sl, gw, gb, k, 0.5)
            torch.cuda.synchronize()
            torch.testing.assert_close(gw.cpu().double(), w.grad, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(gb.cpu().double(), b_.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("residual", [False, True])
def test_mlp3_fwd_two_chunks_per_wavefront_equals_one(residual):
    """Launches that would not fill the chip's wavefront slots once take the variant in which a wavefront owns two adjacent
    32-scenario chunks (mlp3_fwd_kernel<KS,2>).  Same arithmetic per column: bit-equal to the one-chunk variant (which a launch
    over the first 64 scenarios takes), float64 reference within the usual band, an odd chunk count (dead second chunk) and a
    ragged last chunk included."""
    from neural_inventory_control_amd import _lib
    dev = "cuda"
    gen = torch.Generator().manual_seed(77)
    E, B, n_src, n_out = 17, 4096 + 20, 9, 32
    ld = pad_ld(B, 32)
    assert (ld // 32) % 2 == 1
    rows = (32, 32, 32)
    K = sum(rows)
    segs, dense = [], []
    for si, r in enumerate(rows):
        t = torch.zeros(r, n_src if si else E, ld)
        t[:, :, :B] = torch.randn(r, t.shape[1], B, generator=gen)
        idx = torch.randint(-1, n_src, (E,), generator=gen, dtype=torch.int32) if si else None
        dense.append(t[:, :, :B] if idx is None else
                     torch.where((idx >= 0)[None, :, None], t[:, idx.clamp_min(0).long(), :B], torch.zeros(())))
        segs.append(ops.Mlp3Segment(t.to(dev), idx.to(dev) if idx is not None else None))
    X = torch.cat(dense, dim=0)
    dims = [(32, K), (32, 32), (n_out, 32)]
    W = [(torch.randn(n, k, generator=gen) / k ** 0.5, torch.randn(n, generator=gen) * 0.3) for n, k in dims]
    packed = torch.cat([t.reshape(-1) for wb in W for t in wb]).to(dev)
    packed_t = ops.mlp3_pack_transposed([(w.to(dev), b_.to(dev)) for w, b_ in W], n_out)
    z = lambda r, l=ld: torch.zeros(r, E, l, device=dev)  # noqa: E731
    R = segs[0].tensor if residual else None

    def run(n_cols):
        desc = ops.mlp3_desc(segs, packed, E, n_cols, ld, n_out, 1, 0, packed_t)
        Y, Xh, H1, H2, Ys = z(n_out), z(K), z(32), z(32), (z(n_out) if residual else None)
        ops.mlp3_fwd(desc, Y, Xh, H1, H2, R, Ys)
        torch.cuda.synchronize()
        return Y, Xh, H1, H2, Ys, _lib.lib().nic_last_kernel().decode()

    big, small = run(B), run(64)
    assert big[5] == "mlp3_fwd_kernel<48,2,2>" and small[5] == "mlp3_fwd_kernel<48,1,2>", (big[5], small[5])
    for a, b_ in zip(big[:5], small[:5]):
        if a is not None:
            assert torch.equal(a[:, :, :64], b_[:, :, :64])
            assert float(a[:, :, B:].abs().sum()) == 0.0
    x64 = X.permute(1, 2, 0).reshape(E * B, K).double()
    y, h1, h2 = _mlp3_reference(x64, [(w.double(), b_.double()) for w, b_ in W], 1)
    as_rows = lambda t, r: t.reshape(E, B, r).permute(2, 0, 1)  # noqa: E731
    assert torch.equal(big[1][:, :, :B].cpu(), X)
    torch.testing.assert_close(big[2][:, :, :B].cpu().double(), as_rows(h1, 32), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(big[3][:, :, :B].cpu().double(), as_rows(h2, 32), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(big[0][:, :, :B].cpu().double(), as_rows(y, n_out), rtol=2e-5, atol=2e-6)
    if residual:
        assert torch.equal(big[4], big[0] + R)


def test_mlp3_fwd_with_rows_too_far_apart_for_32_bit_offsets():
    """History rows 64 MiB apart (hist_row_stride = 2^24 floats: 32 rows span 2 GiB) take the forward's 64-bit-address variant
    (mlp3_fwd_kernel<..,0>); same bits as the buffer-addressed variant on compact buffers."""
    from neural_inventory_control_amd import _lib
    dev = "cuda"
    gen = torch.Generator().manual_seed(3)
    E, B, K, n_out = 3, 70, 7, 32
    ld = pad_ld(B, 32)
    t = torch.zeros(K, E, ld)
    t[:, :, :B] = torch.randn(K, E, B, generator=gen)
    segs = [ops.Mlp3Segment(t.to(dev), None)]
    dims = [(32, K), (32, 32), (n_out, 32)]
    W = [(torch.randn(n, k, generator=gen) / k ** 0.5, torch.randn(n, generator=gen) * 0.3) for n, k in dims]
    packed = torch.cat([x.reshape(-1) for wb in W for x in wb]).to(dev)
    packed_t = ops.mlp3_pack_transposed([(w.to(dev), b_.to(dev)) for w, b_ in W], n_out)
    stride = 1 << 24
    out = {}
    for name, hs in (("compact", 0), ("far", stride)):
        desc = ops.mlp3_desc(segs, packed, E, B, ld, n_out, 1, hs, packed_t)
        rows = lambda r: (torch.zeros(r, E, ld, device=dev) if hs == 0 else torch.zeros(r, stride, device=dev))  # noqa: E731
        Y, Xh, H1, H2 = torch.zeros(n_out, E, ld, device=dev), rows(K), rows(32), rows(32)
        ops.mlp3_fwd(desc, Y, Xh, H1, H2)
        torch.cuda.synchronize()
        kern = _lib.lib().nic_last_kernel().decode()
        view = lambda x: x.reshape(x.shape[0], -1)[:, :E * ld].reshape(-1, E, ld).clone()  # noqa: E731
        out[name] = (Y.clone(), view(Xh), view(H1), view(H2), kern)
        del Xh, H1, H2
    assert out["compact"][4] == "mlp3_fwd_kernel<4,1,2>" and out["far"][4] == "mlp3_fwd_kernel<4,1,0>", (out["compact"][4], out["far"][4])
    for a, b_ in zip(out["compact"][:4], out["far"][:4]):
        assert torch.equal(a, b_)
    assert float(out["far"][2].abs().sum()) > 0


def test_segment_sum_forward_aggregation_and_gather_adjoint():
    dev = "cuda"
    gen = torch.Generator().manual_seed(11)
    R, n_src, n_dst, B = 32, 9, 4, 130
    ld = pad_ld(B, 32)
    src = torch.zeros(R, n_src, ld)
    src[:, :, :B] = torch.randn(R, n_src, B, generator=gen)
    lists = [[0, 3, 8], [], [2], [1, 4, 5, 6, 7]]
    offsets = torch.tensor([0, 3, 3, 4, 9], dtype=torch.int32)
    items = torch.tensor([i for l in lists for i in l], dtype=torch.int32)
    scale = torch.tensor([0.5, 1.0, 2.0, 1 / 5 ** 0.5])
    dst = torch.full((R, n_dst, ld), 7.0, device=dev)
    ops.segment_sum(dst, src.to(dev), offsets.to(dev), items.to(dev), scale.to(dev), accumulate=False)
    want = torch.stack([scale[n] * sum((src[:, i] for i in l), torch.zeros(R, ld)) for n, l in enumerate(lists)], dim=1)
    torch.testing.assert_close(dst.cpu()[:, :, :B], want[:, :, :B], rtol=1e-6, atol=1e-6)
    ops.segment_sum(dst, src.to(dev), offsets.to(dev), items.to(dev), None, accumulate=True)
    want2 = want + torch.stack([sum((src[:, i] for i in l), torch.zeros(R, ld)) for l in lists], dim=1)
    torch.testing.assert_close(dst.cpu()[:, :, :B], want2[:, :, :B], rtol=1e-6, atol=1e-6)


def test_segment_sum_terms_equals_the_chain_of_launches():
    """nic_segment_sum_terms (several segment sums and plain addends into one destination, one launch) is bit-equal to the chain of
    nic_segment_sum(accumulate) launches and tensor adds it replaces."""
    dev = "cuda"
    gen = torch.Generator().manual_seed(5)
    R, n_src, n_dst, B = 32, 9, 4, 130
    ld = pad_ld(B, 32)
    mk = lambda n: torch.nn.functional.pad(torch.randn(R, n, B, generator=gen), (0, ld - B)).to(dev)  # noqa: E731
    a, b_, c, d_ = mk(n_dst), mk(n_dst), mk(n_src), mk(n_src)
    off1 = torch.tensor([0, 3, 3, 4, 9], dtype=torch.int32, device=dev)
    it1 = torch.tensor([0, 3, 8, 2, 1, 4, 5, 6, 7], dtype=torch.int32, device=dev)
    off2 = torch.tensor([0, 1, 2, 3, 4], dtype=torch.int32, device=dev)
    it2 = torch.tensor([8, 0, 3, 3], dtype=torch.int32, device=dev)
    sc = torch.tensor([0.5, 1.0, 2.0, 1 / 5 ** 0.5], device=dev)
    want = torch.add(a, b_)
    ops.segment_sum(want, c, off1, it1, sc, accumulate=True)
    ops.segment_sum(want, d_, off2, it2, None, accumulate=True)
    got = torch.full_like(want, 3.0)
    ops.segment_sum_terms(got, [(a, None, None, None), (b_, None, None, None), (c, off1, it1, sc), (d_, off2, it2, None)])
    assert torch.equal(got, want)
    want2 = want.clone()
    ops.segment_sum(want2, c, off1, it1, None, accumulate=True)
    ops.segment_sum_terms(got, [(c, off1, it1, None)], accumulate=True)
    assert torch.equal(got, want2)
    first = torch.zeros_like(want)
    ops.segment_sum(first, c, off1, it1, sc)
    ops.segment_sum(first, d_, off2, it2, None, accumulate=True)
    ops.segment_sum_terms(got, [(c, off1, it1, sc), (d_, off2, it2, None)])
    assert torch.equal(got, first)


@pytest.mark.parametrize("S,Wn", [(21, 3), (5, 2), (1, 0), (4, 0)])
def test_data_driven_head_matches_torch_autograd(S, Wn):
    """nic_head_data_driven_fwd / bwd against `DataDrivenNet.forward`'s tail in tensor ops (neural_networks.py:474-515, :111-138):
    ReLU, adjacency mask, proportional allocation of each warehouse's PIPELINE TOTAL over its stores; float64 autograd for
    dZ and the gradient of every pipeline slot.  Wn = 0: the one-store settings (orders = relu(Z))."""
    dev, B, Ww = "cuda", 205, 4
    ld = pad_ld(B, 32)
    gen = torch.Generator().manual_seed(S * 7 + Wn)
    rows = Wn + S * max(Wn, 1) if Wn else S
    Z = torch.zeros(rows, ld)
    Z[:, :B] = torch.randn(rows, B, generator=gen) * 2
    mask = (torch.rand(S, max(Wn, 1), generator=gen) < 0.7).float()
    if Wn:
        mask[0] = 1.0
        mask[:, -1] = 0.0 if S > 6 else mask[:, -1]          # (a warehouse that serves nobody: its column stays 0)
    wh = torch.zeros(max(Wn, 1), Ww, ld)
    wh[:, :, :B] = torch.rand(max(Wn, 1), Ww, B, generator=gen) * (S / 3)
    n_so = S * max(Wn, 1)
    g_so, g_wo = torch.zeros(n_so, ld), torch.zeros(max(Wn, 1), ld)
    g_so[:, :B] = torch.randn(n_so, B, generator=gen)
    g_wo[:, :B] = torch.randn(max(Wn, 1), B, generator=gen)
    z64 = Z[:, :B].double().requires_grad_(True)
    w64 = wh[:, :, :B].double().requires_grad_(True)
    out = torch.relu(z64)
    if Wn:
        wo_ref = out[:Wn]
        alloc = out[Wn:].reshape(S, Wn, B) * mask.double()[:, :, None]
        scale = torch.clamp(w64.sum(dim=1) / (alloc.sum(dim=0) + 1e-10), max=1.0)      # [Wn][B]
        so_ref = (alloc * scale[None]).reshape(S * Wn, B)
        ((so_ref * g_so[:, :B].double()).sum() + (wo_ref * g_wo[:Wn, :B].double()).sum()).backward()
    else:
        so_ref = out
        (so_ref * g_so[:, :B].double()).sum().backward()
    d = lambda t: t.to(dev)  # noqa: E731
    so, wo = torch.zeros(n_so, ld, device=dev), torch.zeros(max(Wn, 1), ld, device=dev)
    Zd, whd, md = d(Z), d(wh) if Wn else None, d(mask) if Wn else None
    ops.head_data_driven_fwd(Zd, whd, md, so, wo if Wn else None, S, Wn, Ww, B)
    torch.testing.assert_close(so[:, :B].cpu().double(), so_ref.detach(), rtol=2e-6, atol=1e-6)
    if Wn:
        torch.testing.assert_close(wo[:Wn, :B].cpu().double(), wo_ref.detach(), rtol=0, atol=0)
    dZ = torch.full((rows, ld), 7.0, device=dev)
    g_wh = torch.full((max(Wn, 1), Ww, ld), 0.5, device=dev)
    ops.head_data_driven_bwd(Zd, whd, md, d(g_so), d(g_wo) if Wn else None, dZ, g_wh if Wn else None, S, Wn, Ww, B)
    torch.cuda.synchronize()
    torch.testing.assert_close(dZ[:, :B].cpu().double(), z64.grad, rtol=2e-5, atol=2e-5)
    if Wn:
        torch.testing.assert_close(g_wh[:, :, :B].cpu().double() - 0.5, w64.grad, rtol=2e-5, atol=2e-5)
    assert float(dZ[:, B:].sub(7.0).abs().max()) == 0.0


@pytest.mark.parametrize("cap,self_loops", [(True, True), (False, False)])
def test_gnn_group_allocation_matches_torch_autograd(cap, self_loops):
    """nic_gnn_alloc_groups_fwd / bwd (several warehouses, one launch) against the tensor-op formulation
    (neural_networks.py:111-138 per supplying node, :1435-1492): three warehouses of 4 / 0 / 3 internal edges (the middle one
    supplies nobody: its own order passes through, nothing else), orders scattered through an edge -> row map, float64 autograd
    for the gradients w.r.t. the desired quantities and the warehouses' on-hand stock; the demand-edge rows of d_out are cleared."""
    dev, B, Ww = "cuda", 333, 3
    ld = pad_ld(B, 32)
    gen = torch.Generator().manual_seed(5)
    counts, n_dem = [4, 0, 3], 5
    n_int, G = sum(counts), len(counts)
    e_dem = n_int + G
    selfs = [e_dem + n_dem + i if (self_loops and c_) else -1 for i, c_ in enumerate(counts)]
    E = e_dem + n_dem + G
    firsts = [0, 4, 4]
    groups = torch.tensor([[firsts[g_], counts[g_], selfs[g_], n_int + g_] for g_ in range(G)], dtype=torch.int32, device=dev)
    perm = torch.randperm(n_int + G, generator=gen).tolist()            # any injective edge -> order-row map
    order_row = torch.full((E,), -1, dtype=torch.int32)
    for e in range(n_int + G):
        order_row[e] = perm[e]
    n_rows = n_int + G
    out = torch.zeros(E, ld)
    out[:, :B] = torch.rand(E, B, generator=gen) * 3 + 0.01
    on_hand = torch.zeros(G, Ww, ld)
    on_hand[:, 0, :B] = torch.rand(G, B, generator=gen) * 9
    g_orders = torch.zeros(n_rows, ld)
    g_orders[:, :B] = torch.randn(n_rows, B, generator=gen)
    o64, h64 = out[:, :B].double().requires_grad_(True), on_hand[:, 0, :B].double().requires_grad_(True)
    want = [None] * n_rows
    for g_ in range(G):
        want[perm[n_int + g_]] = o64[n_int + g_]
        members = list(range(firsts[g_], firsts[g_] + counts[g_]))
        if not members:
            continue
        ratio = h64[g_] / (o64[members + ([selfs[g_]] if selfs[g_] >= 0 else [])].sum(dim=0) + 1e-10)
        sc = torch.clamp(ratio, max=1.0) if cap else ratio
        for e in members:
            want[perm[e]] = o64[e] * sc
    want = torch.stack(want)
    (want * g_orders[:, :B].double()).sum().backward()
    z = lambda *sh: torch.zeros(*sh, device=dev)  # noqa: E731
    orders, sums, rat, scale = z(n_rows, ld), z(G, ld), z(G, ld), z(G, ld)
    oh_d = on_hand.to(dev)
    ops.gnn_alloc_groups_fwd(out.to(dev), oh_d, orders, sums, rat, scale, groups, order_row.to(dev), cap, B)
    torch.testing.assert_close(orders[:, :B].cpu().double(), want.detach(), rtol=2e-6, atol=1e-6)
    assert float(orders[:, B:].abs().max()) == 0.0
    d_out, g_on = torch.full((E, ld), 9.0, device=dev), torch.full((G, Ww, ld), 0.25, device=dev)
    ops.gnn_alloc_groups_bwd(out.to(dev), oh_d, g_orders.to(dev), sums, rat, scale, d_out, g_on, groups, order_row.to(dev),
                             e_dem, n_dem, cap, B)
    torch.cuda.synchronize()
    ref = o64.grad.clone()
    got = d_out[:, :B].cpu().double()
    rows_written = [e for e in range(E) if e < e_dem + n_dem or e in selfs]
    torch.testing.assert_close(got[rows_written], ref[rows_written], rtol=2e-5, atol=2e-5)
    untouched = [e for e in range(E) if e not in rows_written]
    if untouched:   # (self-loop rows that do not exist in this configuration are nobody's to write)
        assert float(got[untouched].sub(9.0).abs().max()) == 0.0
    supplying = [g_ for g_ in range(G) if counts[g_]]
    torch.testing.assert_close(g_on[supplying, 0, :B].cpu().double() - 0.25, h64.grad[supplying], rtol=2e-5, atol=2e-5)
    assert float(g_on[1, 0, :B].sub(0.25).abs().max()) == 0.0 and float(g_on[:, 1:].sub(0.25).abs().max()) == 0.0
    assert float(d_out[:, B:].sub(9.0).abs().max()) == 0.0   # padding columns untouched


@pytest.mark.parametrize("S,e_self,cap", [(16, 33, True), (3, None, False), (5, 11, False)])
def test_gnn_allocation_head_matches_torch_autograd(S, e_self, cap):
    """nic_gnn_alloc_fwd / bwd against the tensor-op formulation it replaced (neural_networks.py:111-138): orders, and through
    float64 autograd the gradients w.r.t. the desired quantities and the on-hand stock, on a ragged scenario count."""
    dev, B = "cuda", 777
    ld = pad_ld(B, 32)
    gen = torch.Generator().manual_seed(S)
    E, e_sup = 2 * S + 2, S
    out = torch.zeros(E, ld)
    out[:, :B] = torch.rand(E, B, generator=gen) * 3 + 0.01
    on_hand = torch.zeros(ld)
    on_hand[:B] = torch.rand(B, generator=gen) * S * 3
    g_orders = torch.zeros(S + 1, ld)
    g_orders[:, :B] = torch.randn(S + 1, B, generator=gen)
    members = list(range(S)) + ([e_self] if e_self is not None else [])
    o64, h64 = out[:, :B].double().requires_grad_(True), on_hand[:B].double().requires_grad_(True)
    ratio = h64 / (o64[members].sum(dim=0) + 1e-10)
    sc = torch.clamp(ratio, max=1.0) if cap else ratio
    want = torch.cat([o64[:S] * sc, o64[e_sup:e_sup + 1]], dim=0)
    (want * g_orders[:, :B].double()).sum().backward()
    z = lambda *sh: torch.zeros(*sh, device=dev)  # noqa: E731
    orders, sums, rat, scale = z(S + 1, ld), z(ld), z(ld), z(ld)
    ops.gnn_alloc_fwd(out.to(dev), on_hand.to(dev), orders, sums, rat, scale, S, e_self, e_sup, cap, B)
    torch.testing.assert_close(orders[:, :B].cpu().double(), want.detach(), rtol=2e-6, atol=1e-6)
    d_out, g_on = torch.full((E, ld), 9.0, device=dev), torch.full((ld,), 0.25, device=dev)
    ops.gnn_alloc_bwd(out.to(dev), on_hand.to(dev), g_orders.to(dev), sums, rat, scale, d_out, g_on, S, e_self, e_sup, cap, B)
    torch.cuda.synchronize()
    torch.testing.assert_close(d_out[:, :B].cpu().double(), o64.grad, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(g_on[:B].cpu().double() - 0.25, h64.grad, rtol=2e-5, atol=2e-5)
    assert float(d_out[:, B:].sub(9.0).abs().max()) == 0.0   # padding columns untouched


@pytest.mark.parametrize("n_rows,P,stride,T,ld,ignore", [(2048, 2212, 2212, 100, 32768, 0), (1024, 1332, 1332, 100, 16384, 10),
                                                         (16, 2212, 2212, 50, 256, 3), (5, 70, 72, 3, 32, 1), (333, 257, 260, 7, 96, 0),
                                                         (64, 256, 256, 1, 32, 0)])
def test_small_rollout_reduce_matches_float64_sums(n_rows, P, stride, T, ld, ignore):
    """nic_small_rollout_reduce (csrc/small_reduce.hip): the sum of the per-wavefront partial gradients and the total / reported cost
    of a training step of the small policies, against float64 sums of the same buffers; rows and columns past the live ones are
    never read (NaN-poisoned); the same call twice gives the same bits; each pair alone."""
    from neural_inventory_control_amd import small_rollout as sr
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(n_rows * 7 + P)
    slab = torch.full((n_rows + 3, stride), float("nan"), device=dev)
    slab[:n_rows, :P] = torch.randn(n_rows, P, device=dev, generator=g) * 3.0
    rewards = torch.randn(T, ld, device=dev, generator=g).abs() * 10.0
    grad, totals = torch.full((P,), float("nan"), device=dev), torch.zeros(2, device=dev)
    scratch = torch.empty(sr.small_rollout_reduce_scratch(n_rows, P, rewards.numel()), device=dev)
    sr.small_rollout_reduce(slab, n_rows, grad, rewards, ignore, totals, scratch)
    want_g = slab[:n_rows, :P].double().sum(dim=0)
    torch.testing.assert_close(grad.double(), want_g, rtol=2e-6, atol=2e-6 * float(slab[:n_rows, :P].abs().sum(dim=0).max()))
    want_t, want_r = float(rewards.double().sum()), float(rewards[ignore:].double().sum())
    assert abs(float(totals[0]) - want_t) <= 2e-6 * want_t and abs(float(totals[1]) - want_r) <= 2e-6 * want_r
    g1, t1 = grad.clone(), totals.clone()
    sr.small_rollout_reduce(slab, n_rows, grad, rewards, ignore, totals, scratch)
    assert torch.equal(grad, g1) and torch.equal(totals, t1)
    grad.fill_(float("nan"))
    totals.fill_(float("nan"))
    sr.small_rollout_reduce(slab, n_rows, grad, None, 0, None, scratch)
    assert torch.equal(grad, g1) and bool(torch.isnan(totals).all())
    sr.small_rollout_reduce(None, 0, None, rewards, ignore, totals, scratch)
    assert torch.equal(totals, t1)
