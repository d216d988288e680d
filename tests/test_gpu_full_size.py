"""Size-independent properties at BASELINE's FULL per-GPU sizes (the golden fixtures pin the arithmetic at sizes the CPU oracle
finishes in seconds; these pin the size-dependent paths: 2^31 offset guards, ragged last wavefronts, every scenario block of the
grid, the four-periods-per-lane form of the sampler):

  * batch independence - a scenario's trajectory does not depend on the batch it sits in: the first scenarios of the full batch
    reproduce a run of the same scenarios BIT FOR BIT (per-period rewards) on the whole-horizon route and among per-period batches
    that take the same GEMM contraction form (full batch vs its half); a tiny batch on the per-period route takes the split-K
    streamed kernels (round 4) and agrees to 2e-6;
  * additivity of the training step - the parameter gradient of the full batch equals the sum of the gradients of its two halves
    (same global normalisation), i.e. every scenario block contributes once and only once;
  * stock conservation over every scenario of the batch (lost demand: pipeline' = pipeline - sales + orders received);
  * demand comes from `Scenario(sampler="hip")` at full size (131,072 scenarios x T = 100: the four-periods-per-lane form).
"""
from collections import defaultdict

import pytest
import torch

from neural_inventory_control_amd import _lib, workloads
from neural_inventory_control_amd.data_handling import Scenario
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
from neural_inventory_control_amd.rollout import FusedRollout

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(workload, n, T, T_demand=None):
    """T_demand: horizon of the sampled demand trace (default T); the rollout then uses its first T periods."""
    setting, policy, _, _, _ = workloads.get(workload)
    obs = defaultdict(lambda: None, setting["observation_params"])
    sc = Scenario(T_demand or T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n,
                  obs, setting["seeds"], sampler="hip", device=DEV)
    data = {k: v.to(DEV) for k, v in sc.get_data().items()}
    torch.manual_seed(7)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
    F_in = data["initial_inventories"].shape[1] * data["initial_inventories"].shape[2]
    if policy["name"] != "vanilla_one_store":
        F_in += sum(data[k].shape[1] * data[k].shape[2] for k in ("initial_warehouse_inventories", "initial_echelon_inventories")
                    if k in data)
    eng = FusedRollout(model, setting["problem_params"], DEV)
    eng.materialize(F_in)
    return setting, sc, data, model, eng, obs


def _slice(data, lo, hi):
    return {k: v[lo:hi].contiguous() for k, v in data.items()}


def _grads(eng, model, data, T, obs, scale):
    eng.run(data, T, 0, train=True, observation_params=obs, grad_scale=scale)
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("workload,n,T", [("cfg2", 32768, 12), ("cfg4", 16384, 12), ("cfg4", 131072, 8)])
def test_whole_horizon_route_at_full_size(workload, n, T):
    """cfg2 (32,768 one-store scenarios) and cfg4 (16,384 per GPU and the whole 131,072-scenario job on one GPU): the
    whole-horizon kernels, one wavefront per 32 scenarios."""
    # the demand trace is drawn at the BENCHMARK horizon (T = 100) by the one-store form of the sampler (round 6: four scenarios x
    # four periods per lane, every word of a Philox block used); the rollout uses the first T periods
    setting, sc, data, model, eng, obs = _case(workload, n, T, T_demand=100)
    assert _lib.lib().nic_last_kernel() == b"sample_one_store_kernel"
    S = setting["problem_params"]["n_stores"]
    scale = 1.0 / (n * T * S)
    from neural_inventory_control_amd.rollout import KernelTimer
    eng.timer = KernelTimer()   # (records which kernel the C ABI launched for the forward)
    with torch.no_grad():
        eng.run(data, T, 0, train=False, observation_params=obs, demand_soa=sc.demands_soa)
    assert eng.small is not None  # the whole-horizon route
    # (the library picks 16 or 32 scenarios per wavefront from the batch size; the two forms sum a layer's contraction in
    # different orders, so the small batches below are run in the form the full batch took)
    width = 16 if "small_rollout16" in eng.timer.names["small_rollout_fwd"] else 32
    eng.timer = None
    r_full = eng.per_period_rewards().clone()
    assert r_full.shape == (T, n) and bool(torch.isfinite(r_full).all())
    # batch independence, bit for bit: a ragged 45-scenario batch from the middle of the grid and the last 33 scenarios
    for lo, hi in ((0, 45), (n // 2 - 7, n // 2 + 38), (n - 33, n)):
        small = FusedRollout(model, setting["problem_params"], DEV)
        small.small_lane_scenarios = width
        with torch.no_grad():
            small.run(_slice(data, lo, hi), T, 0, train=False, observation_params=obs)
        assert torch.equal(small.per_period_rewards(), r_full[:, lo:hi]), (lo, hi)
    # additivity of the training step over the two halves of the batch
    g_full = _grads(eng, model, data, T, obs, scale)
    half = n // 2
    g_a = _grads(eng, model, _slice(data, 0, half), T, obs, scale)
    g_b = _grads(eng, model, _slice(data, half, n), T, obs, scale)
    for gf, ga, gb in zip(g_full, g_a, g_b):
        assert float((gf - (ga + gb)).norm()) <= 2e-5 * float(gf.norm()) + 1e-12


def test_cfg5_per_period_route_at_full_size():
    """cfg5's per-GPU shape (32,768 scenarios x 64 stores x 3 warehouses, 16 stores per lane of a quad), demand from the HIP
    sampler: batch independence, stock conservation over all 2.1 M store pipelines,
    additivity of the training step."""
    n, T = 32768, 6
    setting, sc, data, model, eng, obs = _case("cfg5", n, T)
    pp = setting["problem_params"]
    S, Wn = pp["n_stores"], pp["n_warehouses"]
    # counters are keyed by (seed, global scenario, period): the benchmark horizon's trace starts with the short horizon's
    s70 = workloads.get("cfg5")[0]
    big = Scenario(70, s70["problem_params"], s70["store_params"], s70["warehouse_params"], s70["echelon_params"], n,
                   defaultdict(lambda: None, s70["observation_params"]), s70["seeds"], sampler="hip", device=DEV)
    assert big.demands_soa.shape[0] == 70 and float(big.demands_soa.min()) >= 0.0
    assert torch.equal(big.demands_soa[:T], sc.demands_soa[:T])
    with torch.no_grad():
        eng.run(data, T, 0, train=False, observation_params=obs, demand_soa=sc.demands_soa)
    assert eng.small is None
    r_full = eng.per_period_rewards().clone()
    states, orders, demand = eng.states.clone(), eng.orders.clone(), eng.demand
    # batch independence.  Bit for bit among batches whose layers take the same contraction form: the full batch against its
    # first half (every layer on the LDS-DMA kernels, whose k order does not depend on the tiling).  Round 4: launches of at
    # most 1,024 output tiles take the streamed kernel, where up to four wavefronts split a layer's contraction and add their
    # partial sums once - a 40-scenario batch therefore differs from the full batch in the last bits of each layer (same
    # products, another association): bounded at 2e-6 of the per-period cost.
    halfb = FusedRollout(model, pp, DEV)
    with torch.no_grad():
        halfb.run(_slice(data, 0, n // 2), T, 0, train=False, observation_params=obs)
    assert torch.equal(halfb.per_period_rewards(), r_full[:, :n // 2])
    del halfb
    for lo, hi in ((0, 40), (n - 70, n)):
        small = FusedRollout(model, pp, DEV)
        with torch.no_grad():
            small.run(_slice(data, lo, hi), T, 0, train=False, observation_params=obs)
        torch.testing.assert_close(small.per_period_rewards(), r_full[:, lo:hi], rtol=2e-6, atol=1e-4)
    Ws = data["initial_inventories"].shape[2]
    for t in range(T):
        st = states[t][:S * Ws].view(S, Ws, -1)[:, :, :n].double()
        nx = states[t + 1][:S * Ws].view(S, Ws, -1)[:, :, :n].double()
        dem = demand[t][:, :n].double()
        sales = torch.minimum(st[:, 0], dem)
        recv = orders[t][:S * Wn].view(S, Wn, -1)[:, :, :n].double().sum(dim=1)   # what every warehouse ships to the store
        assert float((nx.sum(dim=1) - (st.sum(dim=1) - sales + recv)).abs().max()) < 2e-3
    scale = 1.0 / (n * T * S)
    g_full = _grads(eng, model, data, T, obs, scale)
    half = n // 2
    g_a = _grads(eng, model, _slice(data, 0, half), T, obs, scale)
    g_b = _grads(eng, model, _slice(data, half, n), T, obs, scale)
    for gf, ga, gb in zip(g_full, g_a, g_b):
        assert float((gf - (ga + gb)).norm()) <= 2e-5 * float(gf.norm()) + 1e-12


# ---- BASELINE's full sizes at the FULL horizon (round 4: the tests above run 6-12 periods) -----------------------------------------
@pytest.mark.parametrize("workload,n,T", [("cfg2", 32768, 100), ("cfg3", 65536, 100), ("cfg4", 16384, 100), ("cfg5", 32768, 70)])
def test_full_size_full_horizon_training_step_properties(workload, n, T):
    """One training step at BASELINE's scenario count AND horizon (cfg3: 65,536 x 16 x T = 100 - the bench's headline step; cfg5:
    32,768 x 64 x T = 70; cfg2 / cfg4: T = 100 on the whole-horizon kernels), with assertions the bench itself does not make:
      * additivity: total cost and parameter gradient of the full batch = sum over its two halves (same global normalisation) -
        every scenario block and every period contributes exactly once, through T periods of recurrence;
      * stock conservation over every scenario and period of the lost-demand settings (per-period route keeps the history);
      * per-period costs are finite and non-negative, padding columns of the reward buffer stay zero."""
    setting, sc, data, model, eng, obs = _case(workload, n, T)
    pp = setting["problem_params"]
    S, Wn = pp["n_stores"], pp["n_warehouses"]
    scale = 1.0 / (n * T * S)
    tot_full, _ = eng.run(data, T, 0, train=True, observation_params=obs, grad_scale=scale, demand_soa=sc.demands_soa)
    torch.cuda.synchronize()
    g_full = [p.grad.detach().clone() for p in model.parameters()]
    tot_full = float(tot_full)
    r = eng.rewards
    assert r.shape[0] == T and bool(torch.isfinite(r).all()) and float(r.min()) >= 0.0 and float(r[:, n:].abs().sum()) == 0.0
    if eng.small is None and pp["lost_demand"] and pp["n_extra_echelons"] == 0:
        Ws = data["initial_inventories"].shape[2]
        worst = 0.0
        for t in range(0, T, 7):   # (every 7th period: the check reads 3 state-sized blocks per period)
            st = eng.states[t][:S * Ws].view(S, Ws, -1)[:, :, :n].double()
            nx = eng.states[t + 1][:S * Ws].view(S, Ws, -1)[:, :, :n].double()
            dem = eng.demand[t][:, :n].double()
            sales = torch.minimum(st[:, 0], dem)
            recv = eng.orders[t][:S * max(Wn, 1)].view(S, max(Wn, 1), -1)[:, :, :n].double().sum(dim=1)
            worst = max(worst, float((nx.sum(dim=1) - (st.sum(dim=1) - sales + recv)).abs().max()))
        assert worst < 5e-3, worst
    half = n // 2
    ta, _ = eng.run(_slice(data, 0, half), T, 0, train=True, observation_params=obs, grad_scale=scale)
    torch.cuda.synchronize()
    g_a, ta = [p.grad.detach().clone() for p in model.parameters()], float(ta)
    tb, _ = eng.run(_slice(data, half, n), T, 0, train=True, observation_params=obs, grad_scale=scale)
    torch.cuda.synchronize()
    g_b, tb = [p.grad.detach().clone() for p in model.parameters()], float(tb)
    assert abs(tot_full - (ta + tb)) <= 2e-6 * abs(tot_full), (tot_full, ta, tb)
    for gf, ga, gb in zip(g_full, g_a, g_b):
        assert float((gf - (ga + gb)).norm()) <= 3e-5 * float(gf.norm()) + 1e-12, float((gf - (ga + gb)).norm() / gf.norm())


@pytest.mark.parametrize("n", [72, 288, 40])
def test_data_driven_whole_horizon_route_at_the_reference_batch(n):
    """The reference's real-data training batch (many_warehouses_real_data_lost_demand.yml: 72 products per training batch, 288 per
    dev / test batch; 21 stores x 3 warehouses, T = 95, data_driven 64 x 64) and a ragged one (40: the last workgroup has 8 live
    scenarios) on the whole-horizon kernels (csrc/horizon_rollout.hip) against the per-period route of the same engine:
      * per-period rewards and final state agree to rounding (the routes differ in the layers' summation order only), gradients to
        1e-4 of each tensor's norm after 95 periods of recurrence;
      * stock conservation over every scenario and period from the state / order histories the forward kernel leaves;
      * batch independence: a scenario's trajectory does not depend on the block of 16 it sits in or on its column (to 2e-6: the
        GEMM over (period x scenario) columns that contracts the observation rows splits its contraction by the column count);
      * additivity of the training step over two ragged parts; padding columns of every history stay zero."""
    import bench
    T = 95
    setting, policy, sc, data, model, eng, n, T, _ = bench.build_case("real_data_driven", torch.device(DEV), 0, 1, n, T, False)
    obs, pp = setting["observation_params"], setting["problem_params"]
    S, Wn = pp["n_stores"], pp["n_warehouses"]
    eng.materialize(eng.input_rows(data, obs))
    scale = 1.0 / (n * T * S)
    out = {}
    for route in ("horizon", "periods"):
        eng.use_horizon = route == "horizon"
        tot, _ = eng.run(data, T, 0, train=True, observation_params=obs, demand_soa=sc.demands_soa, grad_scale=scale)
        torch.cuda.synchronize()
        assert (eng.horizon is not None) == (route == "horizon")
        out[route] = (float(tot), eng.per_period_rewards().clone(), {k: v.clone() for k, v in eng.final_state().items()},
                      [p.grad.detach().clone() for p in model.parameters()])
    a, b = out["horizon"], out["periods"]
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0])
    torch.testing.assert_close(a[1], b[1], rtol=2e-5, atol=2e-3)
    for k in b[2]:
        torch.testing.assert_close(a[2][k], b[2][k], rtol=2e-5, atol=2e-3)
    for ga, gb in zip(a[3], b[3]):
        assert float((ga - gb).norm()) <= 1e-4 * float(gb.norm()) + 1e-12, float((ga - gb).norm() / gb.norm())
    # ---- the histories of the whole-horizon forward: conservation (lost demand), zero padding
    eng.use_horizon = True
    eng.run(data, T, 0, train=True, observation_params=obs, demand_soa=sc.demands_soa, grad_scale=scale)
    torch.cuda.synchronize()
    Ws = data["initial_inventories"].shape[2]
    X, orders = eng.hz_X, eng.hz_hist[3]
    shift = obs["demand"]["period_shift"]
    worst = 0.0
    for t in range(T - 1):
        st = X[:S * Ws, t].view(S, Ws, -1)[:, :, :n].double()
        nx = X[:S * Ws, t + 1].view(S, Ws, -1)[:, :, :n].double()
        dem = eng.demand[t + shift][:, :n].double()
        recv = orders[:S * Wn, t].view(S, Wn, -1)[:, :, :n].double().sum(dim=1)
        worst = max(worst, float((nx.sum(dim=1) - (st.sum(dim=1) - torch.minimum(st[:, 0], dem) + recv)).abs().max()))
        shipped = orders[S * Wn + Wn:, t, :n].double()      # what the kernel recorded as shipped = the sum of a warehouse's orders
        assert float((shipped - orders[:S * Wn, t].view(S, Wn, -1)[:, :, :n].double().sum(dim=0)).abs().max()) < 1e-3
    assert worst < 5e-3, worst
    for h in [eng.rewards, X[:eng.F_dyn]] + eng.hz_hist + eng.hz_dz:
        assert float(h[..., n:].abs().sum()) == 0.0
    # ---- batch independence (bit for bit) and additivity over ragged parts
    r_full = eng.per_period_rewards().clone()
    g_full = [p.grad.detach().clone() for p in model.parameters()]
    cut = 23 if n > 23 else n // 2
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        part = _slice(data, lo, hi)
        eng.run(part, T, 0, train=True, observation_params=obs, grad_scale=scale)
        torch.cuda.synchronize()
        assert eng.horizon is not None
        torch.testing.assert_close(eng.per_period_rewards(), r_full[:, lo:hi], rtol=2e-6, atol=1e-4)
        parts.append([p.grad.detach().clone() for p in model.parameters()])
    for gf, ga, gb in zip(g_full, *parts):
        assert float((gf - (ga + gb)).norm()) <= 3e-5 * float(gf.norm()) + 1e-12, float((gf - (ga + gb)).norm() / gf.norm())


@pytest.mark.parametrize("S,Wn,hidden,n,T,P", [(8, 2, [48, 16], 19, 9, 5), (3, 1, [16, 64], 5, 4, 3), (30, 4, [64, 32], 33, 6, 4),
                                              (21, 3, [64, 64], 130, 7, 16)])
def test_data_driven_whole_horizon_route_on_other_shapes(S, Wn, hidden, n, T, P):
    """The whole-horizon data_driven kernels away from the reference's own shape: other store / warehouse counts (30 x 4 = 124
    logits rows: the second output tile of every wavefront; 3 x 1: one ragged tile of everything), unequal hidden widths below 64,
    short past-demand windows, batches that leave the last workgroup ragged (19, 5, 33) or span nine of them (130) - against the
    per-period route of the same engine: per-period rewards, final state, gradients."""
    from neural_inventory_control_amd.data_handling import DatasetCreator
    setting = workloads.real_data(n_products=n, n_stores=S, n_warehouses=Wn, weeks=P + T + 6, past_periods=P, seed=S)
    policy = workloads.data_driven_policy()
    policy["neurons_per_hidden_layer"] = {"master": list(hidden)}
    obs = defaultdict(lambda: None, setting["observation_params"])
    shift = obs["demand"]["period_shift"]
    sc = Scenario(shift + T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"],
                  n, obs, setting["seeds"], device=DEV)
    data = {k: v.to(DEV) for k, v in DatasetCreator().split_by_period(sc, [f"(0, {shift + T})"])[0].items()}
    torch.manual_seed(3)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
    eng = FusedRollout(model, setting["problem_params"], DEV)
    eng.materialize(eng.input_rows(data, obs))
    with torch.no_grad():   # (weights away from their initialisation: logits of both signs, allocation scale both clipped and not)
        for p_ in model.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    out = {}
    for route in ("horizon", "periods"):
        eng.use_horizon = route == "horizon"
        tot, rep = eng.run(data, T, 2, train=True, observation_params=obs)
        torch.cuda.synchronize()
        assert (eng.horizon is not None) == (route == "horizon")
        out[route] = (float(tot), float(rep), eng.per_period_rewards().clone(), {k: v.clone() for k, v in eng.final_state().items()},
                      [p_.grad.detach().clone() for p_ in model.parameters()])
        with torch.no_grad():
            t2, _ = eng.run(data, T, 2, train=False, observation_params=obs)
        assert float(t2) == float(tot)
    a, b = out["horizon"], out["periods"]
    assert abs(a[0] - b[0]) <= 5e-6 * abs(b[0]) and abs(a[1] - b[1]) <= 5e-6 * abs(b[1])
    torch.testing.assert_close(a[2], b[2], rtol=2e-5, atol=2e-3)
    for k in b[3]:
        torch.testing.assert_close(a[3][k], b[3][k], rtol=2e-5, atol=2e-3)
    for ga, gb in zip(a[4], b[4]):
        assert float(gb.norm()) > 0 and float((ga - gb).norm()) <= 1e-4 * float(gb.norm()), float((ga - gb).norm() / gb.norm())


def test_data_driven_whole_horizon_route_differential_fuzz():
    """tools/horizon_fuzz.py, a dozen random shapes (1..30 stores, 1..4 warehouses, hidden widths 8..64, 1..250 scenarios, 2..20
    periods): the whole-horizon kernels against the per-period route - totals to 1e-5, rewards to 1e-4, gradients to 1e-3 of their
    norms (measured: 1e-7 .. 2e-6)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "horizon_fuzz.py"), "7", "12"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "suspicious: 0" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("total ") == 12


def test_tape_route_differential_fuzz():
    """tools/tape_fuzz.py, a dozen random cases (the four quantile policies and just-in-time on the one-store real-data shape, 1..700
    series, 2..30 weeks, random policy weights, stand-in forecaster): the tape route against the reference-style loop - totals to
    1e-5 (measured <= 2e-7), gradients of the trainable ones to 2e-2 (knife edges; measured <= 5e-6)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "tape_fuzz.py"), "11", "12"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "suspicious: 0" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("totals ") == 12


def test_gnn_period_kernel_at_full_size_properties():
    """The GNN policy's one-launch-per-period forward (csrc/gnn_period.hip) at the benchmark's size (8,192 scenarios x 16 stores,
    T = 12): batch independence (the first 4,096 scenarios reproduce a run of those scenarios alone - same blocks of 16, same
    arithmetic: per-period rewards bit for bit), additivity of the training step (gradient of the batch = sum of its halves'
    under the same normalisation: every 16-scenario block contributes once), and stock conservation of the stores over every
    scenario (lost demand: pipeline' = pipeline - sales + what arrives from the warehouse's allocation)."""
    from neural_inventory_control_amd.gnn_rollout import GnnRollout
    setting, policy, _, _, _ = workloads.get("gnn")
    obs = defaultdict(lambda: None, setting["observation_params"])
    n, T = 8192, 12
    sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n,
                  obs, setting["seeds"], sampler="hip", device=DEV)
    data = {k: v.to(DEV) for k, v in sc.get_data().items()}
    torch.manual_seed(7)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
    S = setting["problem_params"]["n_stores"]
    scale = 1.0 / (n * T * S)

    def engine():
        e = GnnRollout(model, setting["problem_params"], DEV)
        e.use_period_kernel = True
        e.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
        return e
    eng = engine()
    eng.run(data, T, 0, train=True, observation_params=obs, grad_scale=scale)
    torch.cuda.synchronize()
    assert eng._period
    full_rewards = eng.rewards[:, :n].clone()
    full_grads = [p.grad.detach().clone() for p in model.parameters()]
    # stock conservation of the stores: sum of the pipeline changes by (received from the warehouse) - (sales)
    st = eng.states[:, :eng.F_store, :n].view(T + 1, S, -1, n)
    orders = eng.orders[:, :S, :n]                       # one warehouse: row s = what store s receives (after allocation)
    demand = sc.demands_soa[:T, :, :n]
    sales = torch.minimum(st[:-1, :, 0], demand)
    lhs = st[1:].sum(dim=2)
    rhs = st[:-1].sum(dim=2) - sales + orders
    assert float((lhs - rhs).abs().max()) < 2e-3
    half_grads = None
    for lo, hi in ((0, n // 2), (n // 2, n)):
        e2 = engine()
        e2.run(_slice(data, lo, hi), T, 0, train=True, observation_params=obs, grad_scale=scale)
        torch.cuda.synchronize()
        if lo == 0:
            assert torch.equal(e2.rewards[:, :n // 2], full_rewards[:, :n // 2])
        g = [p.grad.detach().clone() for p in model.parameters()]
        half_grads = g if half_grads is None else [a + b for a, b in zip(half_grads, g)]
    for a, b in zip(full_grads, half_grads):
        assert float((a - b).norm() / (a.norm() + 1e-30)) <= 2e-5


def test_closed_form_at_a_million_chains_properties():
    """base_stock at 1,048,576 chains x T = 40 (the specialised single-store kernel; eight periods of demand per batch, the 40
    periods = 5 batches): batch independence (per-chain rewards of the first 65,536 chains, run alone, bit for bit), the
    per-wavefront sums (total, reported, level gradient) of the full batch equal the sum of the halves', and the final state obeys
    the stock balance of a backlogged / lost-demand store chain over the horizon."""
    from neural_inventory_control_amd.closed_form import ClosedFormRollout
    setting, policy, _, _, _ = workloads.get("base_stock_1m")
    obs = defaultdict(lambda: None, setting["observation_params"])
    n, T = 1 << 20, 40
    sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n,
                  obs, setting["seeds"], sampler="hip", device=DEV)
    data = {k: v.to(DEV) for k, v in sc.get_data().items()}
    torch.manual_seed(7)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)

    def run(d, keep_rewards=False):
        eng = ClosedFormRollout(model, setting["problem_params"], DEV)
        eng.keep_rewards = keep_rewards
        model.zero_grad()
        total, reported = eng.run(d, T, 7, train=True, observation_params=obs)
        total.backward()
        torch.cuda.synchronize()
        return eng, float(total.detach()), float(reported.detach()), [p.grad.detach().clone() for p in model.parameters()]
    with torch.no_grad():
        ClosedFormRollout(model, setting["problem_params"], DEV).model.closed_form_levels()   # materialise the lazy layer
    eng, total, reported, grads = run(data, keep_rewards=True)
    assert (_lib.lib().nic_last_kernel() or b"").decode().startswith("closed_form_kernel<1,4,false,")
    small, _, _, _ = run(_slice(data, 0, 65536), keep_rewards=True)
    assert torch.equal(small.rewards[:, :, :65536], eng.rewards[:, :, :65536])
    parts = [run(_slice(data, lo, hi)) for lo, hi in ((0, n // 2), (n // 2, n))]
    assert abs(total - (parts[0][1] + parts[1][1])) <= 2e-6 * abs(total)
    assert abs(reported - (parts[0][2] + parts[1][2])) <= 2e-6 * abs(reported)
    for g, a, b in zip(grads, parts[0][3], parts[1][3]):
        assert float((g - (a + b)).norm() / (g.norm() + 1e-30)) <= 2e-5


def test_echelon_chain_at_the_workload_size_properties():
    """echelon_stock on the serial system at the benchmark's 131,072 chains x T = 100 (the chain kernel with four levels' tangents as
    packed pairs): batch independence (per-chain rewards of the first 4,096 chains, run alone, bit for bit), the costs of the
    training launch equal the evaluation launch's bit for bit (values never depend on the tangents), the full batch's sums equal
    the halves', and the level gradient agrees with a central finite difference of the total over the bias of the policy's net
    (piecewise-linear cost: exact except for chains that cross a kink inside the step)."""
    from neural_inventory_control_amd.closed_form import ClosedFormRollout
    setting, policy, n, T, _ = workloads.get("echelon_stock")
    obs = defaultdict(lambda: None, setting["observation_params"])
    sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"], setting["echelon_params"], n,
                  obs, setting["seeds"], sampler="hip", device=DEV)
    data = {k: v.to(DEV) for k, v in sc.get_data().items()}
    torch.manual_seed(11)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=DEV)
    with torch.no_grad():
        ClosedFormRollout(model, setting["problem_params"], DEV).model.closed_form_levels()   # materialise the lazy layer

    def run(d, train=True, keep_rewards=False):
        eng = ClosedFormRollout(model, setting["problem_params"], DEV)
        eng.keep_rewards = keep_rewards
        model.zero_grad()
        if train:
            total, reported = eng.run(d, T, 9, train=True, observation_params=obs)
            total.backward()
            grads = [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]
        else:
            with torch.no_grad():
                total, reported = eng.run(d, T, 9, train=False, observation_params=obs)
            grads = []
        torch.cuda.synchronize()
        return eng, float(total.detach()), float(reported.detach()), grads
    eng, total, reported, grads = run(data, keep_rewards=True)
    assert (_lib.lib().nic_last_kernel() or b"").decode().startswith("closed_form_kernel<4,4,true")
    ev, total_e, reported_e, _ = run(data, train=False, keep_rewards=True)
    assert (_lib.lib().nic_last_kernel() or b"").decode().startswith("closed_form_kernel<0,4,true")
    assert torch.equal(ev.rewards, eng.rewards)
    # (the two scalars are torch sums over per-wavefront rows of different widths: equal to the reduction's rounding)
    assert abs(total_e - total) <= 1e-6 * abs(total) and abs(reported_e - reported) <= 1e-6 * abs(reported)
    small, _, _, _ = run(_slice(data, 0, 4096), keep_rewards=True)
    assert torch.equal(small.rewards[:, :, :4096], eng.rewards[:, :, :4096])
    parts = [run(_slice(data, lo, hi)) for lo, hi in ((0, n // 2), (n // 2, n))]
    assert abs(total - (parts[0][1] + parts[1][1])) <= 2e-6 * abs(total)
    assert abs(reported - (parts[0][2] + parts[1][2])) <= 2e-6 * abs(reported)
    for g, a, b in zip(grads, parts[0][3], parts[1][3]):
        assert float((g - (a + b)).norm() / (g.norm() + 1e-30)) <= 2e-5
    # central finite difference over each bias entry, on per-period rewards summed in float64
    bias = [p for p in model.parameters() if p.dim() == 1][0]
    g_bias = [g for p, g in zip([q for q in model.parameters() if q.grad is not None], grads) if p is bias][0]
    eps = 2e-2
    for j in range(bias.numel()):
        vals = []
        for sgn in (+1.0, -1.0):
            with torch.no_grad():
                bias[j] += sgn * eps
            e2, _, _, _ = run(data, train=False, keep_rewards=True)
            vals.append(float(e2.rewards.double().sum()))
            with torch.no_grad():
                bias[j] -= sgn * eps
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - float(g_bias[j])) <= 0.03 * abs(float(g_bias[j])) + 1e-3 * abs(total) / n, (j, fd, float(g_bias[j]))
