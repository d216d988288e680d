"""the HIP demand sampler (`Scenario(sampler="hip")`, csrc/sampler.hip) against the ORACLE's generator
(`oracle.generate_scenario_data` = the reference's numpy path, data_handling.py:125-237, pinned bit-equal to the reference).

numpy's MT19937 + SVD `multivariate_normal` stream cannot be reproduced on a GPU, so parity is distributional — but it is
parity *against the oracle's draws for the same YAML*, not against the parameters the test passed in:

  * the distribution parameters the two paths end up with (per-store `mean.round(3)`, `std = (mean * cv).round(3)`, the
    shifted one-store demand seed) are bit-equal, and so is every static tensor;
  * per store: two-sample Kolmogorov-Smirnov (normal) / chi-square homogeneity (Poisson) between HIP and oracle demand;
  * clip-at-0 semantics: no negative demand, same share of exact zeros;
  * covariance: Frobenius distance between the two sample covariances, judged against the distance between two ORACLE draws
    under different seeds (the HIP sampler must be as close to the oracle as the oracle is to itself);
  * the initial store pipelines are the same multiplier draw scaled by each path's own global demand mean;
  * the reference's shipped checkpoint evaluated on HIP-sampled demand lands on its known dev loss 6.854347 within the
    two-sample confidence interval.
"""
import copy
from collections import defaultdict

import numpy as np
import pytest
import torch
from scipy import stats

from golden_io import Golden
from neural_inventory_control_amd import workloads
from neural_inventory_control_amd.data_handling import Scenario
from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
from neural_inventory_control_amd.rollout import FusedRollout
from oracle import inventory_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FAMILY_ALPHA = 1e-3  # family-wise error rate of every batch of per-store tests (Bonferroni over the stores)


def _oracle_data(setting, n, T, demand_seed=None):
    s = copy.deepcopy(setting)
    if demand_seed is not None:
        s["seeds"]["demand"] = demand_seed
    obs = defaultdict(lambda: None, s["observation_params"])
    data = orc.generate_scenario_data(T, s["problem_params"], s["store_params"], s["warehouse_params"], s["echelon_params"],
                                      n, obs, s["seeds"])
    return s, data


def _hip_scenario(setting, n, T):
    s = copy.deepcopy(setting)
    obs = defaultdict(lambda: None, s["observation_params"])
    sc = Scenario(T, s["problem_params"], s["store_params"], s["warehouse_params"], s["echelon_params"], n, obs, s["seeds"],
                  sampler="hip", device=DEV)
    return s, sc


def _per_store(d):
    """(N, S, T) -> float64 numpy (S, N*T)"""
    return d.permute(1, 0, 2).reshape(d.shape[1], -1).double().cpu().numpy()


@pytest.mark.parametrize("workload,n", [("cfg1", 4096), ("cfg2", 2048), ("cfg3", 2048), ("cfg5", 1024)])
def test_hip_sampler_matches_the_oracle_generator(workload, n):
    setting, _, _, T, _ = workloads.get(workload)
    s_orc, ref = _oracle_data(setting, n, T)
    s_hip, sc = _hip_scenario(setting, n, T)
    got = sc.get_data()
    S = setting["problem_params"]["n_stores"]
    kind = setting["store_params"]["demand"]["distribution"]

    # (1) same distribution parameters after the reference's in-place mutations (:155-158, :175-176, :231-236), bit for bit
    assert s_hip["seeds"]["demand"] == s_orc["seeds"]["demand"]
    for k in ("mean", "std"):
        if k in s_orc["store_params"]["demand"]:
            assert np.array_equal(np.asarray(s_hip["store_params"]["demand"][k]), np.asarray(s_orc["store_params"]["demand"][k])), k
    # ... and every static tensor (numpy host path on both sides)
    assert set(got) == set(ref)
    for k in ref:
        if k not in ("demands", "initial_inventories"):
            assert torch.equal(got[k].cpu(), ref[k]), k

    hip, orc_d = _per_store(got["demands"]), _per_store(ref["demands"])
    assert hip.shape == orc_d.shape == (S, n * T)

    # (2) clip semantics (:145-146): nothing below zero; the atom at zero has the same mass
    if setting["store_params"]["demand"]["clip"]:
        assert hip.min() >= 0.0 and orc_d.min() >= 0.0
        z_hip, z_orc = (hip == 0).mean(axis=1), (orc_d == 0).mean(axis=1)
        se = np.sqrt(np.maximum(z_orc, 1.0 / (n * T)) * 2.0 / (n * T))
        assert np.all(np.abs(z_hip - z_orc) <= 5.0 * se + 1e-12), (z_hip, z_orc)

    # (3) per-store marginals: two-sample tests HIP vs oracle
    alpha = FAMILY_ALPHA / S
    pvals = []
    for s in range(S):
        if kind == "poisson":
            assert np.array_equal(hip[s], np.round(hip[s]))  # integer-valued like np.random.poisson (:205-211)
            top = int(max(hip[s].max(), orc_d[s].max()))
            a = np.bincount(hip[s].astype(np.int64), minlength=top + 1)
            b = np.bincount(orc_d[s].astype(np.int64), minlength=top + 1)
            keep = (a + b) >= 10  # pool the thin tail into one cell
            table = np.stack([np.append(a[keep], a[~keep].sum()), np.append(b[keep], b[~keep].sum())])
            table = table[:, table.sum(axis=0) > 0]
            pvals.append(stats.chi2_contingency(table)[1])
        else:
            pvals.append(stats.ks_2samp(hip[s], orc_d[s]).pvalue)
    assert min(pvals) > alpha, (workload, min(pvals), alpha)

    # (4) first two moments per store against the oracle's SAMPLE moments (z-scores of two-sample differences)
    m_h, m_o, v_h, v_o = hip.mean(axis=1), orc_d.mean(axis=1), hip.var(axis=1), orc_d.var(axis=1)
    # within a scenario-period the stores are correlated, across scenario-periods independent: n*T independent draws per store
    se_mean = np.sqrt((v_h + v_o) / (n * T))
    assert np.all(np.abs(m_h - m_o) <= 5.0 * se_mean), (m_h - m_o, se_mean)
    se_var = np.sqrt(2.0 * (v_h ** 2 + v_o ** 2) / (n * T)) * 1.5  # (clipped / Poisson tails: kurtosis margin)
    assert np.all(np.abs(v_h - v_o) <= 5.0 * se_var), (v_h - v_o, se_var)

    # (5) covariance across stores (:194-201): HIP vs oracle no further apart than two oracle draws under different seeds
    if S > 1:
        c_h, c_o = np.cov(hip), np.cov(orc_d)
        d_hip = np.linalg.norm(c_h - c_o) / np.linalg.norm(c_o)
        selfs = []
        for alt in (1001, 2002, 3003):
            _, other = _oracle_data(setting, n, T, demand_seed=alt)
            selfs.append(np.linalg.norm(np.cov(_per_store(other["demands"])) - c_o) / np.linalg.norm(c_o))
        assert d_hip <= 2.0 * max(selfs), (d_hip, selfs)
        assert d_hip <= 0.03, d_hip
        # the equicorrelated structure itself: every off-diagonal correlation is the YAML's rho
        corr_h = np.corrcoef(hip)
        corr_o = np.corrcoef(orc_d)
        off = ~np.eye(S, dtype=bool)
        assert np.abs(corr_h[off] - corr_o[off]).max() <= 6.0 * np.sqrt(2.0 / (n * T))

    # (6) initial store pipelines (:290-310): the same multiplier draw, scaled by each path's own global demand mean (:298)
    mean_h = torch.from_numpy(m_h).float()
    mean_o = ref["demands"].float().mean(dim=2).mean(dim=0)
    ratio = (mean_o / mean_h)[None, :, None]
    torch.testing.assert_close(got["initial_inventories"].cpu() * ratio, ref["initial_inventories"], rtol=2e-5, atol=1e-6)


def test_independent_across_scenarios_and_periods_like_the_oracle():
    """Lag-1 autocorrelation over periods and correlation between neighbouring scenarios: zero for the oracle's iid draws
    (:185-201) and for the HIP sampler's (scenario, period)-keyed counters, at the same resolution."""
    setting, _, _, T, _ = workloads.get("cfg3")
    n = 4096
    _, sc = _hip_scenario(setting, n, T)
    _, ref = _oracle_data(setting, n, T)
    for name, d in (("hip", sc.get_data()["demands"].cpu()), ("oracle", ref["demands"])):
        x = d.double()
        x = x - x.mean(dim=(0, 2), keepdim=True)
        sd = x.pow(2).mean(dim=(0, 2)).sqrt()
        lag = (x[:, :, 1:] * x[:, :, :-1]).mean(dim=(0, 2)) / sd ** 2
        nb = (x[1:] * x[:-1]).mean(dim=(0, 2)) / sd ** 2
        bound = 5.0 / np.sqrt(n * (T - 1))
        assert float(lag.abs().max()) < bound and float(nb.abs().max()) < bound, (name, float(lag.abs().max()), float(nb.abs().max()))


def test_shipped_checkpoint_on_hip_sampled_demand_reproduces_its_known_answer():
    """The reference's shipped `one_store_lost` checkpoint (best dev loss 6.854347610473633 on numpy demand) evaluated by the
    HIP engine on HIP-sampled Poisson demand: 32,768 scenarios x T=50, first 30 periods ignored.  The two dev sets are
    independent samples of the same process, so the two means must agree within the two-sample confidence interval built
    from the per-scenario costs of BOTH runs (Welch)."""
    g = Golden("checkpoint_kat")
    c = g.fresh_config()
    n, T, ignore = c["dev_samples"], c["periods"], c["ignore"]

    def per_scenario_cost(sampler):
        cc = g.fresh_config()
        # numpy side: exactly the known answer's dev set = the first `dev_samples` rows of the 65,536-scenario draw (the initial
        # pipelines use the demand mean of the whole draw, data_handling.py:298)
        total = cc["scenario_samples"] if sampler == "numpy" else n
        sc = Scenario(cc["scenario_periods"], cc["problem_params"], cc["store_params"], None, None, total,
                      cc["observation_params"], cc["seeds"], sampler=sampler, device=DEV)
        model = NeuralNetworkCreator().create_neural_network(sc, cc["nn_params"], device=DEV)
        eng = FusedRollout(model, cc["problem_params"], DEV)
        eng.materialize(4)
        model.load_state_dict({k: v.to(DEV) for k, v in g.params.items()})
        data = {k: v[:n].to(DEV) for k, v in sc.get_data().items()}
        with torch.no_grad():
            _, reported = eng.run(data, T, ignore, train=False, observation_params=cc["observation_params"],
                                  demand_soa=sc.demands_soa)
        r = eng.per_period_rewards()[ignore:].double().sum(dim=0) / (T - ignore)  # per-scenario mean cost per period
        assert abs(float(r.mean()) - float(reported) / (n * (T - ignore))) < 1e-6
        return r.cpu().numpy()

    ref_cost = per_scenario_cost("numpy")
    assert abs(ref_cost.mean() - 6.854347610473633) < 2e-6  # the known answer on the reference's own demand
    hip_cost = per_scenario_cost("hip")
    se = np.sqrt(ref_cost.var(ddof=1) / n + hip_cost.var(ddof=1) / n)
    assert abs(hip_cost.mean() - 6.854347610473633) <= 4.0 * se, (hip_cost.mean(), se)
    # same spread of per-scenario costs (F-test style bound on the variance ratio at 32,768 scenarios each)
    assert abs(np.log(hip_cost.var() / ref_cost.var())) < 0.1
    assert stats.ks_2samp(hip_cost, ref_cost).pvalue > FAMILY_ALPHA
