"""Whole-horizon rollout of the `data_driven` policy for small batches through `nic_horizon_rollout_fwd/bwd`
(csrc/horizon_rollout.hip): the reference trains DataDrivenNet (neural_networks.py:430-515) on batches of 72 products x 21 stores
(many_warehouses_real_data_lost_demand.yml:44-47), where `Trainer.simulate_batch`'s period loop (trainer.py:190-213) is ~1,050
dependent launches of a few workgroups each.  Here ONE forward and ONE backward launch walk all T periods (16 scenarios per
workgroup, weights in registers, state in LDS); the first layer's contraction with the observation rows of its input is hoisted
out of the serial chain into one GEMM over (period x scenario) columns, and the weight gradients are three such GEMMs.

Descriptor building is pointer plumbing; used by `FusedRollout` when `HorizonPlan.supports(...)`.
"""
from . import _lib, ops
from ._lib import NicHorizonDesc
from .layout import EnvProblem, Table

MAX_HIDDEN, MAX_OUT, MAX_STORES, MAX_STATE_ROWS, MAX_SLOTS = 64, 128, 64, 256, 8


class HorizonPlan:
    def __init__(self, prob: EnvProblem, dims):
        self.prob, self.dims = prob, list(dims)
        self.F_dyn = prob.S * prob.Ws + prob.Wn * prob.Ww
        self.n_ord = prob.S * prob.nsup + prob.Wn

    @staticmethod
    def supports(prob: EnvProblem, head, dims):
        """Shapes the kernels take (the C side re-checks: nic_horizon_rollout_ok)."""
        if head != "data_driven" or len(dims) != 4 or prob.E != 0:
            return False
        F_dyn = prob.S * prob.Ws + prob.Wn * prob.Ww
        n_out = prob.Wn + prob.S * prob.Wn if prob.Wn else prob.S
        return (prob.S <= MAX_STORES and F_dyn <= MAX_STATE_ROWS and F_dyn < dims[0] and dims[3] == n_out <= MAX_OUT
                and 1 <= dims[1] <= MAX_HIDDEN and 1 <= dims[2] <= MAX_HIDDEN
                and 2 <= prob.Ws <= MAX_SLOTS and (prob.Wn == 0 or 2 <= prob.Ww <= MAX_SLOTS))

    def desc(self, prob, T, t0, linears, mask, demand_soa, hist_stride, round_orders=False):
        """linears: the policy's three nn.Linear (their weights are read in place: row-major, any row stride)."""
        d = NicHorizonDesc()
        d.io = prob.make_io(None, None, None, Table.null(), Table.null(), None, None)
        d.T, d.t0 = int(T), int(t0)
        d.H1, d.H2, d.n_out = self.dims[1], self.dims[2], self.dims[3]
        d.round_orders = int(bool(round_orders))
        w1, w2, w3 = (m.weight.detach() for m in linears)
        d.W1, d.ldw1 = w1.data_ptr(), w1.stride(0)
        d.W2, d.ldw2 = w2.data_ptr(), w2.stride(0)
        d.W3, d.ldw3 = w3.data_ptr(), w3.stride(0)
        d.b2, d.b3 = linears[1].bias.data_ptr(), linears[2].bias.data_ptr()
        d.mask = _lib.ptr(mask)
        d.demand = demand_soa.data_ptr()
        d.hist_stride = int(hist_stride)
        self._keep = (prob, linears, mask, demand_soa)
        return d


def offsets_ok(prob: EnvProblem, T, t_last, hidden=0):
    """The kernels address every history with 32-bit element offsets (rows x T x ldb) and the demand trace likewise
    ((t_last + 1) x S x ldb): long evaluation horizons on big batches do not qualify (the C side re-checks)."""
    rows = max(prob.S * prob.Ws + prob.Wn * prob.Ww, prob.S * prob.nsup + 2 * prob.Wn, hidden)
    return (rows + 1) * T * prob.ldb < 2 ** 31 and (t_last + 2) * prob.S * prob.ldb < 2 ** 31


def horizon_ok(desc):
    return bool(_lib.lib().nic_horizon_rollout_ok(desc))


def horizon_fwd(desc, z1_obs, state0, rewards, state_final, state_hist, h1_hist, h2_hist, logits_hist, orders_hist):
    ops._dev(rewards)
    p = _lib.ptr
    _lib.check(_lib.lib().nic_horizon_rollout_fwd(desc, p(z1_obs), p(state0), p(rewards), p(state_final), p(state_hist), p(h1_hist),
                                                  p(h2_hist), p(logits_hist), p(orders_hist), _lib.current_stream()))


def horizon_bwd(desc, state_hist, h1_hist, h2_hist, logits_hist, orders_hist, g_reward: Table, dz1, dz2, dz3):
    ops._dev(dz1)
    p = _lib.ptr
    _lib.check(_lib.lib().nic_horizon_rollout_bwd(desc, p(state_hist), p(h1_hist), p(h2_hist), p(logits_hist), p(orders_hist),
                                                  g_reward.t2(), p(dz1), p(dz2), p(dz3), _lib.current_stream()))
