// Whole-horizon rollout of the CLOSED-FORM policies (base_stock, capped_base_stock, echelon_stock — neural_networks.py:216-229,
// 296-311, 231-294 of the reference): one lane owns one scenario (one store chain) for all T periods, pipelines in registers,
// HBM sees the demand trace (4 B per store-period) and nothing else unless the caller asks for the per-period rewards.
//
// GRADIENTS BY FORWARD MODE.  These policies have 1 .. E+2 scalar parameters ("levels": the base-stock level, the cap, the
// echelon base-stock levels AFTER the reference's tiny `net` + softplus / cumsum, which stay in torch).  Reverse mode would need
// the state of every period (4*F B per scenario-period written and re-read, several times the demand trace); instead every
// state slot carries its tangent with respect to each level (a `Dual<NP>` = value + NP derivatives, all in registers) and the
// derivative of the scenario's total cost falls out of the same single pass.  The local derivative of every primitive is
// exactly the one torch's autograd uses in the reference (clamp passes where x >= min, clamp with a tensor max hands the
// gradient to the max where x > max, minimum splits ties 0.5 / 0.5, an order that is exactly 0 has no placement derivative —
// environment.py:426-429), and forward mode multiplies the same Jacobians in the other order, so the result is the
// reference's gradient up to float rounding.
//
// NIC_HD: the same body is compiled for the host by tests/hostsim and checked against the reference's golden vectors.
#pragma once
#include <math.h>

#include "env_step_body.h"

namespace nic {

constexpr int CF_MAXE = 3;

template <int NP>
struct Dual {
    float v;
    float d[NP > 0 ? NP : 1];
};

template <int NP>
NIC_HD Dual<NP> dconst(float v) {
    Dual<NP> r;
    r.v = v;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = 0.f;
    return r;
}
template <int NP>
NIC_HD Dual<NP> dparam(float v, int which) {  // d(level_which)/d(level_j) = [j == which]
    Dual<NP> r = dconst<NP>(v);
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = j == which ? 1.f : 0.f;
    return r;
}
template <int NP>
NIC_HD Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = a.d[j] + b.d[j];
    return r;
}
template <int NP>
NIC_HD Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = a.d[j] - b.d[j];
    return r;
}
template <int NP>
NIC_HD Dual<NP> dsub(const Dual<NP>& a, float c) {
    Dual<NP> r = a;
    r.v = a.v - c;
    return r;
}
template <int NP>
NIC_HD Dual<NP> dneg(const Dual<NP>& a) {
    Dual<NP> r;
    r.v = -a.v;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = -a.d[j];
    return r;
}
template <int NP>
NIC_HD Dual<NP> dscale(float c, const Dual<NP>& a) {
    Dual<NP> r;
    r.v = c * a.v;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = c * a.d[j];
    return r;
}
// torch.clip(x, min=0): value max(x, 0); clamp's backward passes the gradient where x >= min
template <int NP>
NIC_HD Dual<NP> drelu(const Dual<NP>& a) {
    Dual<NP> r;
    r.v = a.v > 0.f ? a.v : 0.f;
    const float m = a.v >= 0.f ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = m * a.d[j];
    return r;
}
// torch.clip(x, min=0 (constant tensor), max=cap (tensor)): clamp.Tensor — self gets the gradient where min <= x <= max, max
// where x > max (or max < min)
template <int NP>
NIC_HD Dual<NP> dclamp0_cap(const Dual<NP>& x, const Dual<NP>& cap) {
    Dual<NP> r;
    const float lo = x.v > 0.f ? x.v : 0.f;
    r.v = lo < cap.v ? lo : cap.v;
    const float mx = (x.v >= 0.f && x.v <= cap.v) ? 1.f : 0.f;
    const float mc = (x.v > cap.v || cap.v < 0.f) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = mx * x.d[j] + mc * cap.d[j];
    return r;
}
// torch.minimum(a, b): the smaller operand gets the gradient, a tie splits it 0.5 / 0.5
template <int NP>
NIC_HD Dual<NP> dmin(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v < b.v ? a.v : b.v;
    const float wa = a.v < b.v ? 1.f : (a.v == b.v ? 0.5f : 0.f);
    const float wb = 1.f - wa;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = wa * a.d[j] + wb * b.d[j];
    return r;
}
template <int NP>
NIC_HD Dual<NP> dmin_const(const Dual<NP>& a, float c) {  // minimum(a, constant)
    Dual<NP> r;
    r.v = a.v < c ? a.v : c;
    const float wa = a.v < c ? 1.f : (a.v == c ? 0.5f : 0.f);
#pragma unroll
    for (int j = 0; j < NP; ++j) r.d[j] = wa * a.d[j];
    return r;
}
template <int NP>
NIC_HD Dual<NP> dround(const Dual<NP>& a) {  // torch.round: half to even, zero gradient (trainer.py:201-202)
    return dconst<NP>(rintf(a.v));
}

struct CfStatics {  // per-scenario constants of one chain, loaded once
    float p, h, lead;
    float wh_h, wh_lead, wh_edge;
    float e_h[CF_MAXE], e_lead[CF_MAXE];
};

NIC_HD CfStatics cf_load_statics(const NicClosedFormDesc& d, int s, int64_t b) {
    CfStatics c;
    c.p = t2(d.underage, s, b);
    c.h = t2(d.holding, s, b);
    c.lead = t2(d.lead, s, b);
    c.wh_h = d.Wn ? t2(d.wh_holding, 0, b) : 0.f;
    c.wh_lead = d.Wn ? t2(d.wh_lead, 0, b) : 0.f;
    c.wh_edge = (d.Wn && d.wh_edge.p) ? t2(d.wh_edge, 0, b) : 0.f;
#pragma unroll
    for (int e = 0; e < CF_MAXE; ++e) {
        c.e_h[e] = e < d.E ? t2(d.ech_holding, e, b) : 0.f;
        c.e_lead[e] = e < d.E ? t2(d.ech_lead, e, b) : 0.f;
    }
    return c;
}

template <int NP>
struct CfOrders {
    Dual<NP> store, wh, ech[CF_MAXE];
};

// One pipeline = MW register slots (slot 0 = on hand); W <= MW of them are live.  All slot indices below are compile-time
// after unrolling; only the comparison with the (per-scenario) lead time and with W is data.
template <int NP, int MW>
struct CfPipe {
    Dual<NP> s[MW];
};

// sum of the live slots, left to right (x.sum(dim=2) over one pipeline, neural_networks.py:226)
template <int NP, int MW>
NIC_HD Dual<NP> cf_pipe_sum(const CfPipe<NP, MW>& p, int W) {
    Dual<NP> r = p.s[0];
#pragma unroll
    for (int k = 1; k < MW; ++k)
        if (k < W) r = r + p.s[k];
    return r;
}

// new[0] = after + old[1]; new[k] = old[k+1]; new[W-1] = 0; new[L-1] += a if a != 0   (environment.py:405-432)
template <int NP, int MW>
NIC_HD CfPipe<NP, MW> cf_pipe_step(const CfPipe<NP, MW>& old, int W, const Dual<NP>& after, const Dual<NP>& a, float lead) {
    CfPipe<NP, MW> nw;
    const int slot = (int)lead - 1;
    const bool place = a.v != 0.f;  // zero orders are filtered out before the put (:426-429)
#pragma unroll
    for (int k = 0; k < MW; ++k) {
        Dual<NP> v = dconst<NP>(0.f);
        if (k + 1 < MW && k < W - 1) v = old.s[k + 1 < MW ? k + 1 : k];
        if (k == 0) v = after + v;
        if (place && k == slot) v = v + a;
        nw.s[k] = v;
    }
    return nw;
}

// state of one chain: the store's pipeline and, for CHAIN, the warehouse's and up to CF_MAXE echelons'
template <int NP, int MW, bool CHAIN>
struct CfState {
    CfPipe<NP, MW> store;
    CfPipe<NP, MW> wh;
    CfPipe<NP, MW> ech[CHAIN ? CF_MAXE : 1];
};

// one period of dynamics of the chain (environment.py:179-299 for S = 1, Wn <= 1); returns the period cost.
// CHAIN = false compiles the single-store form only (Wn = E = 0 known at compile time: base_stock / capped_base_stock)
template <int NP, int MW, bool CHAIN>
NIC_HD Dual<NP> cf_env_step(const NicClosedFormDesc& d, const CfStatics& c, CfState<NP, MW, CHAIN>& st, float dem,
                            const CfOrders<NP>& o) {
    const Dual<NP> on_hand = st.store.s[0];
    Dual<NP> after = dsub(on_hand, dem);
    Dual<NP> cost;
    if (d.maximize_profit) cost = dscale(-c.p, dmin_const(on_hand, dem)) + dscale(c.h, drelu(after));  // :191-194
    else cost = dscale(c.p, drelu(dneg(after))) + dscale(c.h, drelu(after));                            // :198-201
    if (d.lost_demand) after = drelu(after);                                                            // :204-205
    Dual<NP> total = cost;
    if (CHAIN) {
        // the warehouse ships what the store ordered (no clip, :249); echelon e ships what its downstream neighbour ordered
        const Dual<NP> w_after = st.wh.s[0] - o.store;
        Dual<NP> cw = dscale(c.wh_h, drelu(w_after));
        if (d.wh_edge.p) cw = cw + dscale(c.wh_edge, o.wh);
        total = total + cw;
        Dual<NP> r_e = dconst<NP>(0.f);
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) {
            if (e < d.E) {
                const Dual<NP> ship = (e < d.E - 1) ? o.ech[(e + 1 < CF_MAXE) ? e + 1 : e] : o.wh;
                const Dual<NP> e_after = st.ech[CHAIN ? e : 0].s[0] - ship;
                r_e = r_e + dscale(c.e_h[e], drelu(e_after));
                st.ech[CHAIN ? e : 0] = cf_pipe_step(st.ech[CHAIN ? e : 0], d.We, e_after, o.ech[e], c.e_lead[e]);
            }
        }
        total = total + r_e;
        st.wh = cf_pipe_step(st.wh, d.Ww, w_after, o.wh, c.wh_lead);
    }
    st.store = cf_pipe_step(st.store, d.Ws, after, o.store, c.lead);
    return total;
}

// orders of the closed-form policies from the current state and the levels
template <int NP, int MW, bool CHAIN>
NIC_HD CfOrders<NP> cf_policy(const NicClosedFormDesc& d, const Dual<NP> (&lv)[NIC_CF_MAX_LEVELS],
                              const CfState<NP, MW, CHAIN>& st) {
    CfOrders<NP> o;
    o.store = o.wh = dconst<NP>(0.f);
#pragma unroll
    for (int e = 0; e < CF_MAXE; ++e) o.ech[e] = dconst<NP>(0.f);
    const Dual<NP> store_pos = cf_pipe_sum(st.store, d.Ws);
    if (!CHAIN && d.policy == NIC_CF_BASE_STOCK) {  // clip(level - position, min=0)            neural_networks.py:227-229
        o.store = drelu(lv[0] - store_pos);
    } else if (!CHAIN) {                           // clip(level - position, min=0, max=cap)   :306-311
        o.store = dclamp0_cap(lv[0] - store_pos, lv[1]);
    } else {                                       // echelon base stock                       :247-288
        // locations ordered upstream -> downstream: k = 0..E-1 echelons, E warehouse, E+1 store; level k covers the
        // positions of locations k..E+1; allocation = min(clip(level_k - sum, 0), on-hand of the location upstream of k)
        Dual<NP> pos[CF_MAXE];
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) pos[e] = e < d.E ? cf_pipe_sum(st.ech[CHAIN ? e : 0], d.We) : dconst<NP>(0.f);
        const Dual<NP> wh_pos = cf_pipe_sum(st.wh, d.Ww);
#pragma unroll
        for (int k = 0; k < CF_MAXE + 2; ++k) {
            if (k < d.E + 2) {
                // pos[:, k:].sum(dim=1), in location order (echelons k.., warehouse, store)
                Dual<NP> s = dconst<NP>(0.f);
                bool first = true;
#pragma unroll
                for (int e = 0; e < CF_MAXE; ++e)
                    if (e >= k && e < d.E) {
                        s = first ? pos[e] : s + pos[e];
                        first = false;
                    }
                if (k <= d.E) {
                    s = first ? wh_pos : s + wh_pos;
                    first = false;
                }
                s = first ? store_pos : s + store_pos;
                const Dual<NP> want = drelu(lv[k < NIC_CF_MAX_LEVELS ? k : 0] - s);
                Dual<NP> a;
                if (k == 0) a = dmin_const(want, 1000000.f);  // the outside supplier never binds (:262)
                else if (k <= d.E) {                           // on hand of echelon k-1
                    Dual<NP> up = dconst<NP>(0.f);
#pragma unroll
                    for (int e = 0; e < CF_MAXE; ++e)
                        if (e == k - 1) up = st.ech[CHAIN ? e : 0].s[0];
                    a = dmin(want, up);
                } else a = dmin(want, st.wh.s[0]);             // on hand of the warehouse
                if (k < d.E) o.ech[k < CF_MAXE ? k : 0] = a;
                else if (k == d.E) o.wh = a;
                else o.store = a;
            }
        }
    }
    if (d.round_orders) {  // discrete allocation (trainer.py:201-202)
        o.store = dround(o.store);
        o.wh = dround(o.wh);
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) o.ech[e] = dround(o.ech[e]);
    }
    return o;
}

template <int NP, int MW>
NIC_HD void cf_pipe_load(CfPipe<NP, MW>& p, const float* src, int W, int64_t ldb) {
#pragma unroll
    for (int k = 0; k < MW; ++k) p.s[k] = dconst<NP>(k < W ? src[(int64_t)k * ldb] : 0.f);
}
template <int NP, int MW>
NIC_HD void cf_pipe_store(const CfPipe<NP, MW>& p, float* dst, int W, int64_t ldb) {
#pragma unroll
    for (int k = 0; k < MW; ++k)
        if (k < W) dst[(int64_t)k * ldb] = p.s[k].v;
}

// Whole horizon of chain (store s, scenario b).  MW = register slots per pipeline (>= every live pipeline length).
// Outputs (each may be NULL):
//   reward_hist [T][S][ldb]   per-period cost of this chain
//   totals      [2][S][ldb]   sum over all periods / over periods >= ignore_periods
//   state_final [S][F][ldb]
//   g_levels    [NP] (returned through `g`): d(totals[0]) / d(level_j) of this chain
template <int NP, int MW, bool CHAIN>
NIC_HD void closed_form_chain(const NicClosedFormDesc& d, float* reward_hist, float* totals, float* state_final, int s,
                              int64_t b, float (&g)[NP > 0 ? NP : 1]) {
    const int64_t ldb = d.ldb;
    const int F = d.Ws + d.Wn * d.Ww + d.E * d.We;
    const CfStatics c = cf_load_statics(d, s, b);
    Dual<NP> lv[NIC_CF_MAX_LEVELS];
#pragma unroll
    for (int j = 0; j < NIC_CF_MAX_LEVELS; ++j) lv[j] = j < d.n_levels ? dparam<NP>(d.levels[j], j) : dconst<NP>(0.f);
    CfState<NP, MW, CHAIN> st;
    const float* s0 = d.state0 + (int64_t)s * F * ldb + b;
    cf_pipe_load(st.store, s0, d.Ws, ldb);
    cf_pipe_load(st.wh, s0 + (int64_t)d.Ws * ldb, CHAIN ? d.Ww : 0, ldb);
#pragma unroll
    for (int e = 0; e < (CHAIN ? CF_MAXE : 1); ++e)
        cf_pipe_load(st.ech[e], s0 + (int64_t)(d.Ws + d.Ww + e * d.We) * ldb, (CHAIN && e < d.E) ? d.We : 0, ldb);
    Dual<NP> total = dconst<NP>(0.f);
    float reported = 0.f;
    const float* dem_p = d.demand + ((int64_t)d.t0 * d.S + s) * ldb + b;
    const int64_t dem_stride = (int64_t)d.S * ldb;
    float dem = dem_p[0];
    for (int t = 0; t < d.T; ++t) {
        const float dem_next = t + 1 < d.T ? dem_p[(int64_t)(t + 1) * dem_stride] : 0.f;  // next period's demand in flight
        const CfOrders<NP> o = cf_policy<NP, MW, CHAIN>(d, lv, st);
        const Dual<NP> r = cf_env_step<NP, MW, CHAIN>(d, c, st, dem, o);
        total = total + r;
        if (t >= d.ignore_periods) reported += r.v;
        if (reward_hist) reward_hist[((int64_t)t * d.S + s) * ldb + b] = r.v;
        dem = dem_next;
    }
    if (totals) {
        totals[(int64_t)s * ldb + b] = total.v;
        totals[((int64_t)d.S + s) * ldb + b] = reported;
    }
    if (state_final) {
        float* f0 = state_final + (int64_t)s * F * ldb + b;
        cf_pipe_store(st.store, f0, d.Ws, ldb);
        if (CHAIN) {
            cf_pipe_store(st.wh, f0 + (int64_t)d.Ws * ldb, d.Ww, ldb);
#pragma unroll
            for (int e = 0; e < CF_MAXE; ++e)
                if (e < d.E) cf_pipe_store(st.ech[CHAIN ? e : 0], f0 + (int64_t)(d.Ws + d.Ww + e * d.We) * ldb, d.We, ldb);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) g[j] = total.d[j];
}

}  // namespace nic
