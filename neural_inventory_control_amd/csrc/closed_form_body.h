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

// sum of the live slots, left to right (x.sum(dim=2) over one pipeline, neural_networks.py:226).  INVARIANT of every pipeline
// here: slots at or past its length W hold exact zeros with zero tangents (cf_pipe_load writes them, cf_pipe_step shifts a zero
// in at the top and places orders below W only), so adding all MW register slots is the sum of the W live ones - same additions in
// the same order followed by "+ 0" - and no select on W is needed (round 6: the echelon chain's period was ~1,500 instructions,
// a third of them selects on the run-time lengths).
template <int NP, int MW>
NIC_HD Dual<NP> cf_pipe_sum(const CfPipe<NP, MW>& p, int /*W*/) {
    Dual<NP> r = p.s[0];
#pragma unroll
    for (int k = 1; k < MW; ++k) r = r + p.s[k];
    return r;
}

// The same two functions for a pipeline whose length WC is a compile-time constant and whose lead time is the same for every
// lane of the wavefront (a scenario-uniform lead-time table - every shipped setting): no select on W, and the order is placed by
// ONE wave-uniform branch instead of a compare + two selects per slot.  Same additions in the same order (round 5: the period loop
// of the 10^6-chain launch is bound by vector issue - 75 instructions per wave-period, a third of them these selects).
template <int NP, int MW, int WC>
NIC_HD Dual<NP> cf_pipe_sum_c(const CfPipe<NP, MW>& p) {
    Dual<NP> r = p.s[0];
#pragma unroll
    for (int k = 1; k < WC; ++k) r = r + p.s[k];
    return r;
}
template <int NP, int MW, int WC>
NIC_HD CfPipe<NP, MW> cf_pipe_step_c(const CfPipe<NP, MW>& old, const Dual<NP>& after, const Dual<NP>& a, int uniform_slot) {
    CfPipe<NP, MW> nw;
    const bool place = a.v != 0.f;  // zero orders are filtered out before the put (:426-429)
#pragma unroll
    for (int k = 0; k < MW; ++k) {
        Dual<NP> v = dconst<NP>(0.f);
        if (k + 1 < MW && k < WC - 1) v = old.s[k + 1 < MW ? k + 1 : k];
        if (k == 0) v = after + v;
        nw.s[k] = v;
    }
#pragma unroll
    for (int k = 0; k < MW; ++k)
        if (k == uniform_slot) {   // (uniform: a scalar branch)
            const Dual<NP> w = nw.s[k] + a;
            nw.s[k].v = place ? w.v : nw.s[k].v;
#pragma unroll
            for (int j = 0; j < NP; ++j) nw.s[k].d[j] = place ? w.d[j] : nw.s[k].d[j];
        }
    return nw;
}

// new[0] = after + old[1]; new[k] = old[k+1]; new[W-1] = 0; new[L-1] += a if a != 0   (environment.py:405-432)
// (the shift needs no W either: old[W] is an exact zero, so new[W - 1] = old[W] IS the reference's "new[W-1] = 0".  The order
// lands through m * a + v with m = 1 in its slot and 0 elsewhere: fma(1, a, v) = round(a + v), the reference's addition, and
// fma(0, a, v) = v for the finite orders of these policies - one multiply-add per value and tangent instead of an add + a select)
template <int NP, int MW>
NIC_HD CfPipe<NP, MW> cf_pipe_step(const CfPipe<NP, MW>& old, int /*W*/, const Dual<NP>& after, const Dual<NP>& a, float lead) {
    CfPipe<NP, MW> nw;
    const int slot = (a.v != 0.f) ? (int)lead - 1 : -1;  // zero orders are filtered out before the put (:426-429)
#pragma unroll
    for (int k = 0; k < MW; ++k) {
        Dual<NP> v = dconst<NP>(0.f);
        if (k + 1 < MW) v = old.s[k + 1 < MW ? k + 1 : k];
        if (k == 0) v = after + v;
        const float m = k == slot ? 1.f : 0.f;
        v.v = fmaf(m, a.v, v.v);
#pragma unroll
        for (int j = 0; j < NP; ++j) v.d[j] = fmaf(m, a.d[j], v.d[j]);
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
// FL: -1 = the setting's switches are read from the descriptor; 0..7 = compiled in (bit 0 round_orders, 1 lost_demand,
// 2 maximize_profit): the whole-horizon loop of the specialised variant is instantiated per combination, so that a wave-uniform
// switch costs one branch in front of the loop instead of a select per value and derivative in every period
template <int NP, int MW, bool CHAIN, int WC = 0, int FL = -1>
NIC_HD Dual<NP> cf_env_step(const NicClosedFormDesc& d, const CfStatics& c, CfState<NP, MW, CHAIN>& st, float dem,
                            const CfOrders<NP>& o, int uniform_slot = -2) {
    const bool maximize_profit = FL < 0 ? (bool)d.maximize_profit : (bool)(FL & 4);
    const bool lost_demand = FL < 0 ? (bool)d.lost_demand : (bool)(FL & 2);
    const int nE = (CHAIN && NP > 0) ? NP - 2 : d.E;   // training kernels of the chain: one level per location, E + 2 = NP, compiled in
    const Dual<NP> on_hand = st.store.s[0];
    Dual<NP> after = dsub(on_hand, dem);
    Dual<NP> cost;
    if (maximize_profit) cost = dscale(-c.p, dmin_const(on_hand, dem)) + dscale(c.h, drelu(after));  // :191-194
    else cost = dscale(c.p, drelu(dneg(after))) + dscale(c.h, drelu(after));                            // :198-201
    if (lost_demand) after = drelu(after);                                                            // :204-205
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
            if (e < nE) {
                const Dual<NP> ship = (e < nE - 1) ? o.ech[(e + 1 < CF_MAXE) ? e + 1 : e] : o.wh;
                const Dual<NP> e_after = st.ech[CHAIN ? e : 0].s[0] - ship;
                r_e = r_e + dscale(c.e_h[e], drelu(e_after));
                st.ech[CHAIN ? e : 0] = cf_pipe_step(st.ech[CHAIN ? e : 0], d.We, e_after, o.ech[e], c.e_lead[e]);
            }
        }
        total = total + r_e;
        st.wh = cf_pipe_step(st.wh, d.Ww, w_after, o.wh, c.wh_lead);
    }
    if (WC > 0) st.store = cf_pipe_step_c<NP, MW, (WC > 0 ? WC : 1)>(st.store, after, o.store, uniform_slot);
    else st.store = cf_pipe_step(st.store, d.Ws, after, o.store, c.lead);
    return total;
}

// orders of the closed-form policies from the current state and the levels
template <int NP, int MW, bool CHAIN, int WC = 0, int FL = -1>
NIC_HD CfOrders<NP> cf_policy(const NicClosedFormDesc& d, const Dual<NP> (&lv)[NIC_CF_MAX_LEVELS],
                              const CfState<NP, MW, CHAIN>& st) {
    CfOrders<NP> o;
    o.store = o.wh = dconst<NP>(0.f);
#pragma unroll
    for (int e = 0; e < CF_MAXE; ++e) o.ech[e] = dconst<NP>(0.f);
    const int nE = (CHAIN && NP > 0) ? NP - 2 : d.E;
    const Dual<NP> store_pos = WC > 0 ? cf_pipe_sum_c<NP, MW, (WC > 0 ? WC : 1)>(st.store) : cf_pipe_sum(st.store, d.Ws);
    if (!CHAIN && d.policy == NIC_CF_BASE_STOCK) {  // clip(level - position, min=0)            neural_networks.py:227-229
        o.store = drelu(lv[0] - store_pos);
    } else if (!CHAIN) {                           // clip(level - position, min=0, max=cap)   :306-311
        o.store = dclamp0_cap(lv[0] - store_pos, lv[1]);
    } else {                                       // echelon base stock                       :247-288
        // locations ordered upstream -> downstream: k = 0..E-1 echelons, E warehouse, E+1 store; level k covers the
        // positions of locations k..E+1; allocation = min(clip(level_k - sum, 0), on-hand of the location upstream of k)
        Dual<NP> pos[CF_MAXE];
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) pos[e] = e < nE ? cf_pipe_sum(st.ech[CHAIN ? e : 0], d.We) : dconst<NP>(0.f);
        const Dual<NP> wh_pos = cf_pipe_sum(st.wh, d.Ww);
#pragma unroll
        for (int k = 0; k < CF_MAXE + 2; ++k) {
            if (k < nE + 2) {
                // pos[:, k:].sum(dim=1), in location order (echelons k.., warehouse, store)
                Dual<NP> s = dconst<NP>(0.f);
                bool first = true;
#pragma unroll
                for (int e = 0; e < CF_MAXE; ++e)
                    if (e >= k && e < nE) {
                        s = first ? pos[e] : s + pos[e];
                        first = false;
                    }
                if (k <= nE) {
                    s = first ? wh_pos : s + wh_pos;
                    first = false;
                }
                s = first ? store_pos : s + store_pos;
                const Dual<NP> want = drelu(lv[k < NIC_CF_MAX_LEVELS ? k : 0] - s);
                Dual<NP> a;
                if (k == 0) a = dmin_const(want, 1000000.f);  // the outside supplier never binds (:262)
                else if (k <= nE) {                            // on hand of echelon k-1
                    Dual<NP> up = dconst<NP>(0.f);
#pragma unroll
                    for (int e = 0; e < CF_MAXE; ++e)
                        if (e == k - 1) up = st.ech[CHAIN ? e : 0].s[0];
                    a = dmin(want, up);
                } else a = dmin(want, st.wh.s[0]);             // on hand of the warehouse
                if (k < nE) o.ech[k < CF_MAXE ? k : 0] = a;
                else if (k == nE) o.wh = a;
                else o.store = a;
            }
        }
    }
    if (FL < 0 ? (bool)d.round_orders : (bool)(FL & 1)) {  // discrete allocation (trainer.py:201-202)
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
constexpr int kCfAhead = 8;   // periods of demand fetched per batch (one batch in flight beside the one being simulated)
// the period loop of a chain (see closed_form_chain), one instantiation per combination of the setting's switches where they
// are compiled in
template <int NP, int MW, bool CHAIN, int WC, int FL>
NIC_HD void cf_horizon(const NicClosedFormDesc& d, const CfStatics& c, const Dual<NP> (&lv)[NIC_CF_MAX_LEVELS],
                       CfState<NP, MW, CHAIN>& st, const float* dem_p, int64_t dem_stride, int uniform_slot, float* reward_hist,
                       int s, int64_t b, int64_t ldb, Dual<NP>& total, float& reported) {
    float cur[kCfAhead], nxt[kCfAhead];
#pragma unroll
    for (int j = 0; j < kCfAhead; ++j) cur[j] = dem_p[(int64_t)(j < d.T ? j : d.T - 1) * dem_stride];
    for (int t0 = 0; t0 < d.T; t0 += kCfAhead) {
#pragma unroll
        for (int j = 0; j < kCfAhead; ++j) {
            const int tn = t0 + kCfAhead + j;
            nxt[j] = dem_p[(int64_t)(tn < d.T ? tn : d.T - 1) * dem_stride];
        }
#pragma unroll
        for (int j = 0; j < kCfAhead; ++j) {
            const int t = t0 + j;
            if (t < d.T) {
                const CfOrders<NP> o = cf_policy<NP, MW, CHAIN, WC, FL>(d, lv, st);
                const Dual<NP> r = cf_env_step<NP, MW, CHAIN, WC, FL>(d, c, st, cur[j], o, uniform_slot);
                total = total + r;
                if (t >= d.ignore_periods) reported += r.v;
                if (reward_hist) reward_hist[((int64_t)t * d.S + s) * ldb + b] = r.v;
            }
        }
#pragma unroll
        for (int j = 0; j < kCfAhead; ++j) cur[j] = nxt[j];
    }
}

// WC > 0: the store pipeline has exactly WC slots and its lead time is wave-uniform (the caller checked both) - see cf_pipe_step_c
template <int NP, int MW, bool CHAIN, int WC = 0>
NIC_HD void closed_form_chain(const NicClosedFormDesc& d, float* reward_hist, float* totals, float* state_final, int s,
                              int64_t b, float (&g)[NP > 0 ? NP : 1], float* chain_sums = nullptr) {
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
    // The demand rows do not depend on the state: kCfAhead periods are fetched at a time, one batch ahead of the batch being
    // simulated (unconditional loads of a clamped row; round 5: with ONE period in flight the SQ counters showed the waves of the
    // 10^6-chain launch parked at s_waitcnt for 54 % of their cycles and the vector unit 58 % busy - the next row was issued only
    // ~75 instructions before its use).  Same arithmetic in the same order: results are bit-identical.
#if defined(__HIP_DEVICE_COMPILE__)
    const int uniform_slot = WC > 0 ? __builtin_amdgcn_readfirstlane((int)c.lead - 1) : -2;
#else
    const int uniform_slot = WC > 0 ? (int)c.lead - 1 : -2;
#endif
#define NIC_CF_RUN(FLV) cf_horizon<NP, MW, CHAIN, WC, FLV>(d, c, lv, st, dem_p, dem_stride, uniform_slot, reward_hist, s, b, ldb, total, reported)
    if (WC > 0) {
        switch ((d.round_orders ? 1 : 0) | (d.lost_demand ? 2 : 0) | (d.maximize_profit ? 4 : 0)) {
            case 0: NIC_CF_RUN(0); break;
            case 1: NIC_CF_RUN(1); break;
            case 2: NIC_CF_RUN(2); break;
            case 3: NIC_CF_RUN(3); break;
            case 4: NIC_CF_RUN(4); break;
            case 5: NIC_CF_RUN(5); break;
            case 6: NIC_CF_RUN(6); break;
            default: NIC_CF_RUN(7); break;
        }
    } else {
        NIC_CF_RUN(-1);
    }
#undef NIC_CF_RUN
    if (totals) {
        totals[(int64_t)s * ldb + b] = total.v;
        totals[((int64_t)d.S + s) * ldb + b] = reported;
    }
    if (chain_sums) {   // (the caller's registers: the kernel adds them over the wavefront)
        chain_sums[0] = total.v;
        chain_sums[1] = reported;
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
