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

constexpr int CF_MAXF = NIC_CF_MAX_STATE;  // state slots of one chain: Ws + Wn*Ww + E*We
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

template <int NP>
NIC_HD Dual<NP> cf_at(const Dual<NP> (&a)[CF_MAXF], int idx) {
    Dual<NP> r = dconst<NP>(0.f);
#pragma unroll
    for (int k = 0; k < CF_MAXF; ++k)
        if (k == idx) r = a[k];
    return r;
}

// sum of the slots of segment [o, o+W), left to right (x.sum(dim=2) over one pipeline, neural_networks.py:226)
template <int NP>
NIC_HD Dual<NP> cf_segment_sum(const Dual<NP> (&st)[CF_MAXF], int o, int W) {
    Dual<NP> r = dconst<NP>(0.f);
#pragma unroll
    for (int k = 0; k < CF_MAXF; ++k)
        if (k >= o && k < o + W) r = r + st[k];
    return r;
}

// new[o] = after + old[o+1]; new[k] = old[k+1]; new[o+W-1] = 0; new[o+L-1] += a if a != 0   (environment.py:405-432)
template <int NP>
NIC_HD void cf_segment_step(const Dual<NP> (&old)[CF_MAXF], Dual<NP> (&nw)[CF_MAXF], int o, int W, const Dual<NP>& after,
                            const Dual<NP>& a, float lead) {
    const int slot = o + (int)lead - 1;
#pragma unroll
    for (int k = 0; k < CF_MAXF; ++k) {
        if (k >= o && k < o + W) {
            Dual<NP> v = dconst<NP>(0.f);
            const Dual<NP> nxt = (k + 1 < CF_MAXF) ? old[(k + 1 < CF_MAXF) ? k + 1 : k] : dconst<NP>(0.f);
            if (k == o) v = after + nxt;
            else if (k < o + W - 1) v = nxt;
            if (a.v != 0.f && k == slot) v = v + a;  // zero orders are filtered out before the put (:426-429)
            nw[k] = v;
        }
    }
}

// one period of dynamics of the chain (environment.py:179-299 for S = 1, Wn <= 1); returns the period cost
template <int NP>
NIC_HD Dual<NP> cf_env_step(const NicClosedFormDesc& d, const CfStatics& c, const Dual<NP> (&st)[CF_MAXF],
                            Dual<NP> (&nx)[CF_MAXF], float dem, const CfOrders<NP>& o) {
#pragma unroll
    for (int k = 0; k < CF_MAXF; ++k) nx[k] = dconst<NP>(0.f);
    const Dual<NP> on_hand = st[0];
    Dual<NP> after = dsub(on_hand, dem);
    Dual<NP> cost;
    if (d.maximize_profit) cost = dscale(-c.p, dmin_const(on_hand, dem)) + dscale(c.h, drelu(after));  // :191-194
    else cost = dscale(c.p, drelu(dneg(after))) + dscale(c.h, drelu(after));                            // :198-201
    if (d.lost_demand) after = drelu(after);                                                            // :204-205
    cf_segment_step(st, nx, 0, d.Ws, after, o.store, c.lead);
    Dual<NP> total = cost;
    if (d.Wn) {
        const Dual<NP> w_after = cf_at(st, d.Ws) - o.store;  // ships what the store ordered (no clip, :249)
        Dual<NP> cw = dscale(c.wh_h, drelu(w_after));
        if (d.wh_edge.p) cw = cw + dscale(c.wh_edge, o.wh);
        cf_segment_step(st, nx, d.Ws, d.Ww, w_after, o.wh, c.wh_lead);
        total = total + cw;
    }
    if (d.E > 0) {
        Dual<NP> r_e = dconst<NP>(0.f);
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) {
            if (e < d.E) {
                const Dual<NP> ship = (e < d.E - 1) ? o.ech[(e + 1 < CF_MAXE) ? e + 1 : e] : o.wh;
                const int off = d.Ws + d.Ww + e * d.We;
                const Dual<NP> e_after = cf_at(st, off) - ship;
                r_e = r_e + dscale(c.e_h[e], drelu(e_after));
                cf_segment_step(st, nx, off, d.We, e_after, o.ech[e], c.e_lead[e]);
            }
        }
        total = total + r_e;
    }
    return total;
}

// orders of the closed-form policies from the current state and the levels
template <int NP>
NIC_HD CfOrders<NP> cf_policy(const NicClosedFormDesc& d, const Dual<NP> (&lv)[NIC_CF_MAX_LEVELS], const Dual<NP> (&st)[CF_MAXF]) {
    CfOrders<NP> o;
    o.store = o.wh = dconst<NP>(0.f);
#pragma unroll
    for (int e = 0; e < CF_MAXE; ++e) o.ech[e] = dconst<NP>(0.f);
    const Dual<NP> store_pos = cf_segment_sum(st, 0, d.Ws);
    if (d.policy == NIC_CF_BASE_STOCK) {           // clip(level - position, min=0)            neural_networks.py:227-229
        o.store = drelu(lv[0] - store_pos);
    } else if (d.policy == NIC_CF_CAPPED) {        // clip(level - position, min=0, max=cap)   :306-311
        o.store = dclamp0_cap(lv[0] - store_pos, lv[1]);
    } else {                                       // echelon base stock                       :247-288
        // locations ordered upstream -> downstream: k = 0..E-1 echelons, E warehouse, E+1 store; level k covers the
        // positions of locations k..E+1; allocation = min(clip(level_k - sum, 0), on-hand of the location upstream of k)
        Dual<NP> pos[CF_MAXE + 2];
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e)
            pos[e] = e < d.E ? cf_segment_sum(st, d.Ws + d.Ww + e * d.We, d.We) : dconst<NP>(0.f);
        const Dual<NP> wh_pos = cf_segment_sum(st, d.Ws, d.Ww);
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
                else if (k <= d.E) a = dmin(want, cf_at(st, d.Ws + d.Ww + (k - 1) * d.We));  // on hand of echelon k-1
                else a = dmin(want, cf_at(st, d.Ws));                                         // on hand of the warehouse
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

// Whole horizon of chain (store s, scenario b).  Outputs (each may be NULL):
//   reward_hist [T][S][ldb]   per-period cost of this chain
//   totals      [2][S][ldb]   sum over all periods / over periods >= ignore_periods
//   state_final [S][F][ldb]
//   g_levels    [NP] (returned through `g`): d(totals[0]) / d(level_j) of this chain
template <int NP>
NIC_HD void closed_form_chain(const NicClosedFormDesc& d, float* reward_hist, float* totals, float* state_final, int s,
                              int64_t b, float (&g)[NP > 0 ? NP : 1]) {
    const int64_t ldb = d.ldb;
    const int F = d.Ws + d.Wn * d.Ww + d.E * d.We;
    const CfStatics c = cf_load_statics(d, s, b);
    Dual<NP> lv[NIC_CF_MAX_LEVELS];
#pragma unroll
    for (int j = 0; j < NIC_CF_MAX_LEVELS; ++j) lv[j] = j < d.n_levels ? dparam<NP>(d.levels[j], j) : dconst<NP>(0.f);
    Dual<NP> st[CF_MAXF], nx[CF_MAXF];
#pragma unroll
    for (int k = 0; k < CF_MAXF; ++k) st[k] = dconst<NP>(k < F ? d.state0[((int64_t)s * F + k) * ldb + b] : 0.f);
    Dual<NP> total = dconst<NP>(0.f);
    float reported = 0.f;
    const float* dem_p = d.demand + ((int64_t)d.t0 * d.S + s) * ldb + b;
    const int64_t dem_stride = (int64_t)d.S * ldb;
    float dem = dem_p[0];
    for (int t = 0; t < d.T; ++t) {
        const float dem_next = t + 1 < d.T ? dem_p[(int64_t)(t + 1) * dem_stride] : 0.f;  // next period's demand in flight
        const CfOrders<NP> o = cf_policy(d, lv, st);
        const Dual<NP> r = cf_env_step(d, c, st, nx, dem, o);
        total = total + r;
        if (t >= d.ignore_periods) reported += r.v;
        if (reward_hist) reward_hist[((int64_t)t * d.S + s) * ldb + b] = r.v;
#pragma unroll
        for (int k = 0; k < CF_MAXF; ++k) st[k] = nx[k];
        dem = dem_next;
    }
    if (totals) {
        totals[(int64_t)s * ldb + b] = total.v;
        totals[((int64_t)d.S + s) * ldb + b] = reported;
    }
    if (state_final)
#pragma unroll
        for (int k = 0; k < CF_MAXF; ++k)
            if (k < F) state_final[((int64_t)s * F + k) * ldb + b] = st[k].v;
#pragma unroll
    for (int j = 0; j < NP; ++j) g[j] = total.d[j];
}

}  // namespace nic
