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

// Tangents are held as PAIRS (two floats in an aligned register pair) plus, for an odd NP, one single float: every tangent
// operation below is one packed instruction per pair (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 - the chain's period is bound by
// vector issue) by construction, not by the vectoriser's choice (round 6: left to itself it also packed unrelated scalars and paid
// 2-4 register moves per packed add).
typedef float CfPair __attribute__((vector_size(8)));

NIC_HD CfPair cf_pair(float x) {
    CfPair r = {x, x};
    return r;
}
NIC_HD CfPair cf_pair_fma(float m, CfPair a, CfPair v) {   // fma(m, a, v) per lane (one rounding)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(cf_pair(m), a, v);
#else
    CfPair r = {fmaf(m, a[0], v[0]), fmaf(m, a[1], v[1])};
    return r;
#endif
}

template <int NP>
struct Tan {   // NP derivatives
    static constexpr int kPairs = NP / 2;
    static constexpr bool kOdd = (NP & 1) != 0;
    CfPair p[kPairs > 0 ? kPairs : 1];
    float t;   // the last derivative of an odd NP
    NIC_HD float get(int j) const { return (kOdd && j == NP - 1) ? t : p[j >> 1][j & 1]; }
};
#define NIC_CF_TAN_OP(EXPR_PAIR, EXPR_TAIL)                            \
    Tan<NP> r;                                                         \
    _Pragma("unroll") for (int i = 0; i < Tan<NP>::kPairs; ++i) r.p[i] = (EXPR_PAIR); \
    if (Tan<NP>::kPairs == 0) r.p[0] = cf_pair(0.f);                   \
    r.t = Tan<NP>::kOdd ? (EXPR_TAIL) : 0.f;                           \
    return r
template <int NP>
NIC_HD Tan<NP> tan_zero() { NIC_CF_TAN_OP(cf_pair(0.f), 0.f); }
template <int NP>
NIC_HD Tan<NP> tan_unit(int which) {
    NIC_CF_TAN_OP((CfPair{2 * i == which ? 1.f : 0.f, 2 * i + 1 == which ? 1.f : 0.f}), which == NP - 1 ? 1.f : 0.f);
}
template <int NP>
NIC_HD Tan<NP> operator+(const Tan<NP>& a, const Tan<NP>& b) { NIC_CF_TAN_OP(a.p[i] + b.p[i], a.t + b.t); }
template <int NP>
NIC_HD Tan<NP> operator-(const Tan<NP>& a, const Tan<NP>& b) { NIC_CF_TAN_OP(a.p[i] - b.p[i], a.t - b.t); }
template <int NP>
NIC_HD Tan<NP> operator-(const Tan<NP>& a) { NIC_CF_TAN_OP(-a.p[i], -a.t); }
template <int NP>
NIC_HD Tan<NP> operator*(float c, const Tan<NP>& a) { NIC_CF_TAN_OP(cf_pair(c) * a.p[i], c * a.t); }
template <int NP>
NIC_HD Tan<NP> tan_fma(float m, const Tan<NP>& a, const Tan<NP>& v) {   // fma(m, a, v) per derivative (one rounding)
    NIC_CF_TAN_OP(cf_pair_fma(m, a.p[i], v.p[i]), fmaf(m, a.t, v.t));
}
template <int NP>
NIC_HD Tan<NP> tan_select(bool c, const Tan<NP>& a, const Tan<NP>& b) {
    NIC_CF_TAN_OP((CfPair{c ? a.p[i][0] : b.p[i][0], c ? a.p[i][1] : b.p[i][1]}), c ? a.t : b.t);
}
#undef NIC_CF_TAN_OP

template <int NP>
struct Dual {
    float v;
    Tan<NP> d;
};

template <int NP>
NIC_HD Dual<NP> dconst(float v) {
    Dual<NP> r;
    r.v = v;
    r.d = tan_zero<NP>();
    return r;
}
template <int NP>
NIC_HD Dual<NP> dparam(float v, int which) {  // d(level_which)/d(level_j) = [j == which]
    Dual<NP> r;
    r.v = v;
    r.d = tan_unit<NP>(which);
    return r;
}
template <int NP>
NIC_HD Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v + b.v;
    r.d = a.d + b.d;
    return r;
}
template <int NP>
NIC_HD Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v - b.v;
    r.d = a.d - b.d;
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
    r.d = -a.d;
    return r;
}
template <int NP>
NIC_HD Dual<NP> dscale(float c, const Dual<NP>& a) {
    Dual<NP> r;
    r.v = c * a.v;
    r.d = c * a.d;
    return r;
}
// torch.clip(x, min=0): value max(x, 0); clamp's backward passes the gradient where x >= min
template <int NP>
NIC_HD Dual<NP> drelu(const Dual<NP>& a) {
    Dual<NP> r;
    r.v = a.v > 0.f ? a.v : 0.f;
    r.d = (a.v >= 0.f ? 1.f : 0.f) * a.d;
    return r;
}
// torch.clip(x, min=0 (constant tensor), max=cap (tensor)): clamp.Tensor — self gets the gradient where min <= x <= max, max
// where x > max (or max < min).  (Tangents: the second product is added by a fused multiply-add - every VALUE in this file keeps
// the reference's separate multiply and add; at most one of the two factors is not 0 here, so the sum is exact either way)
template <int NP>
NIC_HD Dual<NP> dclamp0_cap(const Dual<NP>& x, const Dual<NP>& cap) {
    Dual<NP> r;
    const float lo = x.v > 0.f ? x.v : 0.f;
    r.v = lo < cap.v ? lo : cap.v;
    const float mx = (x.v >= 0.f && x.v <= cap.v) ? 1.f : 0.f;
    const float mc = (x.v > cap.v || cap.v < 0.f) ? 1.f : 0.f;
    r.d = tan_fma(mx, x.d, mc * cap.d);
    return r;
}
// torch.minimum(a, b): the smaller operand gets the gradient, a tie splits it 0.5 / 0.5
template <int NP>
NIC_HD Dual<NP> dmin(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    r.v = a.v < b.v ? a.v : b.v;
    const float wa = a.v < b.v ? 1.f : (a.v == b.v ? 0.5f : 0.f);
    r.d = tan_fma(wa, a.d, (1.f - wa) * b.d);
    return r;
}
template <int NP>
NIC_HD Dual<NP> dmin_const(const Dual<NP>& a, float c) {  // minimum(a, constant)
    Dual<NP> r;
    r.v = a.v < c ? a.v : c;
    r.d = (a.v < c ? 1.f : (a.v == c ? 0.5f : 0.f)) * a.d;
    return r;
}
// c * clip(a, min=0) and minimum(clip(x, min=0), up) with the clip's 0 / 1 mask folded into the factor of the tangents:
// c * (m * d) = (c * m) * d and wa * (m * d) = (wa * m) * d exactly (m is 0 or 1, wa is 0, 0.5 or 1), one packed multiply less each
// (round 6)
template <int NP>
NIC_HD Dual<NP> dscale_relu(float c, const Dual<NP>& a) {
    Dual<NP> r;
    r.v = c * (a.v > 0.f ? a.v : 0.f);
    r.d = (c * (a.v >= 0.f ? 1.f : 0.f)) * a.d;
    return r;
}
// acc + c * clip(a, min=0): the tangent's multiply and add are one fused multiply-add per pair (tangents only)
template <int NP>
NIC_HD Dual<NP> dadd_scale_relu(const Dual<NP>& acc, float c, const Dual<NP>& a) {
    Dual<NP> r;
    r.v = acc.v + c * (a.v > 0.f ? a.v : 0.f);
    r.d = tan_fma(c * (a.v >= 0.f ? 1.f : 0.f), a.d, acc.d);
    return r;
}
// p * clip(-a, min=0) + h * clip(a, min=0) (underage + holding cost of what is left after demand, environment.py:198-201): the
// two clips pass the gradient on opposite sides of 0 (both at exactly 0), so the tangent is (h [a >= 0] - p [a <= 0]) * a'
template <int NP>
NIC_HD Dual<NP> dcost_backlog(float p, float h, const Dual<NP>& a) {
    Dual<NP> r;
    const float na = -a.v;
    r.v = p * (na > 0.f ? na : 0.f) + h * (a.v > 0.f ? a.v : 0.f);
    r.d = (h * (a.v >= 0.f ? 1.f : 0.f) - p * (na >= 0.f ? 1.f : 0.f)) * a.d;
    return r;
}
template <int NP>
NIC_HD Dual<NP> dmin_relu(const Dual<NP>& x, const Dual<NP>& b) {
    Dual<NP> r;
    const float want = x.v > 0.f ? x.v : 0.f;
    r.v = want < b.v ? want : b.v;
    const float wa = want < b.v ? 1.f : (want == b.v ? 0.5f : 0.f);
    r.d = tan_fma(wa * (x.v >= 0.f ? 1.f : 0.f), x.d, (1.f - wa) * b.d);
    return r;
}
template <int NP>
NIC_HD Dual<NP> dmin_relu_const(const Dual<NP>& x, float c) {
    Dual<NP> r;
    const float want = x.v > 0.f ? x.v : 0.f;
    r.v = want < c ? want : c;
    r.d = ((want < c ? 1.f : (want == c ? 0.5f : 0.f)) * (x.v >= 0.f ? 1.f : 0.f)) * x.d;
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
    float hot[MW];   // 1 in the slot an order placed now lands in (lead - 1), else 0: fixed for the horizon (cf_pipe_set_lead)
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
        nw.hot[k] = old.hot[k];
    }
#pragma unroll
    for (int k = 0; k < MW; ++k)
        if (k == uniform_slot) {   // (uniform: a scalar branch)
            const Dual<NP> w = nw.s[k] + a;
            nw.s[k].v = place ? w.v : nw.s[k].v;
            nw.s[k].d = tan_select(place, w.d, nw.s[k].d);
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
    // the slot's 0 / 1 factor = [the order is not 0] x [slot == lead - 1], both as floats in vector registers (round 6: as lane
    // masks the second one lived in 32 scalar registers, and each slot's factor took a scalar AND between two vector
    // instructions - a round trip through the scalar unit in the middle of the period's dependency chain)
    const float placed = (a.v != 0.f) ? 1.f : 0.f;  // zero orders are filtered out before the put (:426-429)
    (void)lead;
#pragma unroll
    for (int k = 0; k < MW; ++k) {
        Dual<NP> v = dconst<NP>(0.f);
        if (k + 1 < MW) v = old.s[k + 1 < MW ? k + 1 : k];
        if (k == 0) v = after + v;
        const float m = placed * old.hot[k];
        v.v = fmaf(m, a.v, v.v);
        v.d = tan_fma(m, a.d, v.d);
        nw.s[k] = v;
        nw.hot[k] = old.hot[k];
    }
    return nw;
}
template <int NP, int MW>
NIC_HD void cf_pipe_set_lead(CfPipe<NP, MW>& p, float lead) {
#pragma unroll
    for (int k = 0; k < MW; ++k) {
        float h = k == (int)lead - 1 ? 1.f : 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(h));   // (kept a number: the optimiser would otherwise fold the product above back into lane masks)
#endif
        p.hot[k] = h;
    }
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
    if (maximize_profit) cost = dscale(-c.p, dmin_const(on_hand, dem)) + dscale_relu(c.h, after);    // :191-194
    else cost = dcost_backlog(c.p, c.h, after);                                                         // :198-201
    if (lost_demand) after = drelu(after);                                                            // :204-205
    Dual<NP> total = cost;
    if (CHAIN) {
        // the warehouse ships what the store ordered (no clip, :249); echelon e ships what its downstream neighbour ordered
        const Dual<NP> w_after = st.wh.s[0] - o.store;
        if (d.wh_edge.p) total = total + (dscale_relu(c.wh_h, w_after) + dscale(c.wh_edge, o.wh));
        else total = dadd_scale_relu(total, c.wh_h, w_after);
        Dual<NP> r_e = dconst<NP>(0.f);
#pragma unroll
        for (int e = 0; e < CF_MAXE; ++e) {
            if (e < nE) {
                const Dual<NP> ship = (e < nE - 1) ? o.ech[(e + 1 < CF_MAXE) ? e + 1 : e] : o.wh;
                const Dual<NP> e_after = st.ech[CHAIN ? e : 0].s[0] - ship;
                r_e = dadd_scale_relu(r_e, c.e_h[e], e_after);
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
        // tangents of the sums pos[:, k:].sum(dim=1) for every k at once, from the store upwards (each is the next one plus one
        // location: E + 1 packed additions per pair instead of (E + 1)(E + 2) / 2; the VALUES below are summed per k in the
        // reference's order)
        Dual<NP> suf[CF_MAXE + 2];
        suf[CF_MAXE + 1] = store_pos;
        suf[CF_MAXE] = wh_pos + store_pos;
#pragma unroll
        for (int e = CF_MAXE - 1; e >= 0; --e) suf[e] = e < nE ? pos[e] + suf[e + 1] : suf[e + 1];
#pragma unroll
        for (int k = 0; k < CF_MAXE + 2; ++k) {
            if (k < nE + 2) {
                // pos[:, k:].sum(dim=1), in location order (echelons k.., warehouse, store)
                float sv = 0.f;
                bool first = true;
#pragma unroll
                for (int e = 0; e < CF_MAXE; ++e)
                    if (e >= k && e < nE) {
                        sv = first ? pos[e].v : sv + pos[e].v;
                        first = false;
                    }
                if (k <= nE) {
                    sv = first ? wh_pos.v : sv + wh_pos.v;
                    first = false;
                }
                sv = first ? store_pos.v : sv + store_pos.v;
                // (location k's suffix: echelon k for k < nE, the warehouse for k == nE, the store for k == nE + 1)
                Dual<NP> s = k < nE ? suf[k < CF_MAXE ? k : 0] : (k == nE ? suf[CF_MAXE] : suf[CF_MAXE + 1]);
                s.v = sv;
                const Dual<NP> want = lv[k < NIC_CF_MAX_LEVELS ? k : 0] - s;   // (its clip(min=0) is inside the minimum below)
                Dual<NP> a;
                if (k == 0) a = dmin_relu_const(want, 1000000.f);  // the outside supplier never binds (:262)
                else if (k <= nE) {                            // on hand of echelon k-1
                    Dual<NP> up = dconst<NP>(0.f);
#pragma unroll
                    for (int e = 0; e < CF_MAXE; ++e)
                        if (e == k - 1) up = st.ech[CHAIN ? e : 0].s[0];
                    a = dmin_relu(want, up);
                } else a = dmin_relu(want, st.wh.s[0]);        // on hand of the warehouse
                if (k < nE) o.ech[k < CF_MAXE ? k : 0] = a;
                else if (k == nE) o.wh = a;
                else o.store = a;
            }
        }
    }
    // (a launch with tangents never rounds - the entry point refuses the combination: rounded orders have zero gradient)
    if (NP == 0 && (FL < 0 ? (bool)d.round_orders : (bool)(FL & 1))) {  // discrete allocation (trainer.py:201-202)
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
    cf_pipe_set_lead(st.store, c.lead);
    cf_pipe_set_lead(st.wh, c.wh_lead);
#pragma unroll
    for (int e = 0; e < (CHAIN ? CF_MAXE : 1); ++e) cf_pipe_set_lead(st.ech[e], c.e_lead[e < CF_MAXE ? e : 0]);
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
    for (int j = 0; j < NP; ++j) g[j] = total.d.get(j);
}

}  // namespace nic
