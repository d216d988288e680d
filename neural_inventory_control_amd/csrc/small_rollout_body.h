// Whole-horizon rollout of the SMALL policies (one store, optional warehouse + echelons, 32-wide MLP): one lane owns one
// scenario for all T periods.  Pipeline slots, hidden activations and gradients live in REGISTERS across the unrolled
// horizon; HBM sees only the demand trace (4 B per scenario-period), the per-period reward and — when training — the
// activations the backward sweep needs.  This replaces T x (4-5 GEMM launches + head + env step) by ONE kernel for
// vanilla_one_store (neural_networks.py:195-214) and vanilla_serial (:314-355) on one_store_* / serial_system settings,
// whose 32 x 32 layers are far too small to fill a launch (cfg 1, 2, 4 are launch-bound in the reference too).
//
// MLP weights are wave-uniform: they are read through the scalar path (s_load) straight into FMA operands.
// NIC_HD: the same body runs in tests/hostsim on the CPU.  Arithmetic of the dynamics = env_step_body.h for S = 1
// (a sum over one store / one warehouse is exact, so there is no summation-order issue here).
#pragma once
#include <math.h>

#include "env_step_body.h"
#include "policy_heads_body.h"

namespace nic {

constexpr int SR_MAXF = NIC_SR_MAX_INPUTS;   // state slots = MLP inputs
constexpr int SR_H = NIC_SR_HIDDEN;          // hidden width
constexpr int SR_MAXOUT = NIC_SR_MAX_OUTPUTS;
constexpr int SR_MAXE = 3;

// ELU negative branch.  libm's expm1f costs several hundred cycles per call on the device and there are 32 * n_hidden of
// them per scenario-period, more than the FMAs of the layers themselves; this form is a 6-term series near zero (relative
// error < 2e-7 for x > -0.35) and exp(x) - 1 on the hardware exponential below that (value <= -0.29, absolute error ~1e-7).
// (both forms are evaluated and one is selected: as an `if` this was a divergent branch per activation - save / restore of the
// exec mask and a scalar branch around ~8 vector instructions, 48 times per scenario-period in the whole-horizon kernels, with
// both sides executed anyway whenever the lanes of a wavefront disagree.  Same values as before, bit for bit.)
NIC_HD float expm1_neg(float x) {
    const float p = fmaf(x, fmaf(x, fmaf(x, fmaf(x, fmaf(x, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
#if defined(__HIP_DEVICE_COMPILE__)
    const float e = __expf(x) - 1.f;
#else
    const float e = expf(x) - 1.f;
#endif
    return x > -0.35f ? x * p : e;   // (the unselected form may overflow to inf for a large |x|; it is only ever discarded)
}
// elu(x) = x for x > 0, expm1(x) otherwise.  (NaN stays NaN: both comparisons are false, exp(NaN) - 1 is selected.)
NIC_HD float elu1(float x) {
    const float neg = expm1_neg(x);
    return x > 0.f ? x : neg;
}
NIC_HD float elu1_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

struct SrStatics {  // per-scenario constants, loaded once
    float p, h, lead;                       // store underage / holding cost, lead time
    float wh_h, wh_lead, wh_edge;           // warehouse
    float e_h[SR_MAXE], e_lead[SR_MAXE];    // echelons
};

NIC_HD SrStatics sr_load_statics(const NicSmallRolloutDesc& d, int64_t b) {
    SrStatics s;
    s.p = t2(d.underage, 0, b);
    s.h = t2(d.holding, 0, b);
    s.lead = t2(d.lead, 0, b);
    s.wh_h = d.Wn ? t2(d.wh_holding, 0, b) : 0.f;
    s.wh_lead = d.Wn ? t2(d.wh_lead, 0, b) : 0.f;
    s.wh_edge = (d.Wn && d.wh_edge.p) ? t2(d.wh_edge, 0, b) : 0.f;
#pragma unroll
    for (int e = 0; e < SR_MAXE; ++e) {
        s.e_h[e] = e < d.E ? t2(d.ech_holding, e, b) : 0.f;
        s.e_lead[e] = e < d.E ? t2(d.ech_lead, e, b) : 0.f;
    }
    return s;
}

// ---- MLP ----------------------------------------------------------------------------------------------------------
// packed weights: [W1 (32 x F), b1 (32)] [W_l (32 x 32), b_l (32)]... [Wout (n_out x 32), bout (n_out)]
NIC_HD int sr_hidden_offset(const NicSmallRolloutDesc& d, int l) {  // l = 0 .. n_hidden-1
    return l == 0 ? 0 : (SR_H * d.F + SR_H) + (l - 1) * (SR_H * SR_H + SR_H);
}
NIC_HD int sr_out_offset(const NicSmallRolloutDesc& d) { return sr_hidden_offset(d, d.n_hidden); }

NIC_HD void sr_layer_first(const NicSmallRolloutDesc& d, const float (&x)[SR_MAXF], float (&y)[SR_H]) {
    const float* W = d.weights;
    const float* bias = d.weights + SR_H * d.F;
#pragma unroll
    for (int n = 0; n < SR_H; ++n) {
        float acc = bias[n];
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k)
            if (k < d.F) acc = fmaf(W[n * d.F + k], x[k], acc);
        y[n] = elu1(acc);
    }
}
NIC_HD void sr_layer_hidden(const float* Wl, const float (&x)[SR_H], float (&y)[SR_H]) {
    const float* bias = Wl + SR_H * SR_H;
#pragma unroll
    for (int n = 0; n < SR_H; ++n) {
        float acc = bias[n];
#pragma unroll
        for (int k = 0; k < SR_H; ++k) acc = fmaf(Wl[n * SR_H + k], x[k], acc);
        y[n] = elu1(acc);
    }
}
NIC_HD void sr_layer_out(const NicSmallRolloutDesc& d, const float (&x)[SR_H], float (&z)[SR_MAXOUT]) {
    const float* W = d.weights + sr_out_offset(d);
    const float* bias = W + d.n_out * SR_H;
#pragma unroll
    for (int n = 0; n < SR_MAXOUT; ++n) {
        float acc = 0.f;
        if (n < d.n_out) {
            acc = bias[n];
#pragma unroll
            for (int k = 0; k < SR_H; ++k) acc = fmaf(W[n * SR_H + k], x[k], acc);
        }
        z[n] = acc;
    }
}

// ---- head transcendentals ------------------------------------------------------------------------------------------------
// libm's log1pf(expf(x)) and 1 / (1 + expf(-x)) are ~100 / ~40 instructions per call on the device; the whole-horizon kernels are
// bound by the length of a period's instruction stream, and in-kernel stamps had cfg2's ONE softplus per lane at
// 0.75 of a 6-us forward period.  On the device: hardware exp / log / reciprocal (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp each),
// softplus as max(x, 0) + log1p(exp(-|x|)) with Kahan's correction log1p(e) = log(u) e / (u - 1), u = fl(1 + e), which keeps the
// RELATIVE accuracy of small orders (x << 0) that log(1 + e) alone loses; relative error <= ~4e-7 for |x| <= 5, growing like
// |x| 6e-8 with the exponent's rounding as in the ELU above.  With rounded orders (evaluation with discrete allocation) the libm
// forms are used: a knife-edge order must round as the oracle's does.  The host build (tests/hostsim) keeps the libm forms.
NIC_HD float sr_softplus1(float z, bool exact) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (exact) return softplus1_fwd(z);
    const float x = z + 1.f;
    const float e = __expf(-fabsf(x));   // (0, 1]
    const float u = 1.f + e, dlt = u - 1.f;
    const float l = dlt == 0.f ? e : __logf(u) * (e * __builtin_amdgcn_rcpf(dlt));
    const float sp = fmaxf(x, 0.f) + l;
    return x > 20.f ? x : sp;
#else
    (void)exact;
    return softplus1_fwd(z);
#endif
}
NIC_HD float sr_sigmoid(float x, bool exact) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (exact) return sigmoidf_(x);
    return __builtin_amdgcn_rcpf(1.f + __expf(-x));
#else
    (void)exact;
    return sigmoidf_(x);
#endif
}
// d softplus(z + 1) / dz = sigmoid(z + 1) (1 above nn.Softplus's threshold)
NIC_HD float sr_softplus1_grad(float z) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float x = z + 1.f;
    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return x > 20.f ? 1.f : sg;
#else
    return softplus1_grad(z);
#endif
}

// ---- orders from logits (heads) -------------------------------------------------------------------------------------
struct SrOrders {
    float store, wh, ech[SR_MAXE];
};

NIC_HD SrOrders sr_head(const NicSmallRolloutDesc& d, const float (&z)[SR_MAXOUT], const float (&st)[SR_MAXF]) {
    SrOrders o;
    o.store = o.wh = 0.f;
#pragma unroll
    for (int e = 0; e < SR_MAXE; ++e) o.ech[e] = 0.f;
    if (d.head == 0) {
        o.store = sr_softplus1(z[0], d.round_orders != 0);  // neural_networks.py:211-212
        if (d.round_orders) o.store = rintf(o.store);  // trainer.py:201-202
        return o;
    }
    // serial (:335-349): rows [echelons..., warehouse, store]; row j = sigmoid(z_j) * upstream_j,
    // upstream = [upper bound, echelon on-hands..., warehouse on-hand]
#pragma unroll
    for (int j = 0; j < SR_MAXE + 2; ++j) {
        if (j < d.E + 2) {
            float up = d.upper_bound;
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k) {
                if (j >= 1 && j <= d.E && k == d.Ws + d.Ww + (j - 1) * d.We) up = st[k];
                if (j == d.E + 1 && k == d.Ws) up = st[k];
            }
            const float a = sr_sigmoid(z[j], d.round_orders != 0) * up;
            if (j < d.E) o.ech[j] = a;
            else if (j == d.E) o.wh = a;
            else o.store = a;
        }
    }
    if (d.round_orders) {  // discrete allocation (trainer.py:201-202): torch.round = round half to even
        o.store = rintf(o.store);
        o.wh = rintf(o.wh);
#pragma unroll
        for (int e = 0; e < SR_MAXE; ++e) o.ech[e] = rintf(o.ech[e]);
    }
    return o;
}

// ---- pipelines held in the st[] register array: segment [o, o+W) ------------------------------------------------------
NIC_HD float sr_at(const float (&a)[SR_MAXF], int idx) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k)
        if (k == idx) r = a[k];
    return r;
}

// new[o] = after + old[o+1]; new[k] = old[k+1]; new[o+W-1] = 0; new[o+L-1] += a if a != 0   (environment.py:405-432)
NIC_HD void sr_segment_fwd(const float (&old)[SR_MAXF], float (&nw)[SR_MAXF], int o, int W, float after, float a, float lead) {
    const int slot = o + (int)lead - 1;
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) {
        if (k >= o && k < o + W) {
            float v = 0.f;
            const float nxt = (k + 1 < SR_MAXF) ? old[(k + 1 < SR_MAXF) ? k + 1 : k] : 0.f;
            if (k == o) v = after + nxt;
            else if (k < o + W - 1) v = nxt;
            if (a != 0.f && k == slot) v += a;
            nw[k] = v;
        }
    }
}

// gradient of one segment: g_old[o] = g_on_hand, g_old[o+1] = g_new[o], g_old[k] = g_new[k-1]; returns the placement
// gradient of the order (0 for an order that is exactly 0)
NIC_HD float sr_segment_bwd(const float (&gn)[SR_MAXF], float (&go)[SR_MAXF], int o, int W, float g_on_hand, float a,
                            float lead) {
    const int slot = o + (int)lead - 1;
    float ga = 0.f;
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) {
        if (k >= o && k < o + W) {
            float v;
            if (k == o) v = g_on_hand;
            else if (k == o + 1) v = gn[(k >= 1) ? k - 1 : 0];
            else v = gn[(k >= 1) ? k - 1 : 0];
            go[k] = v;
            if (a != 0.f && k == slot) ga = gn[k];
        }
    }
    return ga;
}

// one period of dynamics for the one-store chain; returns the period cost, writes the next state
NIC_HD float sr_env_fwd(const NicSmallRolloutDesc& d, const SrStatics& c, const float (&st)[SR_MAXF], float (&nx)[SR_MAXF],
                        float dem, const SrOrders& o) {
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) nx[k] = 0.f;
    const float on_hand = st[0];
    float after = on_hand - dem;
    float cost;
    if (d.maximize_profit) cost = (-c.p) * (on_hand < dem ? on_hand : dem) + c.h * relu(after);
    else cost = c.p * relu(-after) + c.h * relu(after);
    if (d.lost_demand) after = relu(after);
    sr_segment_fwd(st, nx, 0, d.Ws, after, o.store, c.lead);
    float total = cost;
    if (d.Wn) {
        const float w_after = sr_at(st, d.Ws) - o.store;  // ships what the store ordered (no clip, :249)
        float cw = c.wh_h * relu(w_after);
        if (d.wh_edge.p) cw = cw + c.wh_edge * o.wh;
        sr_segment_fwd(st, nx, d.Ws, d.Ww, w_after, o.wh, c.wh_lead);
        total += cw;
    }
    if (d.E > 0) {
        float r_e = 0.f;
#pragma unroll
        for (int e = 0; e < SR_MAXE; ++e) {
            if (e < d.E) {
                const float ship = (e < d.E - 1) ? o.ech[(e + 1 < SR_MAXE) ? e + 1 : e] : o.wh;
                const int off = d.Ws + d.Ww + e * d.We;
                const float e_after = sr_at(st, off) - ship;
                r_e += c.e_h[e] * relu(e_after);
                sr_segment_fwd(st, nx, off, d.We, e_after, o.ech[e], c.e_lead[e]);
            }
        }
        total += r_e;
    }
    return total;
}

// backward of the same period: gn = gradient w.r.t. the next state; fills go (gradient w.r.t. this period's state through
// the dynamics) and the gradients of the orders
NIC_HD SrOrders sr_env_bwd(const NicSmallRolloutDesc& d, const SrStatics& c, const float (&st)[SR_MAXF],
                           const float (&gn)[SR_MAXF], float (&go)[SR_MAXF], float dem, const SrOrders& o, float gr) {
    SrOrders g;
    g.store = g.wh = 0.f;
#pragma unroll
    for (int e = 0; e < SR_MAXE; ++e) g.ech[e] = 0.f;
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) go[k] = 0.f;

    // echelons, upstream first
    float g_to_wh = 0.f;
    if (d.E > 0) {
        float prev = 0.f;
#pragma unroll
        for (int e = 0; e < SR_MAXE; ++e) {
            if (e < d.E) {
                const float ship = (e < d.E - 1) ? o.ech[(e + 1 < SR_MAXE) ? e + 1 : e] : o.wh;
                const int off = d.Ws + d.Ww + e * d.We;
                const float e_after = sr_at(st, off) - ship;
                float g_after = sr_at(gn, off);
                if (e_after >= 0.f) g_after += gr * c.e_h[e];
                const float ga = sr_segment_bwd(gn, go, off, d.We, g_after, o.ech[e], c.e_lead[e]);
                g.ech[e] = ga - prev;
                prev = g_after;
            }
        }
        g_to_wh = -prev;
    }
    // warehouse
    float g_w_after = 0.f;
    if (d.Wn) {
        const float w_after = sr_at(st, d.Ws) - o.store;
        g_w_after = sr_at(gn, d.Ws);
        if (w_after >= 0.f) g_w_after += gr * c.wh_h;
        float ga = sr_segment_bwd(gn, go, d.Ws, d.Ww, g_w_after, o.wh, c.wh_lead);
        if (d.wh_edge.p) ga += gr * c.wh_edge;
        g.wh = ga + g_to_wh;
    }
    // store
    const float on_hand = st[0];
    const float after = on_hand - dem;
    float g_after = gn[0];
    if (d.lost_demand && !(after >= 0.f)) g_after = 0.f;
    float g_on_hand;
    if (d.maximize_profit) {
        if (after >= 0.f) g_after += gr * c.h;
        const float share = on_hand < dem ? 1.f : (on_hand == dem ? 0.5f : 0.f);
        g_on_hand = g_after + gr * (-c.p) * share;
    } else {
        float gc = 0.f;
        if (-after >= 0.f) gc += -c.p;
        if (after >= 0.f) gc += c.h;
        g_on_hand = g_after + gr * gc;
    }
    float ga = sr_segment_bwd(gn, go, 0, d.Ws, g_on_hand, o.store, c.lead);
    if (d.Wn) ga += -g_w_after;
    g.store = ga;
    return g;
}

// head backward: gradients of the orders -> dZ, plus the head's own contribution to the state gradient
NIC_HD void sr_head_bwd(const NicSmallRolloutDesc& d, const float (&z)[SR_MAXOUT], const float (&st)[SR_MAXF],
                        const SrOrders& g, float (&dz)[SR_MAXOUT], float (&go)[SR_MAXF]) {
#pragma unroll
    for (int n = 0; n < SR_MAXOUT; ++n) dz[n] = 0.f;
    if (d.head == 0) {
        dz[0] = g.store * sr_softplus1_grad(z[0]);
        return;
    }
#pragma unroll
    for (int j = 0; j < SR_MAXE + 2; ++j) {
        if (j < d.E + 2) {
            const float gj = j < d.E ? g.ech[(j < SR_MAXE) ? j : 0] : (j == d.E ? g.wh : g.store);
            const float sg = sr_sigmoid(z[j], false);   // (training only: orders are never rounded here)
            float up = d.upper_bound;
            int up_idx = -1;
            if (j >= 1 && j <= d.E) up_idx = d.Ws + d.Ww + (j - 1) * d.We;
            if (j == d.E + 1) up_idx = d.Ws;
            if (up_idx >= 0) up = sr_at(st, up_idx);
            dz[j] = gj * up * sg * (1.f - sg);
            const float g_up = gj * sg;
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k)
                if (k == up_idx) go[k] += g_up;
        }
    }
}

// ---- whole-horizon forward of one scenario --------------------------------------------------------------------------
// hist layouts (row-major rows of T*ldb floats, so that every row is a contraction-contiguous operand of the weight-
// gradient GEMM): states_hist [F][T][ldb], hidden_hist [n_hidden*32][T][ldb], logits_hist [n_out][T][ldb]
template <int NL>
NIC_HD void small_rollout_fwd_scenario(const NicSmallRolloutDesc& d, float* rewards, float* state_final,
                                       float* states_hist, float* hidden_hist, float* logits_hist, int64_t b) {
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    const SrStatics c = sr_load_statics(d, b);
    float st[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) st[k] = k < d.F ? d.state0[(int64_t)k * ldb + b] : 0.f;
    for (int t = 0; t < d.T; ++t) {
        float h1[SR_H], h2[SR_H], h3[SR_H], z[SR_MAXOUT];
        sr_layer_first(d, st, h1);
        if (NL >= 2) sr_layer_hidden(d.weights + sr_hidden_offset(d, 1), h1, h2);
        if (NL >= 3) sr_layer_hidden(d.weights + sr_hidden_offset(d, 2), h2, h3);
        const float(&last)[SR_H] = (NL == 1) ? h1 : ((NL == 2) ? h2 : h3);
        sr_layer_out(d, last, z);
        if (states_hist) {
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k)
                if (k < d.F) states_hist[k * tl + t * ldb + b] = st[k];
#pragma unroll
            for (int n = 0; n < SR_H; ++n) {
                hidden_hist[n * tl + t * ldb + b] = h1[n];
                if (NL >= 2) hidden_hist[(SR_H + n) * tl + t * ldb + b] = h2[n];
                if (NL >= 3) hidden_hist[(2 * SR_H + n) * tl + t * ldb + b] = h3[n];
            }
#pragma unroll
            for (int n = 0; n < SR_MAXOUT; ++n)
                if (n < d.n_out) logits_hist[n * tl + t * ldb + b] = z[n];
        }
        const SrOrders o = sr_head(d, z, st);
        float nx[SR_MAXF];
        const float dem = d.demand[(int64_t)(t + d.t0) * ldb + b];
        rewards[(int64_t)t * ldb + b] = sr_env_fwd(d, c, st, nx, dem, o);
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) st[k] = nx[k];
    }
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k)
        if (k < d.F) state_final[(int64_t)k * ldb + b] = st[k];
}

// ---- whole-horizon backward of one scenario ---------------------------------------------------------------------------
// Reverse sweep over the stored activations.  Emits dZ of every layer ([32*n_hidden][T][ldb], [n_out][T][ldb]) for the
// weight-gradient GEMMs (contraction over T*ldb); the state gradient is carried in registers from period to period.
template <int NL>
NIC_HD void small_rollout_bwd_scenario(const NicSmallRolloutDesc& d, const float* states_hist, const float* hidden_hist,
                                       const float* logits_hist, const NicTable2& g_reward, float* dz_hidden, float* dz_out,
                                       int64_t b) {
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    const SrStatics c = sr_load_statics(d, b);
    const float gr = g_reward.p[b * g_reward.scn_stride];
    float gn[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) gn[k] = 0.f;
    const float* Wout = d.weights + sr_out_offset(d);
    for (int t = d.T - 1; t >= 0; --t) {
        float st[SR_MAXF], z[SR_MAXOUT];
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) st[k] = k < d.F ? states_hist[k * tl + t * ldb + b] : 0.f;
#pragma unroll
        for (int n = 0; n < SR_MAXOUT; ++n) z[n] = n < d.n_out ? logits_hist[n * tl + t * ldb + b] : 0.f;
        const SrOrders o = sr_head(d, z, st);
        const float dem = d.demand[(int64_t)(t + d.t0) * ldb + b];
        float go[SR_MAXF], dz[SR_MAXOUT];
        const SrOrders g = sr_env_bwd(d, c, st, gn, go, dem, o, gr);
        sr_head_bwd(d, z, st, g, dz, go);
#pragma unroll
        for (int n = 0; n < SR_MAXOUT; ++n)
            if (n < d.n_out) dz_out[n * tl + t * ldb + b] = dz[n];
        // output layer -> last hidden layer
        float dh[SR_H];
#pragma unroll
        for (int k = 0; k < SR_H; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < SR_MAXOUT; ++n)
                if (n < d.n_out) acc = fmaf(Wout[n * SR_H + k], dz[n], acc);
            dh[k] = acc * elu1_grad_from_out(hidden_hist[((NL - 1) * SR_H + k) * tl + t * ldb + b]);
        }
        // hidden layers NL-1 .. 1 (0-based l): dZ_l = dh; dH_{l-1} = W_l^T dZ_l * elu'(H_{l-1})
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
#pragma unroll
            for (int n = 0; n < SR_H; ++n) dz_hidden[(l * SR_H + n) * tl + t * ldb + b] = dh[n];
            const float* Wl = d.weights + sr_hidden_offset(d, l);
            float dprev[SR_H];
#pragma unroll
            for (int k = 0; k < SR_H; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int n = 0; n < SR_H; ++n) acc = fmaf(Wl[n * SR_H + k], dh[n], acc);
                dprev[k] = acc * elu1_grad_from_out(hidden_hist[((l - 1) * SR_H + k) * tl + t * ldb + b]);
            }
#pragma unroll
            for (int k = 0; k < SR_H; ++k) dh[k] = dprev[k];
        }
#pragma unroll
        for (int n = 0; n < SR_H; ++n) dz_hidden[n * tl + t * ldb + b] = dh[n];
        // first layer -> state (the reference detaches vanilla_serial's MLP input, neural_networks.py:329)
        if (!d.detach_input) {
            const float* W1 = d.weights;
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k) {
                if (k < d.F) {
                    float acc = 0.f;
#pragma unroll
                    for (int n = 0; n < SR_H; ++n) acc = fmaf(W1[n * d.F + k], dh[n], acc);
                    go[k] += acc;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) gn[k] = go[k];
    }
}

}  // namespace nic
