// Per-scenario bodies of the policy heads: what the reference's policies do AFTER the master MLP
// (logits -> feasible orders).  One lane = one scenario; all loops are over that scenario's own locations.
// NIC_HD so tests/hostsim can run the same arithmetic on the CPU.
#pragma once
#include <math.h>
#include <stdint.h>

#include "env_step_body.h"

namespace nic {

NIC_HD float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// The softmax of the warehouse head on the device: e = exp(z - max) on the hardware exponential (v_exp_f32: the argument is <= 0,
// relative error ~1 ulp + |z - max| 6e-8 from the exponent's rounding) and the share e / denom as e * rcp(denom) (v_rcp_f32, 1 ulp;
// the reciprocal of a (scenario, warehouse) is computed once) instead of libm's expf (~15 instructions) and an IEEE division (~10)
// per store - a lane of BASELINE cfg5's head walks 48 (store, warehouse) pairs.  The host build (tests/hostsim, the oracle's
// twin) keeps expf and the division.
NIC_HD float head_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}
NIC_HD float head_share(float e, float denom) {
#if defined(__HIP_DEVICE_COMPILE__)
    return e * __builtin_amdgcn_rcpf(denom);
#else
    return e / denom;
#endif
}

// softplus(z + 1) with nn.Softplus defaults (beta 1, threshold 20)   neural_networks.py:211-212
NIC_HD float softplus1_fwd(float z) {
    const float x = z + 1.f;
    return x > 20.f ? x : log1pf(expf(x));
}
NIC_HD float softplus1_grad(float z) {
    const float x = z + 1.f;
    if (x > 20.f) return 1.f;
    const float e = expf(x);
    return e / (e + 1.f);
}

// vanilla_warehouse head.  neural_networks.py:393-426 + apply_softmax_feasibility_function :140-166.
//   Z rows: s*Wn + w (store s <- warehouse w), then S*Wn + w (warehouse w's own order logit)
// Store loops run in batches of kHeadBatch with the loads of a batch issued before its arithmetic (one lane walks up to
// S stores three times: with one load in flight at a time the kernel was pure latency).  Loads are unconditional (the
// adjacency only selects which values are used), the arithmetic and its order are unchanged.
constexpr int kHeadBatch = 8;

NIC_HD void head_warehouse_fwd_one(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int transshipment,
                                   float* store_orders, float* wh_orders, int S, int Wn, int Ww, int64_t ldb, int64_t b,
                                   int w) {
    const float stock = wh_inv[(int64_t)w * Ww * ldb + b];  // on-hand slot of warehouse w (:146)
    const float* Zw = Z + (int64_t)w * ldb + b;             // row s*Wn + w  ->  Zw[s * Wn * ldb]
    const int64_t rs = (int64_t)Wn * ldb;
    const int32_t* aw = adj + w * S;
    // pass 1: max over connected logits (and the constant-1 'keep' logit unless transshipment)
    float m = transshipment ? -INFINITY : 1.f;
    int n_conn = 0;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (a[u]) {
                m = z[u] > m ? z[u] : m;
                ++n_conn;
            }
    }
    // pass 2: denominator
    float denom = transshipment ? 0.f : head_exp(1.f - m);
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (a[u]) denom += head_exp(z[u] - m);
    }
    // pass 3: shares of the on-hand stock
    float* ow = store_orders + (int64_t)w * ldb + b;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], o[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) o[u] = (a[u] && n_conn > 0) ? (head_share(head_exp(z[u] - m), denom)) * stock : 0.f;
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) ow[(s0 + u) * rs] = o[u];
    }
    wh_orders[(int64_t)w * ldb + b] = sigmoidf_(Z[((int64_t)S * Wn + w) * ldb + b]) * ub;  // :422
}

// warehouses are independent of each other: the device kernel gives lane q of a scenario's quad the warehouses w = q, q+4, ...
NIC_HD void head_warehouse_fwd_scenario(const float* Z, const float* wh_inv, const int32_t* adj, float ub,
                                        int transshipment, float* store_orders, float* wh_orders, int S, int Wn,
                                        int Ww, int64_t ldb, int64_t b) {
    for (int w = 0; w < Wn; ++w)
        head_warehouse_fwd_one(Z, wh_inv, adj, ub, transshipment, store_orders, wh_orders, S, Wn, Ww, ldb, b, w);
}

NIC_HD void head_warehouse_bwd_one(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int transshipment,
                                   const float* g_store_orders, const float* g_wh_orders, float* dZ, float* g_wh_inv, int S,
                                   int Wn, int Ww, int64_t ldb, int64_t b, int w) {
    const float stock = wh_inv[(int64_t)w * Ww * ldb + b];
    const float* Zw = Z + (int64_t)w * ldb + b;
    const float* gw = g_store_orders + (int64_t)w * ldb + b;
    const int64_t rs = (int64_t)Wn * ldb;
    const int32_t* aw = adj + w * S;
    float m = transshipment ? -INFINITY : 1.f;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (a[u]) m = z[u] > m ? z[u] : m;
    }
    float denom = transshipment ? 0.f : head_exp(1.f - m);
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (a[u]) denom += head_exp(z[u] - m);
    }
    // order_s = y_s * stock;  g_y_s = g_order_s * stock;  g_stock = sum_s g_order_s * y_s
    // softmax backward: dz_s = y_s * (g_y_s - sum_j y_j g_y_j)   (the keep column has g_y = 0)
    float dot = 0.f, g_stock = 0.f;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], go[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            go[u] = gw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (a[u]) {
                const float y = head_share(head_exp(z[u] - m), denom);
                dot += y * (go[u] * stock);
                g_stock += go[u] * y;
            }
    }
    float* dw = dZ + (int64_t)w * ldb + b;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], go[kHeadBatch], dz[kHeadBatch];
        int a[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            go[u] = gw[s * rs];
            a[u] = s0 + u < S ? aw[s] : 0;
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            dz[u] = 0.f;
            if (a[u]) {
                const float y = head_share(head_exp(z[u] - m), denom);
                dz[u] = y * (go[u] * stock - dot);
            }
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) dw[(s0 + u) * rs] = dz[u];
    }
    g_wh_inv[(int64_t)w * Ww * ldb + b] += g_stock;
    const float sg = sigmoidf_(Z[((int64_t)S * Wn + w) * ldb + b]);
    dZ[((int64_t)S * Wn + w) * ldb + b] = g_wh_orders[(int64_t)w * ldb + b] * ub * sg * (1.f - sg);
}

NIC_HD void head_warehouse_bwd_scenario(const float* Z, const float* wh_inv, const int32_t* adj, float ub,
                                        int transshipment, const float* g_store_orders, const float* g_wh_orders,
                                        float* dZ, float* g_wh_inv, int S, int Wn, int Ww, int64_t ldb, int64_t b) {
    for (int w = 0; w < Wn; ++w)
        head_warehouse_bwd_one(Z, wh_inv, adj, ub, transshipment, g_store_orders, g_wh_orders, dZ, g_wh_inv, S, Wn, Ww, ldb,
                               b, w);
}

// ---- the same head with the stores of a scenario split over FOUR lanes (the env step's "quad") -----------------------------
// One lane per (scenario, warehouse) walks up to S stores three (forward) or four (backward) times, one dependent batch of loads
// after the other: 9.5 / 13.4 us per period at BASELINE cfg3 and 31 / 43 us at cfg5 for ~10 / ~50 MB, i.e. 0.1 - 0.2 of the HBM
// roofline.  Here lane q of the quad keeps the logits of stores q, q+4, ... in registers (ONE load per logit, all of a lane's
// loads in flight together), and the softmax's three reductions - maximum, denominator, and in the backward sum_j y_j g_j -
// cross the quad through the caller's exchange (LDS on the device, plain arrays in the host build) in the fixed order
// ((p0 + p1) + p2) + p3.  exp(z - m) is evaluated once per logit.  Pieces are NIC_HD: tests/hostsim runs the four lanes in a
// loop, so host and device agree bit for bit.
template <int MAXSQ>
struct HeadLane {
    float z[MAXSQ], e[MAXSQ], g[MAXSQ];
    int a[MAXSQ];
};

// loads this lane's logits (and, backward, the order gradients); returns the lane's maximum over connected stores and counts them
// `zrow` (optional, [Wn][S]): the row of Z / dZ that holds the logit of (warehouse w, store s) when the logits layer only
// computes the connected pairs (compact logits: the rollout engine on sparse many-warehouse graphs); pairs without an edge hold
// any valid row (their value is loaded and never used).  nullptr: the reference's row s * Wn + w.
template <int MAXSQ, bool BWD>
NIC_HD float head_quad_load(HeadLane<MAXSQ>& L, const float* Z, const float* g_store_orders, const int32_t* adj, int S, int Wn,
                            int64_t ldb, int64_t b, int w, int q, int& n_conn, const int32_t* zrow = nullptr) {
    const float* Zw = Z + (int64_t)w * ldb + b;
    const float* gw = BWD ? g_store_orders + (int64_t)w * ldb + b : nullptr;
    const int64_t rs = (int64_t)Wn * ldb;
    const int32_t* aw = adj + w * S;
    // every load is unconditional and its value is used unconditionally (a load that only feeds one side of a select gets sunk
    // into a branch by hipcc and then costs one exposed round trip PER STORE: 16 x vmcnt(0) in the first version of this piece)
    if (zrow) {
        int zr[MAXSQ];
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) {
            const int s = q + kQuad * u;
            const int sc = s < S ? s : S - 1;
            zr[u] = zrow[w * S + sc];
            if (BWD) L.g[u] = gw[sc * rs];
            L.a[u] = aw[sc];
        }
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) L.z[u] = Z[(int64_t)zr[u] * ldb + b];
    } else {
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) {
            const int s = q + kQuad * u;
            const int sc = s < S ? s : S - 1;
            L.z[u] = Zw[sc * rs];
            if (BWD) L.g[u] = gw[sc * rs];
            L.a[u] = aw[sc];
        }
    }
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u) L.a[u] &= -(int)(q + kQuad * u < S);   // stores past S are not connected
    float m = -INFINITY;
    n_conn = 0;
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u)
        if (L.a[u]) {
            m = L.z[u] > m ? L.z[u] : m;
            ++n_conn;
        }
    return m;
}
NIC_HD float head_quad_max(float m0, float m1, float m2, float m3, int transshipment) {
    float m = transshipment ? -INFINITY : 1.f;   // the constant-1 'keep' logit (:150-153)
    m = m0 > m ? m0 : m;
    m = m1 > m ? m1 : m;
    m = m2 > m ? m2 : m;
    return m3 > m ? m3 : m;
}
// e = exp(z - m) of the connected stores; returns the lane's share of the denominator
template <int MAXSQ>
NIC_HD float head_quad_exp(HeadLane<MAXSQ>& L, float m) {
    float d = 0.f;
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u) {
        L.e[u] = L.a[u] ? head_exp(L.z[u] - m) : 0.f;
        d += L.e[u];
    }
    return d;
}
NIC_HD float head_quad_denom(float d0, float d1, float d2, float d3, float m, int transshipment) {
    return (transshipment ? 0.f : head_exp(1.f - m)) + combine4(d0, d1, d2, d3);
}
template <int MAXSQ>
NIC_HD void head_quad_fwd_store(const HeadLane<MAXSQ>& L, float denom, float stock, int n_conn, float* store_orders, int S, int Wn,
                                int64_t ldb, int64_t b, int w, int q) {
    float* ow = store_orders + (int64_t)w * ldb + b;
    const int64_t rs = (int64_t)Wn * ldb;
    float o[MAXSQ];
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u) o[u] = (L.a[u] && n_conn > 0) ? head_share(L.e[u], denom) * stock : 0.f;
    if (S == kQuad * MAXSQ) {  // every lane owns MAXSQ stores: straight-line stores
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) ow[(q + kQuad * u) * rs] = o[u];
    } else {
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u)
            if (q + kQuad * u < S) ow[(q + kQuad * u) * rs] = o[u];
    }
}
// backward: y = e / denom replaces e; the lane's shares of sum_j y_j (g_j stock) and of d/d(stock)
template <int MAXSQ>
NIC_HD void head_quad_bwd_dots(HeadLane<MAXSQ>& L, float denom, float stock, float& dot, float& g_stock) {
    dot = 0.f;
    g_stock = 0.f;
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u) {  // (e is exactly 0 for stores that are not connected: their terms add +0)
        L.e[u] = L.a[u] ? head_share(L.e[u], denom) : 0.f;
        const float gz = L.a[u] ? L.g[u] : 0.f;
        dot += L.e[u] * (gz * stock);
        g_stock += gz * L.e[u];
    }
}
template <int MAXSQ>
NIC_HD void head_quad_bwd_store(const HeadLane<MAXSQ>& L, float dot, float stock, float* dZ, int S, int Wn, int64_t ldb, int64_t b,
                                int w, int q, const int32_t* zrow = nullptr) {
    float* dw = dZ + (int64_t)w * ldb + b;
    const int64_t rs = (int64_t)Wn * ldb;
    float d[MAXSQ];
#pragma unroll
    for (int u = 0; u < MAXSQ; ++u) d[u] = L.a[u] ? L.e[u] * (L.g[u] * stock - dot) : 0.f;
    if (zrow) {   // compact logits: only connected pairs have a row
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) {
            const int s = q + kQuad * u;
            if (s < S && L.a[u]) dZ[(int64_t)zrow[w * S + s] * ldb + b] = d[u];
        }
        return;
    }
    if (S == kQuad * MAXSQ) {
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u) dw[(q + kQuad * u) * rs] = d[u];
    } else {
#pragma unroll
        for (int u = 0; u < MAXSQ; ++u)
            if (q + kQuad * u < S) dw[(q + kQuad * u) * rs] = d[u];
    }
}
// the warehouse's own order (:422) and its gradient: one lane per (scenario, warehouse)
// (first_wh_row: row of Z / dZ of warehouse 0's own logit; < 0 = the reference's S * Wn)
NIC_HD void head_wh_order_fwd(const float* Z, float ub, float* wh_orders, int S, int Wn, int64_t ldb, int64_t b, int w,
                              int first_wh_row = -1) {
    const int64_t row = (first_wh_row >= 0 ? first_wh_row : S * Wn) + w;
    wh_orders[(int64_t)w * ldb + b] = sigmoidf_(Z[row * ldb + b]) * ub;
}
NIC_HD void head_wh_order_bwd(const float* Z, float ub, const float* g_wh_orders, float* dZ, int S, int Wn, int64_t ldb,
                              int64_t b, int w, int first_wh_row = -1) {
    const int64_t row = (first_wh_row >= 0 ? first_wh_row : S * Wn) + w;
    const float sg = sigmoidf_(Z[row * ldb + b]);
    dZ[row * ldb + b] = g_wh_orders[(int64_t)w * ldb + b] * ub * sg * (1.f - sg);
}

// host-side composition of the quad pieces (what the device kernels do with LDS in between): tests/hostsim
template <int MAXSQ>
NIC_HD void head_warehouse_fwd_quad_scenario(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int transshipment,
                                             float* store_orders, float* wh_orders, int S, int Wn, int Ww, int64_t ldb, int64_t b) {
    for (int w = 0; w < Wn; ++w) {
        HeadLane<MAXSQ> L[kQuad];
        float mq[kQuad], dq[kQuad];
        int nq[kQuad];
        for (int q = 0; q < kQuad; ++q) mq[q] = head_quad_load<MAXSQ, false>(L[q], Z, nullptr, adj, S, Wn, ldb, b, w, q, nq[q]);
        const float m = head_quad_max(mq[0], mq[1], mq[2], mq[3], transshipment);
        const int n_conn = nq[0] + nq[1] + nq[2] + nq[3];
        for (int q = 0; q < kQuad; ++q) dq[q] = head_quad_exp<MAXSQ>(L[q], m);
        const float denom = head_quad_denom(dq[0], dq[1], dq[2], dq[3], m, transshipment);
        const float stock = wh_inv[(int64_t)w * Ww * ldb + b];
        for (int q = 0; q < kQuad; ++q) head_quad_fwd_store<MAXSQ>(L[q], denom, stock, n_conn, store_orders, S, Wn, ldb, b, w, q);
        head_wh_order_fwd(Z, ub, wh_orders, S, Wn, ldb, b, w);
    }
}
template <int MAXSQ>
NIC_HD void head_warehouse_bwd_quad_scenario(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int transshipment,
                                             const float* g_store_orders, const float* g_wh_orders, float* dZ, float* g_wh_inv,
                                             int S, int Wn, int Ww, int64_t ldb, int64_t b) {
    for (int w = 0; w < Wn; ++w) {
        HeadLane<MAXSQ> L[kQuad];
        float mq[kQuad], dq[kQuad], tq[kQuad], sq[kQuad];
        int nq[kQuad];
        for (int q = 0; q < kQuad; ++q)
            mq[q] = head_quad_load<MAXSQ, true>(L[q], Z, g_store_orders, adj, S, Wn, ldb, b, w, q, nq[q]);
        const float m = head_quad_max(mq[0], mq[1], mq[2], mq[3], transshipment);
        for (int q = 0; q < kQuad; ++q) dq[q] = head_quad_exp<MAXSQ>(L[q], m);
        const float denom = head_quad_denom(dq[0], dq[1], dq[2], dq[3], m, transshipment);
        const float stock = wh_inv[(int64_t)w * Ww * ldb + b];
        for (int q = 0; q < kQuad; ++q) head_quad_bwd_dots<MAXSQ>(L[q], denom, stock, tq[q], sq[q]);
        const float dot = combine4(tq[0], tq[1], tq[2], tq[3]);
        for (int q = 0; q < kQuad; ++q) head_quad_bwd_store<MAXSQ>(L[q], dot, stock, dZ, S, Wn, ldb, b, w, q);
        g_wh_inv[(int64_t)w * Ww * ldb + b] += combine4(sq[0], sq[1], sq[2], sq[3]);
        head_wh_order_bwd(Z, ub, g_wh_orders, dZ, S, Wn, ldb, b, w);
    }
}

// vanilla_serial head.  neural_networks.py:335-349: rows [E echelons..., warehouse, store]; row j is
// sigmoid(Z[j]) * upstream_j with upstream = [upper bound, echelon on-hands..., warehouse on-hand].
NIC_HD float serial_upstream(const float* wh_inv, const float* ech_inv, float ub, int j, int E, int We, int64_t ldb,
                             int64_t b) {
    if (j == 0) return ub;
    if (j <= E) return ech_inv[(int64_t)(j - 1) * We * ldb + b];
    return wh_inv[b];
}

NIC_HD void head_serial_fwd_scenario(const float* Z, const float* wh_inv, const float* ech_inv, float ub,
                                     float* store_orders, float* wh_orders, float* ech_orders, int E, int Ww, int We,
                                     int64_t ldb, int64_t b) {
    (void)Ww;
    for (int j = 0; j < E + 2; ++j) {
        const float a = sigmoidf_(Z[(int64_t)j * ldb + b]) * serial_upstream(wh_inv, ech_inv, ub, j, E, We, ldb, b);
        if (j < E) ech_orders[(int64_t)j * ldb + b] = a;
        else if (j == E) wh_orders[b] = a;
        else store_orders[b] = a;
    }
}

NIC_HD void head_serial_bwd_scenario(const float* Z, const float* wh_inv, const float* ech_inv, float ub,
                                     const float* g_store_orders, const float* g_wh_orders, const float* g_ech_orders,
                                     float* dZ, float* g_wh_inv, float* g_ech_inv, int E, int Ww, int We, int64_t ldb,
                                     int64_t b) {
    (void)Ww;
    for (int j = 0; j < E + 2; ++j) {
        const float g = j < E ? g_ech_orders[(int64_t)j * ldb + b] : (j == E ? g_wh_orders[b] : g_store_orders[b]);
        const float sg = sigmoidf_(Z[(int64_t)j * ldb + b]);
        const float up = serial_upstream(wh_inv, ech_inv, ub, j, E, We, ldb, b);
        dZ[(int64_t)j * ldb + b] = g * up * sg * (1.f - sg);
        const float g_up = g * sg;
        if (j >= 1 && j <= E) g_ech_inv[(int64_t)(j - 1) * We * ldb + b] += g_up;
        else if (j == E + 1) g_wh_inv[b] += g_up;
    }
}

// data_driven head (DataDrivenNet.forward, neural_networks.py:474-515 + apply_proportional_allocation :111-138), one (scenario,
// warehouse): Z rows = [Wn warehouse orders | S x Wn store orders (store-major)] BEFORE the output ReLU; mask [S][Wn] = adjacency.
// out = relu(Z); the warehouse's own order passes through; its store orders are masked, summed in store order and scaled by
// min(1, pipeline total of the warehouse / (sum + 1e-10)).  Loads in batches of kHeadBatch ahead of the (serial, store-ordered)
// arithmetic, as in the vanilla head above.
NIC_HD void head_data_driven_fwd_one(const float* Z, const float* wh, const float* mask, float* so, float* wo, int S, int Wn,
                                     int Ww, int64_t ldb, int64_t b, int w) {
    wo[(int64_t)w * ldb + b] = fmaxf(Z[(int64_t)w * ldb + b], 0.f);
    float avail = 0.f;
    for (int k = 0; k < Ww; ++k) avail += wh[((int64_t)w * Ww + k) * ldb + b];
    const float* Zw = Z + (int64_t)(Wn + w) * ldb + b;   // row Wn + s*Wn + w -> Zw[s * rs]
    const int64_t rs = (int64_t)Wn * ldb;
    float sum = 0.f;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], mk[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            mk[u] = mask[s * Wn + w];
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) sum += fmaxf(z[u], 0.f) * mk[u];
    }
    const float sc = fminf(avail / (sum + 1e-10f), 1.f);
    float* ow = so + (int64_t)w * ldb + b;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], mk[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            mk[u] = mask[s * Wn + w];
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) ow[(s0 + u) * rs] = fmaxf(z[u], 0.f) * mk[u] * sc;
    }
}

// adjoint: dZ (every row of this warehouse written) and g_wh[w][k][b] += d(scale) / (sum + eps) for every slot k of the pipeline
NIC_HD void head_data_driven_bwd_one(const float* Z, const float* wh, const float* mask, const float* g_so, const float* g_wo,
                                     float* dZ, float* g_wh, int S, int Wn, int Ww, int64_t ldb, int64_t b, int w) {
    dZ[(int64_t)w * ldb + b] = Z[(int64_t)w * ldb + b] > 0.f ? g_wo[(int64_t)w * ldb + b] : 0.f;
    float avail = 0.f;
    for (int k = 0; k < Ww; ++k) avail += wh[((int64_t)w * Ww + k) * ldb + b];
    const float* Zw = Z + (int64_t)(Wn + w) * ldb + b;
    const float* gw = g_so + (int64_t)w * ldb + b;
    float* dZw = dZ + (int64_t)(Wn + w) * ldb + b;
    const int64_t rs = (int64_t)Wn * ldb;
    float sum = 0.f, dot = 0.f;
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], mk[kHeadBatch], gs[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            mk[u] = mask[s * Wn + w];
            gs[u] = gw[s * rs];
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) {
                const float a = fmaxf(z[u], 0.f) * mk[u];
                sum += a;
                dot += gs[u] * a;
            }
    }
    const float den = sum + 1e-10f, ratio = avail / den;
    const float sc = fminf(ratio, 1.f);
    const float d_scale = ratio <= 1.f ? dot : 0.f;          // torch.clip(max = 1) passes the gradient where ratio <= 1
    const float common = -(d_scale * avail / (den * den));
    for (int s0 = 0; s0 < S; s0 += kHeadBatch) {
        float z[kHeadBatch], mk[kHeadBatch], gs[kHeadBatch];
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            z[u] = Zw[s * rs];
            mk[u] = mask[s * Wn + w];
            gs[u] = gw[s * rs];
        }
#pragma unroll
        for (int u = 0; u < kHeadBatch; ++u)
            if (s0 + u < S) {
                const float da = gs[u] * sc + common;
                dZw[(s0 + u) * rs] = z[u] > 0.f ? da * mk[u] : 0.f;
            }
    }
    const float g_av = d_scale / den;
    for (int k = 0; k < Ww; ++k) g_wh[((int64_t)w * Ww + k) * ldb + b] += g_av;
}

}  // namespace nic
