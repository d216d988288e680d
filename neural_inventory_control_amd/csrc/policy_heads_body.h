// Per-scenario bodies of the policy heads: what the reference's policies do AFTER the master MLP
// (logits -> feasible orders).  One lane = one scenario; all loops are over that scenario's own locations.
// NIC_HD so tests/hostsim can run the same arithmetic on the CPU.
#pragma once
#include <math.h>
#include <stdint.h>

#include "env_step_body.h"

namespace nic {

NIC_HD float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

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
    float denom = transshipment ? 0.f : expf(1.f - m);
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
            if (a[u]) denom += expf(z[u] - m);
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
        for (int u = 0; u < kHeadBatch; ++u) o[u] = (a[u] && n_conn > 0) ? (expf(z[u] - m) / denom) * stock : 0.f;
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
    float denom = transshipment ? 0.f : expf(1.f - m);
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
            if (a[u]) denom += expf(z[u] - m);
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
                const float y = expf(z[u] - m) / denom;
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
                const float y = expf(z[u] - m) / denom;
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

}  // namespace nic
