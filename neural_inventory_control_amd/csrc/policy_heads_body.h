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
NIC_HD void head_warehouse_fwd_one(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int transshipment,
                                   float* store_orders, float* wh_orders, int S, int Wn, int Ww, int64_t ldb, int64_t b,
                                   int w) {
    const float stock = wh_inv[(int64_t)w * Ww * ldb + b];  // on-hand slot of warehouse w (:146)
    // pass 1: max over connected logits (and the constant-1 'keep' logit unless transshipment)
    float m = transshipment ? -INFINITY : 1.f;
    int n_conn = 0;
    for (int s = 0; s < S; ++s)
        if (adj[w * S + s]) {
            const float z = Z[((int64_t)s * Wn + w) * ldb + b];
            m = z > m ? z : m;
            ++n_conn;
        }
    // pass 2: denominator
    float denom = transshipment ? 0.f : expf(1.f - m);
    for (int s = 0; s < S; ++s)
        if (adj[w * S + s]) denom += expf(Z[((int64_t)s * Wn + w) * ldb + b] - m);
    // pass 3: shares of the on-hand stock
    for (int s = 0; s < S; ++s) {
        float o = 0.f;
        if (adj[w * S + s] && n_conn > 0) o = (expf(Z[((int64_t)s * Wn + w) * ldb + b] - m) / denom) * stock;
        store_orders[((int64_t)s * Wn + w) * ldb + b] = o;
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
    float m = transshipment ? -INFINITY : 1.f;
    for (int s = 0; s < S; ++s)
        if (adj[w * S + s]) {
            const float z = Z[((int64_t)s * Wn + w) * ldb + b];
            m = z > m ? z : m;
        }
    float denom = transshipment ? 0.f : expf(1.f - m);
    for (int s = 0; s < S; ++s)
        if (adj[w * S + s]) denom += expf(Z[((int64_t)s * Wn + w) * ldb + b] - m);
    // order_s = y_s * stock;  g_y_s = g_order_s * stock;  g_stock = sum_s g_order_s * y_s
    // softmax backward: dz_s = y_s * (g_y_s - sum_j y_j g_y_j)   (the keep column has g_y = 0)
    float dot = 0.f, g_stock = 0.f;
    for (int s = 0; s < S; ++s)
        if (adj[w * S + s]) {
            const float y = expf(Z[((int64_t)s * Wn + w) * ldb + b] - m) / denom;
            const float go = g_store_orders[((int64_t)s * Wn + w) * ldb + b];
            dot += y * (go * stock);
            g_stock += go * y;
        }
    for (int s = 0; s < S; ++s) {
        float dz = 0.f;
        if (adj[w * S + s]) {
            const float y = expf(Z[((int64_t)s * Wn + w) * ldb + b] - m) / denom;
            const float go = g_store_orders[((int64_t)s * Wn + w) * ldb + b];
            dz = y * (go * stock - dot);
        }
        dZ[((int64_t)s * Wn + w) * ldb + b] = dz;
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
