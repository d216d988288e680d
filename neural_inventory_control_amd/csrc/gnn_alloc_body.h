// Proportional allocation head of the GNN policy for ONE supplying warehouse (neural_networks.py:111-138 via :1435-1492), one lane
// = one scenario: shared by the stand-alone launches (csrc/mlp3.hip: nic_gnn_alloc_fwd / _bwd) and the launches fused with the env
// step (csrc/gnn_alloc_env.hip).  out [E][ldb] = desired quantity per edge; members = internal edges 0..S-1 (+ e_self if >= 0);
// orders [S+1][ldb] = out[s] * scale for the stores, out[e_supplier] for the warehouse's own order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nic {
constexpr int kAllocBatch = 8;

// `out` rows are out_ld apart and this scenario is column out_b of them (the period kernel keeps the desired quantities of its
// 16 scenarios in LDS: out_ld = 16, out_b = the scenario's index in the block); everything else is indexed by (b, ldb).
__device__ __forceinline__ void gnn_alloc_fwd_one(const float* __restrict__ out, int64_t out_ld, int64_t out_b,
                                                  const float* __restrict__ on_hand, float* __restrict__ orders,
                                                  float* __restrict__ sums, float* __restrict__ ratio, float* __restrict__ scale, int S,
                                                  int e_self, int e_sup, int cap_at_one, int64_t b, int64_t ldb) {
#pragma clang fp contract(off)   // separate multiplies and adds, like the aten ops this replaces
    // rows fetched kAllocBatch at a time (unconditionally: a row past S reads row 0 and is dropped by a select), then added in
    // store order - as `sum += out[s]` in a loop with a run-time trip count every row was one dependent memory round trip
    const float oh = on_hand[b], self_v = out[(int64_t)(e_self >= 0 ? e_self : 0) * out_ld + out_b],
                sup_v = out[(int64_t)e_sup * out_ld + out_b];
    float sum = 0.f;
    for (int s0 = 0; s0 < S; s0 += kAllocBatch) {
        float v[kAllocBatch];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) v[u] = out[(int64_t)(s0 + u < S ? s0 + u : 0) * out_ld + out_b];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) sum = s0 + u < S ? sum + v[u] : sum;
    }
    if (e_self >= 0) sum += self_v;
    const float r = oh / (sum + 1e-10f);
    const float sc = cap_at_one ? fminf(r, 1.f) : r;   // torch.clamp(max = 1)
    sums[b] = sum;
    ratio[b] = r;
    scale[b] = sc;
    for (int s0 = 0; s0 < S; s0 += kAllocBatch) {
        float v[kAllocBatch];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) v[u] = out[(int64_t)(s0 + u < S ? s0 + u : 0) * out_ld + out_b];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u)
            if (s0 + u < S) orders[(int64_t)(s0 + u) * ldb + b] = v[u] * sc;
    }
    orders[(int64_t)S * ldb + b] = sup_v;
}
__device__ __forceinline__ void gnn_alloc_fwd_one(const float* __restrict__ out, const float* __restrict__ on_hand,
                                                  float* __restrict__ orders, float* __restrict__ sums, float* __restrict__ ratio,
                                                  float* __restrict__ scale, int S, int e_self, int e_sup, int cap_at_one, int64_t b,
                                                  int64_t ldb) {
    gnn_alloc_fwd_one(out, ldb, b, on_hand, orders, sums, ratio, scale, S, e_self, e_sup, cap_at_one, b, ldb);
}

// adjoint: g_orders [S+1][ldb] -> d_out [E][ldb] (every row written) and g_on_hand[b] += d_scale / (sum + eps).
// clamp(max) passes the gradient where ratio <= 1 (torch's rule); the self loop's allocation feeds nothing.
__device__ __forceinline__ void gnn_alloc_bwd_one(const float* __restrict__ out, const float* __restrict__ on_hand,
                                                  const float* g_orders, const float* __restrict__ sums,
                                                  const float* __restrict__ ratio, const float* __restrict__ scale,
                                                  float* __restrict__ d_out, float* g_on_hand, int S, int E, int e_self, int e_sup,
                                                  int cap_at_one, int64_t b, int64_t ldb) {
#pragma clang fp contract(off)
    const float rt = ratio[b], sm = sums[b], oh = on_hand[b], sc = scale[b], g_sup = g_orders[(int64_t)S * ldb + b], goh = g_on_hand[b];
    float dot = 0.f;
    for (int s0 = 0; s0 < S; s0 += kAllocBatch) {   // (batched like the forward: same products, same order of additions)
        float g[kAllocBatch], o[kAllocBatch];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) {
            const int64_t row = (int64_t)(s0 + u < S ? s0 + u : 0) * ldb + b;
            g[u] = g_orders[row];
            o[u] = out[row];
        }
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) dot = s0 + u < S ? dot + g[u] * o[u] : dot;
    }
    const float passes = cap_at_one ? (rt <= 1.f ? 1.f : 0.f) : 1.f;
    const float d_scale = dot * passes;
    const float den = sm + 1e-10f;
    const float common = -(d_scale * oh / (den * den));
    for (int e0 = 0; e0 < E; e0 += kAllocBatch) {
        float g[kAllocBatch];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) g[u] = g_orders[(int64_t)(e0 + u < S ? e0 + u : 0) * ldb + b];
#pragma unroll
        for (int u = 0; u < kAllocBatch; ++u) {
            const int e = e0 + u;
            if (e < E) {
                float v = 0.f;
                if (e < S || e == e_self) v = common;
                if (e < S) v += g[u] * sc;
                if (e == e_sup) v = g_sup;
                d_out[(int64_t)e * ldb + b] = v;
            }
        }
    }
    g_on_hand[b] = goh + d_scale / den;
}
}  // namespace nic
