// Per-scenario body of one period of inventory dynamics (forward + analytic backward).
//
// One GPU lane owns one scenario b: every reduction the reference performs inside a period is over the stores /
// warehouses of ONE scenario (environment.py:234,247,270,299), so a lane loops over its locations and needs no
// cross-lane traffic; with the scenario-minor layout each of its loads is one coalesced 256-B wave access.
//
// The body is NIC_HD (host + device) so that tests can run the identical arithmetic on the CPU against the oracle
// (tests/hostsim) before any GPU time is spent; the product only ever launches the __global__ wrappers in
// env_step.hip.  Compiled with -ffp-contract=off so that mul/add round separately like the reference's aten ops.
#pragma once
#include <stdint.h>

#include "../../include/nic_rollout.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NIC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define NIC_HD inline
#endif

namespace nic {

NIC_HD float t2(const NicTable2& t, int loc, int64_t b) { return t.p[loc * t.loc_stride + b * t.scn_stride]; }
NIC_HD float t3(const NicTable3& t, int loc, int sup, int64_t b) {
    return t.p[loc * t.loc_stride + sup * t.sup_stride + b * t.scn_stride];
}
NIC_HD float relu(float x) { return x > 0.f ? x : 0.f; }

// Sums over the stores of one scenario use four interleaved accumulators (stores s = q mod 4) combined as
// ((a0+a1)+a2)+a3 ("Sum4 order").  Four independent chains give ILP / let four lanes share the work, and the order
// coincides with what the reference's CPU `sum(dim=1)` does for outer-dimension reductions (measured on torch 2.10), which
// keeps knife-edge `>= 0` masks on the warehouse on-hand (environment.py:249-251) on the reference's side of zero more
// often than a serial sum.

// new[0] = on_hand_after + old[1]; new[k] = old[k+1]; new[W-1] = 0            (environment.py:405-412)
template <int MAXW>
NIC_HD void shifted_pipeline(const float* old_slots, int64_t ldb, int W, float on_hand_after, float (&nv)[MAXW]) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        float v = 0.f;
        if (k == 0) v = on_hand_after + old_slots[ldb];
        else if (k < W - 1) v = old_slots[(int64_t)(k + 1) * ldb];
        nv[k] = v;
    }
}

// add `a` into slot L-1 when a != 0 (the reference filters zero orders before the put, environment.py:426-432)
// (round 6: "is anything placed, and inside the pipeline" is folded into the lane's slot INDEX, -1 = nowhere, kept in a vector
// register - each slot is then one vector compare + select.  Written as `a != 0 && k == slot && k < W` the compiler keeps the
// three conditions as lane masks in scalar registers and ANDs them on the scalar unit between two vector instructions, for
// every slot of every order: the closed-form chain ran 25 % faster without those round trips)
NIC_HD int order_slot(float a, float lead, int W) {
    int slot = (int)lead - 1;
    slot = (a != 0.f && slot < W) ? slot : -1;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(slot));   // (or the optimiser folds the select back into the compares)
#endif
    return slot;
}
template <int MAXW>
NIC_HD void place_order(float (&nv)[MAXW], int W, float a, float lead) {
    const int slot = order_slot(a, lead, W);
#pragma unroll
    for (int k = 0; k < MAXW; ++k) nv[k] = (k == slot) ? nv[k] + a : nv[k];
}

template <int MAXW>
NIC_HD void store_pipeline(float* out_slots, int64_t ldb, int W, const float (&nv)[MAXW]) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k)
        if (k < W) out_slots[(int64_t)k * ldb] = nv[k];
}

template <int MAXW>
NIC_HD float pick(const float (&g)[MAXW], int W, int slot) {
    int sl = slot < W ? slot : -1;   // (one lane-local index instead of two lane masks per slot: see order_slot)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(sl));
#endif
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) r = (k == sl) ? g[k] : r;
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// One period is split into pieces that FOUR lanes per scenario execute (the "quad": lane q owns stores s = q, q+4, ...)
// plus per-warehouse and per-scenario pieces.  The pieces are NIC_HD; the device kernels (env_step.hip) run the quad's
// lanes in four wavefronts of one workgroup and exchange partial sums through LDS, the host-side test build runs the same
// pieces in a loop.  Partial sums are combined as ((p0+p1)+p2)+p3, i.e. exactly the Sum4 order described above, so splitting the
// stores over four lanes changes no bit of the result.
// Follows Simulator.step: stores (environment.py:179-234), warehouses (:236-270), echelons (:272-299).
// ------------------------------------------------------------------------------------------------------------
constexpr int kQuad = 4;

NIC_HD float combine4(float p0, float p1, float p2, float p3) { return ((p0 + p1) + p2) + p3; }

// Stores are processed kStoreBatch at a time in three phases (all loads, arithmetic + order placement, all stores): a lane
// then has kStoreBatch x (slots + 3) independent loads in flight instead of one store's worth, which is what this
// latency-bound kernel lacked (one wave per SIMD or two at B = 32k: 1.7 TB/s).  Reading store s + 4 before store s is
// written is also valid for in-place updates (different rows).  The arithmetic and its order are unchanged.
constexpr int kStoreBatch = 4;
constexpr int kSupBatch = 3;  // suppliers per store whose order / lead time are fetched in the load phase

template <int MAXW>
NIC_HD void load_pipeline(const float* slots, int64_t ldb, int W, float (&old)[MAXW]) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k) old[k] = k < W ? slots[(int64_t)k * ldb] : 0.f;
}

// shifted_pipeline on already loaded slots
template <int MAXW>
NIC_HD void shift_loaded(const float (&old)[MAXW], int W, float on_hand_after, float (&nv)[MAXW]) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        float v = 0.f;
        if (k == 0) v = on_hand_after + (MAXW > 1 ? old[1] : 0.f);
        else if (k < W - 1) v = old[k + 1 < MAXW ? k + 1 : k];
        nv[k] = v;
    }
}

// stores s = q, q+4, ...: cost + pipeline update; returns this lane's partial store cost
template <int MAXW>
NIC_HD float env_fwd_stores(const NicEnvStepIO& io, float* store_out, int64_t b, int q) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    const int nsup = d.n_warehouses > 0 ? d.n_warehouses : 1;
    float r = 0.f;
    for (int s0 = q; s0 < d.n_stores; s0 += kQuad * kStoreBatch) {
        float old[kStoreBatch][MAXW], nv[kStoreBatch][MAXW], dem[kStoreBatch], p[kStoreBatch], h[kStoreBatch];
        float ord[kStoreBatch][kSupBatch], lead[kStoreBatch][kSupBatch];  // (suppliers beyond kSupBatch are read in phase 2)
#pragma unroll
        for (int u = 0; u < kStoreBatch; ++u) {
            const int s = s0 + u * kQuad;
            if (s < d.n_stores) {
                load_pipeline<MAXW>(io.store_inv + (int64_t)s * d.store_slots * ldb + b, ldb, d.store_slots, old[u]);
                dem[u] = t2(io.demand, s, b);
                p[u] = t2(io.underage, s, b);
                h[u] = t2(io.holding, s, b);
#pragma unroll
                for (int w = 0; w < kSupBatch; ++w) {
                    if (w < nsup) {
                        ord[u][w] = t3(io.store_orders, s, w, b);
                        lead[u][w] = t3(io.lead_times, s, w, b);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kStoreBatch; ++u) {
            const int s = s0 + u * kQuad;
            if (s < d.n_stores) {
                const float on_hand = old[u][0];
                float after = on_hand - dem[u];
                float c;
                if (d.maximize_profit) {
                    c = (-p[u]) * (on_hand < dem[u] ? on_hand : dem[u]) + h[u] * relu(after);  // :191-194
                } else {
                    c = p[u] * relu(-after) + h[u] * relu(after);  // :198-201
                }
                r += c;
                if (d.lost_demand) after = relu(after);  // :204-205
                shift_loaded<MAXW>(old[u], d.store_slots, after, nv[u]);
#pragma unroll
                for (int w = 0; w < kSupBatch; ++w)
                    if (w < nsup) place_order<MAXW>(nv[u], d.store_slots, ord[u][w], lead[u][w]);
                for (int w = kSupBatch; w < nsup; ++w)
                    place_order<MAXW>(nv[u], d.store_slots, t3(io.store_orders, s, w, b), t3(io.lead_times, s, w, b));
            }
        }
#pragma unroll
        for (int u = 0; u < kStoreBatch; ++u) {
            const int s = s0 + u * kQuad;
            if (s < d.n_stores) store_pipeline<MAXW>(store_out + (int64_t)s * d.store_slots * ldb + b, ldb, d.store_slots, nv[u]);
        }
    }
    return r;
}

// this lane's share of what warehouse w ships: sum of the orders of stores s = q, q+4, ...   (:247)
// The loads of a batch are issued together and then added in the same order as before: written as `a += load` in a loop with a
// run-time trip count the compiler waits for each load before issuing the next (ISA: load, s_waitcnt vmcnt(0), add, branch) -
// 16 stores per lane x 3 warehouses = 48 dependent memory round trips per wavefront for BASELINE cfg5, most of that kernel's time.
constexpr int kShipBatch = 8;
NIC_HD float env_ship_partial(const NicEnvStepIO& io, int w, int64_t b, int q) {
    float a = 0.f;
    for (int s0 = q; s0 < io.dims.n_stores; s0 += kQuad * kShipBatch) {
        float v[kShipBatch];
#pragma unroll
        for (int u = 0; u < kShipBatch; ++u) {
            const int s = s0 + u * kQuad;
            v[u] = t3(io.store_orders, s < io.dims.n_stores ? s : q, w, b);   // (unconditional load of a valid row; masked below)
        }
#pragma unroll
        for (int u = 0; u < kShipBatch; ++u)
            if (s0 + u * kQuad < io.dims.n_stores) a += v[u];
    }
    return a;
}

// warehouse w: ships `shipped` (no feasibility clip: :249), pays holding (+ edge) cost, updates its pipeline; returns cost
template <int MAXW>
NIC_HD float env_fwd_warehouse(const NicEnvStepIO& io, float* wh_out, int w, float shipped, int64_t b) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    const float* inv = io.wh_inv + (int64_t)w * d.warehouse_slots * ldb + b;
    const float after = inv[0] - shipped;
    float c = t2(io.wh_holding, w, b) * relu(after);  // :251
    const float a = t2(io.wh_orders, w, b);
    if (io.wh_edge_costs.p) c = c + t2(io.wh_edge_costs, w, b) * a;  // :254-259
    float nv[MAXW];
    shifted_pipeline<MAXW>(inv, ldb, d.warehouse_slots, after, nv);
    place_order<MAXW>(nv, d.warehouse_slots, a, t2(io.wh_lead_times, w, b));
    store_pipeline<MAXW>(wh_out + (int64_t)w * d.warehouse_slots * ldb + b, ldb, d.warehouse_slots, nv);
    return c;
}

NIC_HD float env_wh_orders_sum(const NicEnvStepIO& io, int64_t b) {
    float s = 0.f;
    for (int w0 = 0; w0 < io.dims.n_warehouses; w0 += kShipBatch) {   // loads of a batch first, then the sum in warehouse order
        float v[kShipBatch];
#pragma unroll
        for (int u = 0; u < kShipBatch; ++u) v[u] = t2(io.wh_orders, w0 + u < io.dims.n_warehouses ? w0 + u : 0, b);
#pragma unroll
        for (int u = 0; u < kShipBatch; ++u)
            if (w0 + u < io.dims.n_warehouses) s += v[u];
    }
    return s;
}

// extra echelons: echelon e ships what echelon e+1 ordered; the last one feeds the warehouses (:282-285); returns cost
template <int MAXW>
NIC_HD float env_fwd_echelons(const NicEnvStepIO& io, float* ech_out, float wh_orders_sum, int64_t b) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    float r_e = 0.f;
    for (int e = 0; e < d.n_echelons; ++e) {
        const float ship = (e < d.n_echelons - 1) ? t2(io.ech_orders, e + 1, b) : wh_orders_sum;
        const float* inv = io.ech_inv + (int64_t)e * d.echelon_slots * ldb + b;
        const float after = inv[0] - ship;
        r_e += t2(io.ech_holding, e, b) * relu(after);  // :287
        float nv[MAXW];
        shifted_pipeline<MAXW>(inv, ldb, d.echelon_slots, after, nv);
        place_order<MAXW>(nv, d.echelon_slots, t2(io.ech_orders, e, b), t2(io.ech_lead_times, e, b));
        store_pipeline<MAXW>(ech_out + (int64_t)e * d.echelon_slots * ldb + b, ldb, d.echelon_slots, nv);
    }
    return r_e;
}

// reference composition of the pieces for one scenario (host-side test build; the device kernels do the same through LDS)
template <int MAXW>
NIC_HD void env_step_fwd_scenario(const NicEnvStepIO& io, float* store_out, float* wh_out, float* ech_out,
                                  float* reward, int64_t b) {
    const NicEnvDims& d = io.dims;
    float rq[kQuad];
    for (int q = 0; q < kQuad; ++q) rq[q] = env_fwd_stores<MAXW>(io, store_out, b, q);
    float total = combine4(rq[0], rq[1], rq[2], rq[3]);
    if (d.n_warehouses > 0) {
        float r_wh = 0.f;
        for (int w = 0; w < d.n_warehouses; ++w) {
            const float shipped = combine4(env_ship_partial(io, w, b, 0), env_ship_partial(io, w, b, 1),
                                           env_ship_partial(io, w, b, 2), env_ship_partial(io, w, b, 3));
            r_wh += env_fwd_warehouse<MAXW>(io, wh_out, w, shipped, b);
        }
        total += r_wh;
    }
    if (d.n_echelons > 0) total += env_fwd_echelons<MAXW>(io, ech_out, env_wh_orders_sum(io, b), b);
    reward[b] = total;
}

// gradient w.r.t. the old pipeline given the gradient of the new one and of the post-demand on-hand:
// old[0] <- g_after (through on_hand_after), old[1] <- g_new[0], old[k] <- g_new[k-1] (k >= 2)
template <int MAXW>
NIC_HD void store_pipeline_grad(float* g_in, int64_t ldb, int W, const float (&gn)[MAXW], float g_on_hand) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        if (k < W) {
            float v;
            if (k == 0) v = g_on_hand;
            else if (k == 1) v = gn[0];
            else v = gn[k - 1];
            g_in[(int64_t)k * ldb] = v;
        }
    }
}

template <int MAXW>
NIC_HD void load_grad(const float* g_out, int64_t ldb, int W, float (&gn)[MAXW]) {
#pragma unroll
    for (int k = 0; k < MAXW; ++k) gn[k] = (g_out != nullptr && k < W) ? g_out[(int64_t)k * ldb] : 0.f;
}

// ------------------------------------------------------------------------------------------------------------
// Backward pieces (what autograd derives for the ops cited above).  Tie rules of torch 2.x: clamp(min=0) passes the
// gradient where x >= 0; minimum splits it 0.5/0.5 on ties; orders that are exactly 0 get no gradient through the
// pipeline placement but still through the warehouse / echelon outflow sums.
// ------------------------------------------------------------------------------------------------------------

// echelons, most upstream first (their outflow gradient lands on downstream orders); returns d/d(sum_w wh_orders)
template <int MAXW>
NIC_HD float env_bwd_echelons(const NicEnvStepIO& io, const float* g_ech_out, float gr, float* g_ech_in,
                              float* g_ech_orders, int64_t b) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    const float wh_orders_sum = env_wh_orders_sum(io, b);
    float prev_g_after = 0.f;
    for (int e = 0; e < d.n_echelons; ++e) {
        const float ship = (e < d.n_echelons - 1) ? t2(io.ech_orders, e + 1, b) : wh_orders_sum;
        const float* inv = io.ech_inv + (int64_t)e * d.echelon_slots * ldb + b;
        const float after = inv[0] - ship;
        float gn[MAXW];
        load_grad<MAXW>(g_ech_out ? g_ech_out + (int64_t)e * d.echelon_slots * ldb + b : nullptr, ldb, d.echelon_slots, gn);
        const float a = t2(io.ech_orders, e, b);
        const float ga = (a != 0.f) ? pick<MAXW>(gn, d.echelon_slots, (int)t2(io.ech_lead_times, e, b) - 1) : 0.f;
        float g_after = gn[0];
        if (after >= 0.f) g_after += gr * t2(io.ech_holding, e, b);
        store_pipeline_grad<MAXW>(g_ech_in + (int64_t)e * d.echelon_slots * ldb + b, ldb, d.echelon_slots, gn, g_after);
        g_ech_orders[(int64_t)e * ldb + b] = ga - prev_g_after;  // echelon e's own order is shipped by echelon e-1
        prev_g_after = g_after;
    }
    return -prev_g_after;
}

// warehouse w; returns the gradient w.r.t. its post-shipping on-hand (every store order from it carries -that)
template <int MAXW>
NIC_HD float env_bwd_warehouse(const NicEnvStepIO& io, const float* g_wh_out, float gr, float g_to_wh_orders, int w,
                               float shipped, float* g_wh_in, float* g_wh_orders, int64_t b) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    const float* inv = io.wh_inv + (int64_t)w * d.warehouse_slots * ldb + b;
    const float after = inv[0] - shipped;
    float gn[MAXW];
    load_grad<MAXW>(g_wh_out ? g_wh_out + (int64_t)w * d.warehouse_slots * ldb + b : nullptr, ldb, d.warehouse_slots, gn);
    const float a = t2(io.wh_orders, w, b);
    float ga = (a != 0.f) ? pick<MAXW>(gn, d.warehouse_slots, (int)t2(io.wh_lead_times, w, b) - 1) : 0.f;
    if (io.wh_edge_costs.p) ga += gr * t2(io.wh_edge_costs, w, b);
    ga += g_to_wh_orders;
    g_wh_orders[(int64_t)w * ldb + b] = ga;
    float g_after = gn[0];
    if (after >= 0.f) g_after += gr * t2(io.wh_holding, w, b);
    store_pipeline_grad<MAXW>(g_wh_in + (int64_t)w * d.warehouse_slots * ldb + b, ldb, d.warehouse_slots, gn, g_after);
    return g_after;
}

// stores s = q, q+4, ...; g_wafter(w) = gradient of warehouse w's post-shipping on-hand (unused when n_warehouses == 0)
template <int MAXW, typename GWAfter>
NIC_HD void env_bwd_stores(const NicEnvStepIO& io, const float* g_store_out, float gr, GWAfter g_wafter,
                           float* g_store_in, float* g_store_orders, int64_t b, int q) {
    const NicEnvDims& d = io.dims;
    const int64_t ldb = d.ldb;
    const int nsup = d.n_warehouses > 0 ? d.n_warehouses : 1;
    // same three-phase batching as env_fwd_stores
    for (int s0 = q; s0 < d.n_stores; s0 += kQuad * kStoreBatch) {
        float gn[kStoreBatch][MAXW], on_hand[kStoreBatch], dem[kStoreBatch], p[kStoreBatch], h[kStoreBatch];
        float ord[kStoreBatch][kSupBatch], lead[kStoreBatch][kSupBatch];
#pragma unroll
        for (int u = 0; u < kStoreBatch; ++u) {
            const int s = s0 + u * kQuad;
            if (s < d.n_stores) {
                on_hand[u] = io.store_inv[(int64_t)s * d.store_slots * ldb + b];
                dem[u] = t2(io.demand, s, b);
                p[u] = t2(io.underage, s, b);
                h[u] = t2(io.holding, s, b);
#pragma unroll
                for (int w = 0; w < kSupBatch; ++w) {
                    if (w < nsup) {
                        ord[u][w] = t3(io.store_orders, s, w, b);
                        lead[u][w] = t3(io.lead_times, s, w, b);
                    }
                }
                load_grad<MAXW>(g_store_out ? g_store_out + (int64_t)s * d.store_slots * ldb + b : nullptr, ldb, d.store_slots,
                                gn[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kStoreBatch; ++u) {
            const int s = s0 + u * kQuad;
            if (s >= d.n_stores) continue;
            const float after = on_hand[u] - dem[u];
            // through the carried inventory (lost demand clips it at 0: gradient where after >= 0)
            float g_after = gn[u][0];
            if (d.lost_demand && !(after >= 0.f)) g_after = 0.f;
            float g_on_hand;
            if (d.maximize_profit) {
                if (after >= 0.f) g_after += gr * h[u];
                const float share = on_hand[u] < dem[u] ? 1.f : (on_hand[u] == dem[u] ? 0.5f : 0.f);  // minimum() tie rule
                g_on_hand = g_after + gr * (-p[u]) * share;
            } else {
                float gc = 0.f;
                if (-after >= 0.f) gc += -p[u];  // d/d(after) of p*clamp(-after, 0)
                if (after >= 0.f) gc += h[u];    // d/d(after) of h*clamp(after, 0)
                g_on_hand = g_after + gr * gc;
            }
            store_pipeline_grad<MAXW>(g_store_in + (int64_t)s * d.store_slots * ldb + b, ldb, d.store_slots, gn[u], g_on_hand);
#pragma unroll
            for (int w = 0; w < kSupBatch; ++w) {
                if (w < nsup) {
                    float ga = (ord[u][w] != 0.f) ? pick<MAXW>(gn[u], d.store_slots, (int)lead[u][w] - 1) : 0.f;
                    if (d.n_warehouses > 0) ga += -g_wafter(w);  // the outflow sum has no zero filter (:247)
                    g_store_orders[((int64_t)s * nsup + w) * ldb + b] = ga;
                }
            }
            for (int w = kSupBatch; w < nsup; ++w) {
                const float a = t3(io.store_orders, s, w, b);
                float ga = (a != 0.f) ? pick<MAXW>(gn[u], d.store_slots, (int)t3(io.lead_times, s, w, b) - 1) : 0.f;
                if (d.n_warehouses > 0) ga += -g_wafter(w);
                g_store_orders[((int64_t)s * nsup + w) * ldb + b] = ga;
            }
        }
    }
}

// reference composition for one scenario (host-side test build)
template <int MAXW>
NIC_HD void env_step_bwd_scenario(const NicEnvStepIO& io, const float* g_store_out, const float* g_wh_out,
                                  const float* g_ech_out, const NicTable2& g_reward, float* g_store_in,
                                  float* g_wh_in, float* g_ech_in, float* g_store_orders, float* g_wh_orders,
                                  float* g_ech_orders, int64_t b) {
    const NicEnvDims& d = io.dims;
    const float gr = g_reward.p[b * g_reward.scn_stride];
    float g_to_wh_orders = 0.f;
    if (d.n_echelons > 0) g_to_wh_orders = env_bwd_echelons<MAXW>(io, g_ech_out, gr, g_ech_in, g_ech_orders, b);
    float gwa[NIC_MAX_WAREHOUSES];
    for (int w = 0; w < d.n_warehouses; ++w) {
        const float shipped = combine4(env_ship_partial(io, w, b, 0), env_ship_partial(io, w, b, 1),
                                       env_ship_partial(io, w, b, 2), env_ship_partial(io, w, b, 3));
        gwa[w] = env_bwd_warehouse<MAXW>(io, g_wh_out, gr, g_to_wh_orders, w, shipped, g_wh_in, g_wh_orders, b);
    }
    for (int q = 0; q < kQuad; ++q)
        env_bwd_stores<MAXW>(io, g_store_out, gr, [&](int w) { return gwa[w]; }, g_store_in, g_store_orders, b, q);
}

// ------------------------------------------------------------------------------------------------------------
// ACCESSOR-GENERIC one-location bodies (round 4).  The whole-horizon kernels (csrc/horizon_rollout.hip) keep the state of 16
// scenarios in LDS and give every (scenario, store) pair its own lane; their blocks are addressed by 32-bit LDS offsets with a
// compile-time row stride, not by NicEnvStepIO's (pointer, stride, stride) tables.  (Measured: handing the io-based bodies LDS
// pointers runs at the same speed - the compiler folds the constant strides - so this is about saying what the kernel does, not
// about time.)  These templates are the SAME arithmetic as one store of env_fwd_stores / env_bwd_stores and as
// env_fwd_warehouse / env_bwd_warehouse, written against an accessor `A` of one scenario:
//   sizes / flags: S() Wn() nsup() Ws() Ww() lost() profit() has_edge()
//   reads:  inv(s,k) dem(s) under(s) hold(s) ord(s,w) lead(s,w) | wh_inv(w,k) wh_hold(w) wh_lead(w) wh_edge(w) wh_ord(w)
//   writes: put_inv(s,k,v) put_wh(w,k,v)                                                 (the state after the period)
//   adjoint: g_out(s,k) gwh_out(w,k) reads; put_g_in(s,k,v) put_g_ord(s,w,v) put_gwh_in(w,k,v) put_gwh_ord(w,v) writes
// IoAccess adapts a NicEnvStepIO (tests/hostsim composes a period from them and compares it with the quad composition bit for bit).
// ------------------------------------------------------------------------------------------------------------
template <int MAXW, class A>
NIC_HD float env_fwd_store_t(const A& a, int s) {
    const int W = a.Ws(), nsup = a.nsup();
    float old[MAXW], nv[MAXW], ord[kSupBatch], lead[kSupBatch];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) old[k] = k < W ? a.inv(s, k) : 0.f;
    const float dem = a.dem(s), p = a.under(s), h = a.hold(s);
#pragma unroll
    for (int w = 0; w < kSupBatch; ++w) {
        if (w < nsup) {
            ord[w] = a.ord(s, w);
            lead[w] = a.lead(s, w);
        }
    }
    const float on_hand = old[0];
    float after = on_hand - dem;
    float c;
    if (a.profit()) {
        c = (-p) * (on_hand < dem ? on_hand : dem) + h * relu(after);  // :191-194
    } else {
        c = p * relu(-after) + h * relu(after);  // :198-201
    }
    if (a.lost()) after = relu(after);  // :204-205
    shift_loaded<MAXW>(old, W, after, nv);
#pragma unroll
    for (int w = 0; w < kSupBatch; ++w)
        if (w < nsup) place_order<MAXW>(nv, W, ord[w], lead[w]);
    for (int w = kSupBatch; w < nsup; ++w) place_order<MAXW>(nv, W, a.ord(s, w), a.lead(s, w));
#pragma unroll
    for (int k = 0; k < MAXW; ++k)
        if (k < W) a.put_inv(s, k, nv[k]);
    return c;
}

// what warehouse w ships = sum of its stores' orders in the Sum4 order (env_ship_partial x 4 + combine4) by ONE lane: sixteen
// orders per batch into the four interleaved accumulators (adding +0 for a store past S changes nothing)
template <class A>
NIC_HD float env_shipped_t(const A& a, int w) {
    const int S = a.S();
    float acc[kQuad] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = a.ord(s0 + u < S ? s0 + u : S - 1, w);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (s0 + u < S) acc[u & 3] += v[u];
    }
    return combine4(acc[0], acc[1], acc[2], acc[3]);
}

template <int MAXW, class A>
NIC_HD float env_fwd_warehouse_t(const A& a, int w, float shipped) {
    const int W = a.Ww();
    float old[MAXW], nv[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) old[k] = k < W ? a.wh_inv(w, k) : 0.f;
    const float after = old[0] - shipped;
    float c = a.wh_hold(w) * relu(after);  // :251
    const float o = a.wh_ord(w);
    if (a.has_edge()) c = c + a.wh_edge(w) * o;  // :254-259
    shift_loaded<MAXW>(old, W, after, nv);
    place_order<MAXW>(nv, W, o, a.wh_lead(w));
#pragma unroll
    for (int k = 0; k < MAXW; ++k)
        if (k < W) a.put_wh(w, k, nv[k]);
    return c;
}

// gradient of warehouse w's post-shipping on-hand (what env_bwd_warehouse returns), without side effects
template <class A>
NIC_HD float env_bwd_wh_g_after_t(const A& a, float gr, int w, float shipped) {
    const float after = a.wh_inv(w, 0) - shipped;
    float g_after = a.gwh_out(w, 0);
    if (after >= 0.f) g_after += gr * a.wh_hold(w);
    return g_after;
}

// (no extra echelons: the gradient from an echelon chain into the warehouse orders, g_to_wh_orders of env_bwd_warehouse, is zero)
template <int MAXW, class A>
NIC_HD float env_bwd_warehouse_t(const A& a, float gr, int w, float shipped) {
    const int W = a.Ww();
    float gn[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) gn[k] = k < W ? a.gwh_out(w, k) : 0.f;
    const float after = a.wh_inv(w, 0) - shipped;
    const float o = a.wh_ord(w);
    float ga = (o != 0.f) ? pick<MAXW>(gn, W, (int)a.wh_lead(w) - 1) : 0.f;
    if (a.has_edge()) ga += gr * a.wh_edge(w);
    ga += 0.f;
    a.put_gwh_ord(w, ga);
    float g_after = gn[0];
    if (after >= 0.f) g_after += gr * a.wh_hold(w);
#pragma unroll
    for (int k = 0; k < MAXW; ++k)
        if (k < W) a.put_gwh_in(w, k, k == 0 ? g_after : (k == 1 ? gn[0] : gn[k - 1]));
    return g_after;
}

template <int MAXW, class A, class GWAfter>
NIC_HD void env_bwd_store_t(const A& a, float gr, GWAfter g_wafter, int s) {
    const int W = a.Ws(), nsup = a.nsup();
    float gn[MAXW], ord[kSupBatch], lead[kSupBatch];
    const float on_hand = a.inv(s, 0);
    const float dem = a.dem(s), p = a.under(s), h = a.hold(s);
#pragma unroll
    for (int w = 0; w < kSupBatch; ++w) {
        if (w < nsup) {
            ord[w] = a.ord(s, w);
            lead[w] = a.lead(s, w);
        }
    }
#pragma unroll
    for (int k = 0; k < MAXW; ++k) gn[k] = k < W ? a.g_out(s, k) : 0.f;
    const float after = on_hand - dem;
    float g_after = gn[0];
    if (a.lost() && !(after >= 0.f)) g_after = 0.f;
    float g_on_hand;
    if (a.profit()) {
        if (after >= 0.f) g_after += gr * h;
        const float share = on_hand < dem ? 1.f : (on_hand == dem ? 0.5f : 0.f);  // minimum() tie rule
        g_on_hand = g_after + gr * (-p) * share;
    } else {
        float gc = 0.f;
        if (-after >= 0.f) gc += -p;
        if (after >= 0.f) gc += h;
        g_on_hand = g_after + gr * gc;
    }
#pragma unroll
    for (int k = 0; k < MAXW; ++k)
        if (k < W) a.put_g_in(s, k, k == 0 ? g_on_hand : (k == 1 ? gn[0] : gn[k - 1]));
#pragma unroll
    for (int w = 0; w < kSupBatch; ++w) {
        if (w < nsup) {
            float ga = (ord[w] != 0.f) ? pick<MAXW>(gn, W, (int)lead[w] - 1) : 0.f;
            if (a.Wn() > 0) ga += -g_wafter(w);  // the outflow sum has no zero filter (:247)
            a.put_g_ord(s, w, ga);
        }
    }
    for (int w = kSupBatch; w < nsup; ++w) {
        const float o = a.ord(s, w);
        float ga = (o != 0.f) ? pick<MAXW>(gn, W, (int)a.lead(s, w) - 1) : 0.f;
        if (a.Wn() > 0) ga += -g_wafter(w);
        a.put_g_ord(s, w, ga);
    }
}

// NicEnvStepIO of one scenario as such an accessor (host-side test build)
struct IoAccess {
    const NicEnvStepIO& io;
    int64_t b;
    float* store_out;
    float* wh_out;
    const float* g_store_out;
    const float* g_wh_out;
    float* g_store_in;
    float* g_wh_in;
    float* g_store_orders;
    float* g_wh_orders;
    NIC_HD int S() const { return io.dims.n_stores; }
    NIC_HD int Wn() const { return io.dims.n_warehouses; }
    NIC_HD int nsup() const { return io.dims.n_warehouses > 0 ? io.dims.n_warehouses : 1; }
    NIC_HD int Ws() const { return io.dims.store_slots; }
    NIC_HD int Ww() const { return io.dims.warehouse_slots; }
    NIC_HD bool lost() const { return io.dims.lost_demand != 0; }
    NIC_HD bool profit() const { return io.dims.maximize_profit != 0; }
    NIC_HD bool has_edge() const { return io.wh_edge_costs.p != nullptr; }
    NIC_HD int64_t si(int s, int k) const { return ((int64_t)s * io.dims.store_slots + k) * io.dims.ldb + b; }
    NIC_HD int64_t wi(int w, int k) const { return ((int64_t)w * io.dims.warehouse_slots + k) * io.dims.ldb + b; }
    NIC_HD float inv(int s, int k) const { return io.store_inv[si(s, k)]; }
    NIC_HD float dem(int s) const { return t2(io.demand, s, b); }
    NIC_HD float under(int s) const { return t2(io.underage, s, b); }
    NIC_HD float hold(int s) const { return t2(io.holding, s, b); }
    NIC_HD float ord(int s, int w) const { return t3(io.store_orders, s, w, b); }
    NIC_HD float lead(int s, int w) const { return t3(io.lead_times, s, w, b); }
    NIC_HD void put_inv(int s, int k, float v) const { store_out[si(s, k)] = v; }
    NIC_HD float wh_inv(int w, int k) const { return io.wh_inv[wi(w, k)]; }
    NIC_HD float wh_hold(int w) const { return t2(io.wh_holding, w, b); }
    NIC_HD float wh_lead(int w) const { return t2(io.wh_lead_times, w, b); }
    NIC_HD float wh_edge(int w) const { return t2(io.wh_edge_costs, w, b); }
    NIC_HD float wh_ord(int w) const { return t2(io.wh_orders, w, b); }
    NIC_HD void put_wh(int w, int k, float v) const { wh_out[wi(w, k)] = v; }
    NIC_HD float g_out(int s, int k) const { return g_store_out ? g_store_out[si(s, k)] : 0.f; }
    NIC_HD float gwh_out(int w, int k) const { return g_wh_out ? g_wh_out[wi(w, k)] : 0.f; }
    NIC_HD void put_g_in(int s, int k, float v) const { g_store_in[si(s, k)] = v; }
    NIC_HD void put_gwh_in(int w, int k, float v) const { g_wh_in[wi(w, k)] = v; }
    NIC_HD void put_g_ord(int s, int w, float v) const { g_store_orders[((int64_t)s * nsup() + w) * io.dims.ldb + b] = v; }
    NIC_HD void put_gwh_ord(int w, float v) const { g_wh_orders[(int64_t)w * io.dims.ldb + b] = v; }
};

}  // namespace nic
