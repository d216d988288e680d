// One period of inventory dynamics on gfx950: forward and analytic backward.
//
// Kernel shape: FOUR lanes per scenario (a "quad": lane q owns stores q, q+4, ... and warehouses q, q+4, ...), laid out
// as 4 wavefronts x 64 scenarios per workgroup, so every global access of a wave is still a contiguous 256-byte row
// segment of a scenario-minor buffer while a 64-store scenario is walked by four lanes (one lane per scenario left the
// chip latency-bound: 0.22 ms for 32k x 64 stores against a 27 us byte floor).  Per-scenario reductions cross the quad
// through LDS in a fixed order (bit-identical to the single-lane Sum4 order).  HBM-bound: algorithmic bytes per
// scenario-period are 4*[2*(S*Ws + Wn*Ww + E*We) + S + (S*max(Wn,1) + Wn + E) + 1] (state read + write, demand, orders,
// reward).  The arithmetic lives in env_step_body.h (shared with the host-side test build).
#include "env_step_body.h"
#include "nic_common.h"

namespace {

// Workgroup = 64 scenarios x 4 lanes ("quad") = 4 wavefronts: wave q handles stores s = q, q+4, ... of its 64 scenarios,
// so a 64-store scenario is walked by four lanes instead of one (B = 32k scenarios alone would give the 1,024 SIMDs half a
// wave each).  Per-scenario reductions (store costs, what each warehouse ships) go through LDS in the fixed Sum4 order;
// then lane q handles warehouses w = q, q+4, ... and lane 0 the serial echelon chain.
constexpr int kLanes = 64;
constexpr int kChunk = 8;  // warehouses whose shipment partials are exchanged per barrier round

template <int MAXW>
__global__ __launch_bounds__(kLanes * nic::kQuad) void env_step_fwd_kernel(NicEnvStepIO io, float* __restrict__ store_out,
                                                                            float* __restrict__ wh_out,
                                                                            float* __restrict__ ech_out,
                                                                            float* __restrict__ reward, int zero_lead_upstream) {
    __shared__ float part[kChunk][nic::kQuad][kLanes];  // shipment partials of one warehouse chunk
    __shared__ float rq[nic::kQuad][kLanes];            // store-cost partials
    __shared__ float cw[NIC_MAX_WAREHOUSES][kLanes];    // warehouse costs
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < io.dims.n_scenarios;
    const int Wn = io.dims.n_warehouses;

    rq[q][x] = live ? nic::env_fwd_stores<MAXW>(io, store_out, b, q) : 0.f;
    for (int wc = 0; wc < Wn; wc += kChunk) {
        for (int i = 0; i < kChunk && wc + i < Wn; ++i) part[i][q][x] = live ? nic::env_ship_partial(io, wc + i, b, q) : 0.f;
        __syncthreads();
        for (int i = q; i < kChunk && wc + i < Wn; i += nic::kQuad) {
            const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
            cw[wc + i][x] = live ? nic::env_fwd_warehouse<MAXW>(io, wh_out, wc + i, shipped, b) : 0.f;
        }
        __syncthreads();
    }
    if (Wn == 0) __syncthreads();
    if (q == 0 && live) {
        float total = nic::combine4(rq[0][x], rq[1][x], rq[2][x], rq[3][x]);
        if (Wn > 0) {
            float r_wh = 0.f;
            for (int w = 0; w < Wn; ++w) r_wh += cw[w][x];
            total += r_wh;
        }
        if (io.dims.n_echelons > 0) total += nic::env_fwd_echelons<MAXW>(io, ech_out, nic::env_wh_orders_sum(io, b), b);
        reward[b] = total;
    }
    // The reference's treatment of an order whose lead time is 0 (environment.py:405-432: its flat index is lead - 1 = -1 from
    // the start of the store's pipeline, i.e. the LAST slot of the store in front - for store 0 the last store of the scenario
    // in front, scenario B - 1 for scenario 0 - through torch's put with wrapped negative indices).  The launch above dropped
    // such orders; with zero_lead_upstream each scenario adds what lands in ITS rows: the orders of its own store s + 1 into
    // the last slot of store s, and the NEXT scenario's store-0 orders into the last slot of its last store (inputs of another
    // workgroup, never its outputs: no race).  The barrier above made this workgroup's new state visible to all its lanes.
    if (zero_lead_upstream && live) {
        const NicEnvDims& d = io.dims;
        const int nsup = d.n_warehouses > 0 ? d.n_warehouses : 1;
        for (int s = q; s < d.n_stores; s += nic::kQuad) {
            const int64_t bs = s > 0 ? b : (b + 1 < d.n_scenarios ? b + 1 : 0);
            float u = 0.f;
            for (int w = 0; w < nsup; ++w) {
                const float a = nic::t3(io.store_orders, s, w, bs);
                if (nic::t3(io.lead_times, s, w, bs) == 0.f && a != 0.f) u += a;
            }
            const int target = s > 0 ? s - 1 : d.n_stores - 1;
            float* slot = store_out + ((int64_t)target * d.store_slots + d.store_slots - 1) * d.ldb + b;
            *slot = *slot + u;
        }
    }
}

template <int MAXW>
__global__ __launch_bounds__(kLanes * nic::kQuad) void env_step_bwd_kernel(
    NicEnvStepIO io, const float* __restrict__ g_store_out, const float* __restrict__ g_wh_out,
    const float* __restrict__ g_ech_out, NicTable2 g_reward, float* __restrict__ g_store_in, float* __restrict__ g_wh_in,
    float* __restrict__ g_ech_in, float* __restrict__ g_store_orders, float* __restrict__ g_wh_orders,
    float* __restrict__ g_ech_orders, int zero_lead_upstream) {
    __shared__ float part[kChunk][nic::kQuad][kLanes];
    __shared__ float gwa[NIC_MAX_WAREHOUSES][kLanes];  // gradient of each warehouse's post-shipping on-hand
    __shared__ float g2wh[kLanes];                     // d/d(sum_w wh_orders) from the echelon chain
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < io.dims.n_scenarios;
    const int Wn = io.dims.n_warehouses;
    const float gr = live ? g_reward.p[b * g_reward.scn_stride] : 0.f;

    if (q == 0)
        g2wh[x] = (live && io.dims.n_echelons > 0) ? nic::env_bwd_echelons<MAXW>(io, g_ech_out, gr, g_ech_in, g_ech_orders, b)
                                                    : 0.f;
    for (int wc = 0; wc < Wn; wc += kChunk) {
        for (int i = 0; i < kChunk && wc + i < Wn; ++i) part[i][q][x] = live ? nic::env_ship_partial(io, wc + i, b, q) : 0.f;
        __syncthreads();  // also publishes g2wh in the first round
        for (int i = q; i < kChunk && wc + i < Wn; i += nic::kQuad) {
            const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
            gwa[wc + i][x] = live ? nic::env_bwd_warehouse<MAXW>(io, g_wh_out, gr, g2wh[x], wc + i, shipped, g_wh_in,
                                                                 g_wh_orders, b)
                                  : 0.f;
        }
        __syncthreads();
    }
    if (live)
        nic::env_bwd_stores<MAXW>(io, g_store_out, gr, [&](int w) { return gwa[w][x]; }, g_store_in, g_store_orders, b, q);
    // adjoint of the zero-lead rule (see the forward kernel): such an order's gradient is the gradient of the slot it was added
    // to - the last slot of the store in front (store 0: of the PREVIOUS scenario's last store) - where the order is not 0.  The
    // lane that wrote g_store_orders[s][.][b] above (lane q owns stores q, q + 4, ...) adds to it.
    if (zero_lead_upstream && live && g_store_out) {
        const NicEnvDims& d = io.dims;
        const int nsup = d.n_warehouses > 0 ? d.n_warehouses : 1;
        for (int s = q; s < d.n_stores; s += nic::kQuad) {
            const int target = s > 0 ? s - 1 : d.n_stores - 1;
            const int64_t bt = s > 0 ? b : (b > 0 ? b - 1 : d.n_scenarios - 1);
            const float g = g_store_out[((int64_t)target * d.store_slots + d.store_slots - 1) * d.ldb + bt];
            for (int w = 0; w < nsup; ++w)
                if (nic::t3(io.lead_times, s, w, b) == 0.f && nic::t3(io.store_orders, s, w, b) != 0.f)
                    g_store_orders[((int64_t)s * nsup + w) * d.ldb + b] += g;
        }
    }
}

int validate(const NicEnvStepIO* io, const char* who) {
    NIC_REQUIRE(io != nullptr, "%s: io is null", who);
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(d.n_scenarios > 0 && d.ldb >= d.n_scenarios, "%s: bad n_scenarios/ldb (%d/%d)", who, d.n_scenarios, d.ldb);
    NIC_REQUIRE(d.n_stores > 0, "%s: n_stores must be positive", who);
    NIC_REQUIRE(d.n_warehouses >= 0 && d.n_echelons >= 0, "%s: negative location count", who);
    NIC_REQUIRE(d.n_warehouses <= NIC_MAX_WAREHOUSES, "%s: at most %d warehouses", who, NIC_MAX_WAREHOUSES);
    NIC_REQUIRE(d.store_slots >= 2 && d.store_slots <= NIC_MAX_SLOTS, "%s: store pipeline length %d outside [2,%d]", who,
                d.store_slots, NIC_MAX_SLOTS);
    NIC_REQUIRE(d.n_warehouses == 0 || (d.warehouse_slots >= 2 && d.warehouse_slots <= NIC_MAX_SLOTS),
                "%s: warehouse pipeline length %d outside [2,%d]", who, d.warehouse_slots, NIC_MAX_SLOTS);
    NIC_REQUIRE(d.n_echelons == 0 || (d.echelon_slots >= 2 && d.echelon_slots <= NIC_MAX_SLOTS),
                "%s: echelon pipeline length %d outside [2,%d]", who, d.echelon_slots, NIC_MAX_SLOTS);
    NIC_REQUIRE(d.n_echelons == 0 || d.n_warehouses > 0, "%s: echelons need a warehouse to feed (environment.py:283)", who);
    NIC_REQUIRE(io->store_inv && io->demand.p && io->store_orders.p && io->underage.p && io->holding.p && io->lead_times.p,
                "%s: null store buffer", who);
    NIC_REQUIRE(d.n_warehouses == 0 || (io->wh_inv && io->wh_orders.p && io->wh_holding.p && io->wh_lead_times.p),
                "%s: null warehouse buffer", who);
    NIC_REQUIRE(d.n_echelons == 0 || (io->ech_inv && io->ech_orders.p && io->ech_holding.p && io->ech_lead_times.p),
                "%s: null echelon buffer", who);
    return 0;
}

int max_slots(const NicEnvDims& d) {
    int m = d.store_slots;
    if (d.n_warehouses > 0 && d.warehouse_slots > m) m = d.warehouse_slots;
    if (d.n_echelons > 0 && d.echelon_slots > m) m = d.echelon_slots;
    return m;
}

}  // namespace

extern "C" {

int nic_env_step_fwd(const NicEnvStepIO* io, float* store_inv_out, float* wh_inv_out, float* ech_inv_out, float* reward,
                     int32_t zero_lead_upstream, void* stream) {
    if (int e = validate(io, "nic_env_step_fwd")) return e;
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(store_inv_out && reward, "nic_env_step_fwd: null output");
    NIC_REQUIRE(d.n_warehouses == 0 || wh_inv_out, "nic_env_step_fwd: null warehouse output");
    NIC_REQUIRE(d.n_echelons == 0 || ech_inv_out, "nic_env_step_fwd: null echelon output");
    const dim3 grid(nic::ceil_div(d.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(d);
    nic::note_kernelf("env_step_fwd_kernel<%d>", m <= 4 ? 4 : (m <= 8 ? 8 : NIC_MAX_SLOTS));
    if (m <= 4)
        hipLaunchKernelGGL(env_step_fwd_kernel<4>, grid, block, 0, s, *io, store_inv_out, wh_inv_out, ech_inv_out, reward, zero_lead_upstream);
    else if (m <= 8)
        hipLaunchKernelGGL(env_step_fwd_kernel<8>, grid, block, 0, s, *io, store_inv_out, wh_inv_out, ech_inv_out, reward, zero_lead_upstream);
    else
        hipLaunchKernelGGL(env_step_fwd_kernel<NIC_MAX_SLOTS>, grid, block, 0, s, *io, store_inv_out, wh_inv_out,
                           ech_inv_out, reward, zero_lead_upstream);
    return nic::check_launch("nic_env_step_fwd");
}

int nic_env_step_bwd(const NicEnvStepIO* io, const float* g_store_out, const float* g_wh_out, const float* g_ech_out,
                     NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_ech_in, float* g_store_orders,
                     float* g_wh_orders, float* g_ech_orders, int32_t zero_lead_upstream, void* stream) {
    if (int e = validate(io, "nic_env_step_bwd")) return e;
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(g_reward.p, "nic_env_step_bwd: null g_reward");
    NIC_REQUIRE(g_store_in && g_store_orders, "nic_env_step_bwd: null store gradient output");
    NIC_REQUIRE(d.n_warehouses == 0 || (g_wh_in && g_wh_orders), "nic_env_step_bwd: null warehouse gradient output");
    NIC_REQUIRE(d.n_echelons == 0 || (g_ech_in && g_ech_orders), "nic_env_step_bwd: null echelon gradient output");
    const dim3 grid(nic::ceil_div(d.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(d);
    nic::note_kernelf("env_step_bwd_kernel<%d>", m <= 4 ? 4 : (m <= 8 ? 8 : NIC_MAX_SLOTS));
#define NIC_LAUNCH_BWD(MW)                                                                                              \
    hipLaunchKernelGGL(env_step_bwd_kernel<MW>, grid, block, 0, s, *io, g_store_out, g_wh_out, g_ech_out, g_reward,      \
                       g_store_in, g_wh_in, g_ech_in, g_store_orders, g_wh_orders, g_ech_orders, zero_lead_upstream)
    if (m <= 4) NIC_LAUNCH_BWD(4);
    else if (m <= 8) NIC_LAUNCH_BWD(8);
    else NIC_LAUNCH_BWD(NIC_MAX_SLOTS);
#undef NIC_LAUNCH_BWD
    return nic::check_launch("nic_env_step_bwd");
}
}
