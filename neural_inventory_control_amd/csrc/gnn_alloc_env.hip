// GNN policy: proportional allocation head + one period of inventory dynamics in ONE launch per direction (round 4) - what
// `nic_gnn_alloc_fwd` followed by `nic_env_step_fwd` do (neural_networks.py:1435-1492 -> environment.py:110-299), and the adjoints
// in reverse order.  At the batch the reference ships for this policy (1,024 scenarios, gnn.yml) a period's ~21 launches are
// dependent-launch latency, not work: every launch that goes is ~5 % of the step.
// Layout: the env step's (64 scenarios x 4 quad lanes per workgroup); the allocation - one lane per scenario - runs on the q = 0
// lanes in front of it (forward) / behind it (backward), with a full workgroup barrier in between: the orders (order gradients)
// go through global memory and are read by OTHER lanes of the workgroup.  Arithmetic: the bodies of gnn_alloc_body.h and
// env_step_body.h, unchanged - bit-identical to the two launches.  One supplying warehouse (the groups form keeps two launches).
#include "env_step_body.h"
#include "gnn_alloc_body.h"
#include "nic_common.h"

namespace {
constexpr int kLanes = 64;
constexpr int kChunk = 8;

template <int MAXW>
__global__ __launch_bounds__(kLanes * nic::kQuad) void gnn_alloc_env_fwd_kernel(NicEnvStepIO io, const float* __restrict__ out,
                                                                                 float* orders, float* __restrict__ sums,
                                                                                 float* __restrict__ ratio, float* __restrict__ scale,
                                                                                 int e_self, int e_sup, int cap_at_one,
                                                                                 float* __restrict__ store_out, float* __restrict__ wh_out,
                                                                                 float* __restrict__ reward) {
    __shared__ float part[kChunk][nic::kQuad][kLanes];
    __shared__ float rq[nic::kQuad][kLanes];
    __shared__ float cw[NIC_MAX_WAREHOUSES][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < io.dims.n_scenarios;
    const int Wn = io.dims.n_warehouses;
    if (q == 0 && live)
        nic::gnn_alloc_fwd_one(out, io.wh_inv, orders, sums, ratio, scale, io.dims.n_stores, e_self, e_sup, cap_at_one, b, io.dims.ldb);
    __syncthreads();   // the orders are read below by all four lanes of a scenario's quad
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
    if (q == 0 && live) {
        float total = nic::combine4(rq[0][x], rq[1][x], rq[2][x], rq[3][x]);
        float r_wh = 0.f;
        for (int w = 0; w < Wn; ++w) r_wh += cw[w][x];
        reward[b] = total + r_wh;
    }
}

template <int MAXW>
__global__ __launch_bounds__(kLanes * nic::kQuad) void gnn_alloc_env_bwd_kernel(
    NicEnvStepIO io, const float* __restrict__ out, const float* __restrict__ sums, const float* __restrict__ ratio,
    const float* __restrict__ scale, int n_edges, int e_self, int e_sup, int cap_at_one, const float* __restrict__ g_store_out,
    const float* __restrict__ g_wh_out, NicTable2 g_reward, float* __restrict__ g_store_in, float* g_wh_in, float* g_orders,
    float* __restrict__ d_out) {
    __shared__ float part[kChunk][nic::kQuad][kLanes];
    __shared__ float gwa[NIC_MAX_WAREHOUSES][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < io.dims.n_scenarios;
    const int S = io.dims.n_stores, Wn = io.dims.n_warehouses;
    const int64_t ldb = io.dims.ldb;
    const float gr = live ? g_reward.p[b * g_reward.scn_stride] : 0.f;
    float* g_store_orders = g_orders;                       // [S][1][ldb]
    float* g_wh_orders = g_orders + (int64_t)S * ldb;       // [1][ldb]
    for (int wc = 0; wc < Wn; wc += kChunk) {
        for (int i = 0; i < kChunk && wc + i < Wn; ++i) part[i][q][x] = live ? nic::env_ship_partial(io, wc + i, b, q) : 0.f;
        __syncthreads();
        for (int i = q; i < kChunk && wc + i < Wn; i += nic::kQuad) {
            const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
            gwa[wc + i][x] = live ? nic::env_bwd_warehouse<MAXW>(io, g_wh_out, gr, 0.f, wc + i, shipped, g_wh_in, g_wh_orders, b) : 0.f;
        }
        __syncthreads();
    }
    if (live)
        nic::env_bwd_stores<MAXW>(io, g_store_out, gr, [&](int w) { return gwa[w][x]; }, g_store_in, g_store_orders, b, q);
    __syncthreads();   // the order gradients (and the warehouse's on-hand gradient) are read below by the q = 0 lane of the quad
    if (q == 0 && live)
        nic::gnn_alloc_bwd_one(out, io.wh_inv, g_orders, sums, ratio, scale, d_out, g_wh_in, S, n_edges, e_self, e_sup, cap_at_one, b, ldb);
}

int check(const NicEnvStepIO* io, const char* who) {
    NIC_REQUIRE(io != nullptr, "%s: io is null", who);
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(d.n_scenarios > 0 && d.ldb >= d.n_scenarios && d.n_stores > 0, "%s: bad sizes", who);
    NIC_REQUIRE(d.n_warehouses == 1 && d.n_echelons == 0, "%s: one supplying warehouse, no extra echelons (else: two launches)", who);
    NIC_REQUIRE(d.store_slots >= 2 && d.store_slots <= NIC_MAX_SLOTS && d.warehouse_slots >= 2 && d.warehouse_slots <= NIC_MAX_SLOTS,
                "%s: pipeline lengths outside [2,%d]", who, NIC_MAX_SLOTS);
    NIC_REQUIRE(io->store_inv && io->wh_inv && io->demand.p && io->store_orders.p && io->wh_orders.p && io->underage.p && io->holding.p &&
                    io->lead_times.p && io->wh_holding.p && io->wh_lead_times.p,
                "%s: null buffer", who);
    return 0;
}
int max_slots(const NicEnvDims& d) { return d.store_slots > d.warehouse_slots ? d.store_slots : d.warehouse_slots; }
}  // namespace

extern "C" {

int nic_gnn_alloc_env_fwd(const NicEnvStepIO* io, const float* out, float* orders, float* sums, float* ratio, float* scale,
                          int32_t e_self, int32_t e_supplier, int32_t cap_at_one, float* store_inv_out, float* wh_inv_out, float* reward,
                          void* stream) {
    if (int e = check(io, "nic_gnn_alloc_env_fwd")) return e;
    NIC_REQUIRE(out && orders && sums && ratio && scale && store_inv_out && wh_inv_out && reward && e_supplier >= 0,
                "nic_gnn_alloc_env_fwd: null buffer / bad edge");
    NIC_REQUIRE(io->store_orders.p == orders && io->wh_orders.p == orders + (int64_t)io->dims.n_stores * io->dims.ldb,
                "nic_gnn_alloc_env_fwd: io's order tables must be the rows of `orders` ([S + 1][ldb])");
    const dim3 grid(nic::ceil_div(io->dims.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(io->dims);
    nic::note_kernelf("gnn_alloc_env_fwd_kernel<%d>", m <= 4 ? 4 : (m <= 8 ? 8 : NIC_MAX_SLOTS));
#define NIC_GAE_FWD(MW)                                                                                                           \
    hipLaunchKernelGGL(gnn_alloc_env_fwd_kernel<MW>, grid, block, 0, s, *io, out, orders, sums, ratio, scale, e_self, e_supplier,  \
                       cap_at_one, store_inv_out, wh_inv_out, reward)
    if (m <= 4) NIC_GAE_FWD(4);
    else if (m <= 8) NIC_GAE_FWD(8);
    else NIC_GAE_FWD(NIC_MAX_SLOTS);
#undef NIC_GAE_FWD
    return nic::check_launch("nic_gnn_alloc_env_fwd");
}

int nic_gnn_alloc_env_bwd(const NicEnvStepIO* io, const float* out, const float* sums, const float* ratio, const float* scale,
                          int32_t n_edges, int32_t e_self, int32_t e_supplier, int32_t cap_at_one, const float* g_store_out,
                          const float* g_wh_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_orders, float* d_out,
                          void* stream) {
    if (int e = check(io, "nic_gnn_alloc_env_bwd")) return e;
    NIC_REQUIRE(out && sums && ratio && scale && g_reward.p && g_store_in && g_wh_in && g_orders && d_out && e_supplier >= 0 &&
                    n_edges > io->dims.n_stores,
                "nic_gnn_alloc_env_bwd: null buffer / bad edge count");
    const dim3 grid(nic::ceil_div(io->dims.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(io->dims);
    nic::note_kernelf("gnn_alloc_env_bwd_kernel<%d>", m <= 4 ? 4 : (m <= 8 ? 8 : NIC_MAX_SLOTS));
#define NIC_GAE_BWD(MW)                                                                                                          \
    hipLaunchKernelGGL(gnn_alloc_env_bwd_kernel<MW>, grid, block, 0, s, *io, out, sums, ratio, scale, n_edges, e_self, e_supplier, \
                       cap_at_one, g_store_out, g_wh_out, g_reward, g_store_in, g_wh_in, g_orders, d_out)
    if (m <= 4) NIC_GAE_BWD(4);
    else if (m <= 8) NIC_GAE_BWD(8);
    else NIC_GAE_BWD(NIC_MAX_SLOTS);
#undef NIC_GAE_BWD
    return nic::check_launch("nic_gnn_alloc_env_bwd");
}
}
