// One period of inventory dynamics on gfx950: forward and analytic backward.
//
// Kernel shape: one lane per scenario, 64-lane workgroups (one wavefront each) so that B scenarios spread over
// B/64 workgroups and the chip's 256 CUs all get work even at B = 32k; every global access of a wave is a
// contiguous 256-byte row segment of a scenario-minor buffer.  HBM-bound: algorithmic bytes per scenario-period are
// 4*[2*(S*Ws + Wn*Ww + E*We) + S + (S*max(Wn,1) + Wn + E) + 1] (state read + write, demand, orders, reward).
// The arithmetic lives in env_step_body.h (shared with the host-side test build).
#include "env_step_body.h"
#include "nic_common.h"

namespace {

constexpr int kBlock = 64;

template <int MAXW>
__global__ __launch_bounds__(kBlock) void env_step_fwd_kernel(NicEnvStepIO io, float* __restrict__ store_out,
                                                              float* __restrict__ wh_out, float* __restrict__ ech_out,
                                                              float* __restrict__ reward) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= io.dims.n_scenarios) return;
    nic::env_step_fwd_scenario<MAXW>(io, store_out, wh_out, ech_out, reward, b);
}

template <int MAXW>
__global__ __launch_bounds__(kBlock) void env_step_bwd_kernel(NicEnvStepIO io, const float* __restrict__ g_store_out,
                                                              const float* __restrict__ g_wh_out,
                                                              const float* __restrict__ g_ech_out, NicTable2 g_reward,
                                                              float* __restrict__ g_store_in, float* __restrict__ g_wh_in,
                                                              float* __restrict__ g_ech_in, float* g_store_orders,
                                                              float* __restrict__ g_wh_orders,
                                                              float* __restrict__ g_ech_orders) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= io.dims.n_scenarios) return;
    nic::env_step_bwd_scenario<MAXW>(io, g_store_out, g_wh_out, g_ech_out, g_reward, g_store_in, g_wh_in, g_ech_in,
                                     g_store_orders, g_wh_orders, g_ech_orders, b);
}

int validate(const NicEnvStepIO* io, const char* who) {
    NIC_REQUIRE(io != nullptr, "%s: io is null", who);
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(d.n_scenarios > 0 && d.ldb >= d.n_scenarios, "%s: bad n_scenarios/ldb (%d/%d)", who, d.n_scenarios, d.ldb);
    NIC_REQUIRE(d.n_stores > 0, "%s: n_stores must be positive", who);
    NIC_REQUIRE(d.n_warehouses >= 0 && d.n_echelons >= 0, "%s: negative location count", who);
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
                     void* stream) {
    if (int e = validate(io, "nic_env_step_fwd")) return e;
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(store_inv_out && reward, "nic_env_step_fwd: null output");
    NIC_REQUIRE(d.n_warehouses == 0 || wh_inv_out, "nic_env_step_fwd: null warehouse output");
    NIC_REQUIRE(d.n_echelons == 0 || ech_inv_out, "nic_env_step_fwd: null echelon output");
    const dim3 grid(nic::ceil_div(d.n_scenarios, kBlock)), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(d);
    if (m <= 4)
        hipLaunchKernelGGL(env_step_fwd_kernel<4>, grid, block, 0, s, *io, store_inv_out, wh_inv_out, ech_inv_out, reward);
    else if (m <= 8)
        hipLaunchKernelGGL(env_step_fwd_kernel<8>, grid, block, 0, s, *io, store_inv_out, wh_inv_out, ech_inv_out, reward);
    else
        hipLaunchKernelGGL(env_step_fwd_kernel<NIC_MAX_SLOTS>, grid, block, 0, s, *io, store_inv_out, wh_inv_out,
                           ech_inv_out, reward);
    return nic::check_launch("nic_env_step_fwd");
}

int nic_env_step_bwd(const NicEnvStepIO* io, const float* g_store_out, const float* g_wh_out, const float* g_ech_out,
                     NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_ech_in, float* g_store_orders,
                     float* g_wh_orders, float* g_ech_orders, void* stream) {
    if (int e = validate(io, "nic_env_step_bwd")) return e;
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(g_reward.p, "nic_env_step_bwd: null g_reward");
    NIC_REQUIRE(g_store_in && g_store_orders, "nic_env_step_bwd: null store gradient output");
    NIC_REQUIRE(d.n_warehouses == 0 || (g_wh_in && g_wh_orders), "nic_env_step_bwd: null warehouse gradient output");
    NIC_REQUIRE(d.n_echelons == 0 || (g_ech_in && g_ech_orders), "nic_env_step_bwd: null echelon gradient output");
    const dim3 grid(nic::ceil_div(d.n_scenarios, kBlock)), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
    const int m = max_slots(d);
#define NIC_LAUNCH_BWD(MW)                                                                                              \
    hipLaunchKernelGGL(env_step_bwd_kernel<MW>, grid, block, 0, s, *io, g_store_out, g_wh_out, g_ech_out, g_reward,      \
                       g_store_in, g_wh_in, g_ech_in, g_store_orders, g_wh_orders, g_ech_orders)
    if (m <= 4) NIC_LAUNCH_BWD(4);
    else if (m <= 8) NIC_LAUNCH_BWD(8);
    else NIC_LAUNCH_BWD(NIC_MAX_SLOTS);
#undef NIC_LAUNCH_BWD
    return nic::check_launch("nic_env_step_bwd");
}
}
