// Policy heads (logits -> feasible orders) on gfx950; one lane per scenario, arithmetic in policy_heads_body.h.
// HBM-bound elementwise kernels over scenario-minor buffers (coalesced 256-B rows per wave).
#include "nic_common.h"
#include "policy_heads_body.h"

namespace {
constexpr int kBlock = 64;

// grid.y = min(Wn, 4): the warehouses of a scenario are independent, so up to four lanes share one scenario
__global__ __launch_bounds__(kBlock) void head_warehouse_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh_inv,
                                                                    const int32_t* __restrict__ adj, float ub, int trans,
                                                                    float* __restrict__ so, float* __restrict__ wo, int S,
                                                                    int Wn, int Ww, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    for (int w = blockIdx.y; w < Wn; w += gridDim.y)
        nic::head_warehouse_fwd_one(Z, wh_inv, adj, ub, trans, so, wo, S, Wn, Ww, ldb, b, w);
}
__global__ __launch_bounds__(kBlock) void head_warehouse_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh_inv,
                                                                    const int32_t* __restrict__ adj, float ub, int trans,
                                                                    const float* __restrict__ gso, const float* __restrict__ gwo,
                                                                    float* __restrict__ dZ, float* g_wh_inv, int S, int Wn,
                                                                    int Ww, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    for (int w = blockIdx.y; w < Wn; w += gridDim.y)
        nic::head_warehouse_bwd_one(Z, wh_inv, adj, ub, trans, gso, gwo, dZ, g_wh_inv, S, Wn, Ww, ldb, b, w);
}
__global__ void head_softplus_fwd_kernel(const float* __restrict__ Z, float* __restrict__ o, int rows, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B)
        for (int r = 0; r < rows; ++r) o[r * ldb + b] = nic::softplus1_fwd(Z[r * ldb + b]);
}
__global__ void head_softplus_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ g, float* __restrict__ dZ,
                                         int rows, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B)
        for (int r = 0; r < rows; ++r) dZ[r * ldb + b] = g[r * ldb + b] * nic::softplus1_grad(Z[r * ldb + b]);
}
__global__ __launch_bounds__(kBlock) void head_serial_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh_inv,
                                                                 const float* __restrict__ ech_inv, float ub,
                                                                 float* __restrict__ so, float* __restrict__ wo,
                                                                 float* __restrict__ eo, int E, int Ww, int We, int B,
                                                                 int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b < B) nic::head_serial_fwd_scenario(Z, wh_inv, ech_inv, ub, so, wo, eo, E, Ww, We, ldb, b);
}
__global__ __launch_bounds__(kBlock) void head_serial_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh_inv,
                                                                 const float* __restrict__ ech_inv, float ub,
                                                                 const float* __restrict__ gso, const float* __restrict__ gwo,
                                                                 const float* __restrict__ geo, float* __restrict__ dZ,
                                                                 float* g_wh_inv, float* g_ech_inv, int E, int Ww, int We,
                                                                 int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b < B) nic::head_serial_bwd_scenario(Z, wh_inv, ech_inv, ub, gso, gwo, geo, dZ, g_wh_inv, g_ech_inv, E, Ww, We, ldb, b);
}
__global__ void round_orders_kernel(float* __restrict__ x, int rows, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B)
        for (int r = blockIdx.y; r < rows; r += gridDim.y) x[r * ldb + b] = rintf(x[r * ldb + b]);
}
}  // namespace

extern "C" {

int nic_round_orders(float* x, int32_t rows, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(x && rows > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_round_orders: bad arguments");
    nic::note_kernel("round_orders_kernel");
    hipLaunchKernelGGL(round_orders_kernel, dim3(nic::ceil_div(n_scenarios, 256), rows < 64 ? rows : 64), dim3(256), 0,
                       nic::as_stream(stream), x, rows, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_round_orders");
}

int nic_head_warehouse_fwd(const float* Z, const float* wh_inv, const int32_t* adjacency, float upper_bound,
                           int32_t transshipment, float* store_orders, float* wh_orders, int32_t S, int32_t Wn, int32_t Ww,
                           int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && wh_inv && adjacency && store_orders && wh_orders, "nic_head_warehouse_fwd: null buffer");
    NIC_REQUIRE(S > 0 && Wn > 0 && Ww > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_warehouse_fwd: bad sizes");
    nic::note_kernel("head_warehouse_fwd_kernel");
    hipLaunchKernelGGL(head_warehouse_fwd_kernel, dim3(nic::ceil_div(n_scenarios, kBlock), Wn < 4 ? Wn : 4), dim3(kBlock), 0,
                       nic::as_stream(stream), Z, wh_inv, adjacency, upper_bound, transshipment, store_orders, wh_orders, S,
                       Wn, Ww, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_warehouse_fwd");
}

int nic_head_warehouse_bwd(const float* Z, const float* wh_inv, const int32_t* adjacency, float upper_bound,
                           int32_t transshipment, const float* g_store_orders, const float* g_wh_orders, float* dZ,
                           float* g_wh_inv, int32_t S, int32_t Wn, int32_t Ww, int32_t n_scenarios, int32_t ldb,
                           void* stream) {
    NIC_REQUIRE(Z && wh_inv && adjacency && g_store_orders && g_wh_orders && dZ && g_wh_inv,
                "nic_head_warehouse_bwd: null buffer");
    NIC_REQUIRE(S > 0 && Wn > 0 && Ww > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_warehouse_bwd: bad sizes");
    nic::note_kernel("head_warehouse_bwd_kernel");
    hipLaunchKernelGGL(head_warehouse_bwd_kernel, dim3(nic::ceil_div(n_scenarios, kBlock), Wn < 4 ? Wn : 4), dim3(kBlock), 0,
                       nic::as_stream(stream), Z, wh_inv, adjacency, upper_bound, transshipment, g_store_orders, g_wh_orders,
                       dZ, g_wh_inv, S, Wn, Ww, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_warehouse_bwd");
}

int nic_head_softplus_fwd(const float* Z, float* orders, int32_t rows, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && orders && rows > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_softplus_fwd: bad arguments");
    nic::note_kernel("head_softplus_fwd_kernel");
    hipLaunchKernelGGL(head_softplus_fwd_kernel, dim3(nic::ceil_div(n_scenarios, 256)), dim3(256), 0, nic::as_stream(stream),
                       Z, orders, rows, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_softplus_fwd");
}

int nic_head_softplus_bwd(const float* Z, const float* g_orders, float* dZ, int32_t rows, int32_t n_scenarios, int32_t ldb,
                          void* stream) {
    NIC_REQUIRE(Z && g_orders && dZ && rows > 0 && n_scenarios > 0 && ldb >= n_scenarios,
                "nic_head_softplus_bwd: bad arguments");
    nic::note_kernel("head_softplus_bwd_kernel");
    hipLaunchKernelGGL(head_softplus_bwd_kernel, dim3(nic::ceil_div(n_scenarios, 256)), dim3(256), 0, nic::as_stream(stream),
                       Z, g_orders, dZ, rows, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_softplus_bwd");
}

int nic_head_serial_fwd(const float* Z, const float* wh_inv, const float* ech_inv, float upper_bound, float* store_orders,
                        float* wh_orders, float* ech_orders, int32_t E, int32_t Ww, int32_t We, int32_t n_scenarios,
                        int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && wh_inv && store_orders && wh_orders && (E == 0 || (ech_inv && ech_orders)),
                "nic_head_serial_fwd: null buffer");
    NIC_REQUIRE(E >= 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_serial_fwd: bad sizes");
    nic::note_kernel("head_serial_fwd_kernel");
    hipLaunchKernelGGL(head_serial_fwd_kernel, dim3(nic::ceil_div(n_scenarios, kBlock)), dim3(kBlock), 0,
                       nic::as_stream(stream), Z, wh_inv, ech_inv, upper_bound, store_orders, wh_orders, ech_orders, E, Ww, We,
                       n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_serial_fwd");
}

int nic_head_serial_bwd(const float* Z, const float* wh_inv, const float* ech_inv, float upper_bound,
                        const float* g_store_orders, const float* g_wh_orders, const float* g_ech_orders, float* dZ,
                        float* g_wh_inv, float* g_ech_inv, int32_t E, int32_t Ww, int32_t We, int32_t n_scenarios,
                        int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && wh_inv && g_store_orders && g_wh_orders && dZ && g_wh_inv && (E == 0 || (ech_inv && g_ech_orders && g_ech_inv)),
                "nic_head_serial_bwd: null buffer");
    NIC_REQUIRE(E >= 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_serial_bwd: bad sizes");
    nic::note_kernel("head_serial_bwd_kernel");
    hipLaunchKernelGGL(head_serial_bwd_kernel, dim3(nic::ceil_div(n_scenarios, kBlock)), dim3(kBlock), 0,
                       nic::as_stream(stream), Z, wh_inv, ech_inv, upper_bound, g_store_orders, g_wh_orders, g_ech_orders, dZ,
                       g_wh_inv, g_ech_inv, E, Ww, We, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_serial_bwd");
}
}
