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
// Quad form (policy_heads_body.h): 64 scenarios x 4 lanes per workgroup, lane q = wavefront q keeps the logits of stores
// q, q+4, ... in registers; the three reductions of the softmax cross the quad through LDS.  S <= 4 * MAXSQ.
constexpr int kLanes = 64;
template <int MAXSQ>
__global__ __launch_bounds__(kLanes * nic::kQuad) void head_warehouse_fwd_quad_kernel(
    const float* __restrict__ Z, const float* __restrict__ wh_inv, const int32_t* __restrict__ adj, float ub, int trans,
    float* __restrict__ so, float* __restrict__ wo, int S, int Wn, int Ww, int B, int64_t ldb) {
    __shared__ float xm[nic::kQuad][kLanes], xd[nic::kQuad][kLanes];
    __shared__ int xn[nic::kQuad][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < B;
    const int64_t bb = live ? b : B - 1;   // dead lanes shadow the last scenario (loads only)
    for (int w = blockIdx.y; w < Wn; w += gridDim.y) {  // (warehouses are independent: grid.y spreads them over workgroups)
        if (w != (int)blockIdx.y) __syncthreads();      // the exchange arrays are reused
        nic::HeadLane<MAXSQ> L;
        int nc;
        xm[q][x] = nic::head_quad_load<MAXSQ, false>(L, Z, nullptr, adj, S, Wn, ldb, bb, w, q, nc);
        xn[q][x] = nc;
        const float stock = wh_inv[(int64_t)w * Ww * ldb + bb];
        __syncthreads();
        const float m = nic::head_quad_max(xm[0][x], xm[1][x], xm[2][x], xm[3][x], trans);
        const int n_conn = xn[0][x] + xn[1][x] + xn[2][x] + xn[3][x];
        xd[q][x] = nic::head_quad_exp<MAXSQ>(L, m);
        __syncthreads();
        const float denom = nic::head_quad_denom(xd[0][x], xd[1][x], xd[2][x], xd[3][x], m, trans);
        if (live) {
            nic::head_quad_fwd_store<MAXSQ>(L, denom, stock, n_conn, so, S, Wn, ldb, b, w, q);
            if (q == (w & 3)) nic::head_wh_order_fwd(Z, ub, wo, S, Wn, ldb, b, w);
        }
    }
}
template <int MAXSQ>
__global__ __launch_bounds__(kLanes * nic::kQuad) void head_warehouse_bwd_quad_kernel(
    const float* __restrict__ Z, const float* __restrict__ wh_inv, const int32_t* __restrict__ adj, float ub, int trans,
    const float* __restrict__ gso, const float* __restrict__ gwo, float* __restrict__ dZ, float* g_wh_inv, int S, int Wn, int Ww,
    int B, int64_t ldb) {
    __shared__ float xm[nic::kQuad][kLanes], xd[nic::kQuad][kLanes], xt[nic::kQuad][kLanes], xs[nic::kQuad][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const bool live = b < B;
    const int64_t bb = live ? b : B - 1;
    for (int w = blockIdx.y; w < Wn; w += gridDim.y) {
        if (w != (int)blockIdx.y) __syncthreads();
        nic::HeadLane<MAXSQ> L;
        int nc;
        xm[q][x] = nic::head_quad_load<MAXSQ, true>(L, Z, gso, adj, S, Wn, ldb, bb, w, q, nc);
        const float stock = wh_inv[(int64_t)w * Ww * ldb + bb];
        __syncthreads();
        const float m = nic::head_quad_max(xm[0][x], xm[1][x], xm[2][x], xm[3][x], trans);
        xd[q][x] = nic::head_quad_exp<MAXSQ>(L, m);
        __syncthreads();
        const float denom = nic::head_quad_denom(xd[0][x], xd[1][x], xd[2][x], xd[3][x], m, trans);
        float dq, sq;
        nic::head_quad_bwd_dots<MAXSQ>(L, denom, stock, dq, sq);
        xt[q][x] = dq;
        xs[q][x] = sq;
        __syncthreads();
        if (live) {
            const float dot = nic::combine4(xt[0][x], xt[1][x], xt[2][x], xt[3][x]);
            nic::head_quad_bwd_store<MAXSQ>(L, dot, stock, dZ, S, Wn, ldb, b, w, q);
            if (q == (w & 3)) {
                g_wh_inv[(int64_t)w * Ww * ldb + b] += nic::combine4(xs[0][x], xs[1][x], xs[2][x], xs[3][x]);
                nic::head_wh_order_bwd(Z, ub, gwo, dZ, S, Wn, ldb, b, w);
            }
        }
    }
}
// ---- data_driven head (neural_networks.py:474-515 + :111-138) -------------------------------------------------------------------
// Z rows = [Wn warehouse orders | S x Wn store orders (store-major)] BEFORE the output ReLU.  One lane = (scenario, warehouse):
// out = relu(Z); the warehouse's own order passes through; its store orders are masked by the adjacency, summed in store order
// and scaled by min(1, pipeline total of the warehouse / (sum + 1e-10)) - upstream passes the whole [Ww] pipeline as "available
// inventory" and `apply_proportional_allocation` sums it.  Wn == 0 (one-store settings): orders = relu(Z), one lane per scenario.
__global__ void head_data_driven_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh, const float* __restrict__ mask,
                                            float* __restrict__ so, float* __restrict__ wo, int S, int Wn, int Ww, int B,
                                            int64_t ldb) {
#pragma clang fp contract(off)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (Wn == 0) {
        for (int s = 0; s < S; ++s) so[(int64_t)s * ldb + b] = fmaxf(Z[(int64_t)s * ldb + b], 0.f);
        return;
    }
    nic::head_data_driven_fwd_one(Z, wh, mask, so, wo, S, Wn, Ww, ldb, b, blockIdx.y);
}

// adjoint: dZ (every row written) and g_wh[w][k][b] += d(scale) / (sum + eps) for every slot k of the pipeline
__global__ void head_data_driven_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ wh, const float* __restrict__ mask,
                                            const float* __restrict__ g_so, const float* __restrict__ g_wo, float* __restrict__ dZ,
                                            float* __restrict__ g_wh, int S, int Wn, int Ww, int B, int64_t ldb) {
#pragma clang fp contract(off)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (Wn == 0) {
        for (int s = 0; s < S; ++s) dZ[(int64_t)s * ldb + b] = Z[(int64_t)s * ldb + b] > 0.f ? g_so[(int64_t)s * ldb + b] : 0.f;
        return;
    }
    nic::head_data_driven_bwd_one(Z, wh, mask, g_so, g_wo, dZ, g_wh, S, Wn, Ww, ldb, b, blockIdx.y);
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
    if (S <= 64) {   // stores split over the four lanes of a quad, logits held in registers
        const dim3 grid(nic::ceil_div(n_scenarios, kLanes), Wn < 8 ? Wn : 8), block(kLanes * nic::kQuad);
        const int sq = S <= 16 ? 4 : (S <= 32 ? 8 : 16);
        nic::note_kernelf("head_warehouse_fwd_quad_kernel<%d>", sq);
#define NIC_LAUNCH_HEAD(SQ)                                                                                              \
    hipLaunchKernelGGL(head_warehouse_fwd_quad_kernel<SQ>, grid, block, 0, nic::as_stream(stream), Z, wh_inv, adjacency,   \
                       upper_bound, transshipment, store_orders, wh_orders, S, Wn, Ww, n_scenarios, (int64_t)ldb)
        if (sq == 4) NIC_LAUNCH_HEAD(4);
        else if (sq == 8) NIC_LAUNCH_HEAD(8);
        else NIC_LAUNCH_HEAD(16);
#undef NIC_LAUNCH_HEAD
        return nic::check_launch("nic_head_warehouse_fwd");
    }
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
    if (S <= 64) {
        const dim3 grid(nic::ceil_div(n_scenarios, kLanes), Wn < 8 ? Wn : 8), block(kLanes * nic::kQuad);
        const int sq = S <= 16 ? 4 : (S <= 32 ? 8 : 16);
        nic::note_kernelf("head_warehouse_bwd_quad_kernel<%d>", sq);
#define NIC_LAUNCH_HEAD(SQ)                                                                                              \
    hipLaunchKernelGGL(head_warehouse_bwd_quad_kernel<SQ>, grid, block, 0, nic::as_stream(stream), Z, wh_inv, adjacency,   \
                       upper_bound, transshipment, g_store_orders, g_wh_orders, dZ, g_wh_inv, S, Wn, Ww, n_scenarios,      \
                       (int64_t)ldb)
        if (sq == 4) NIC_LAUNCH_HEAD(4);
        else if (sq == 8) NIC_LAUNCH_HEAD(8);
        else NIC_LAUNCH_HEAD(16);
#undef NIC_LAUNCH_HEAD
        return nic::check_launch("nic_head_warehouse_bwd");
    }
    nic::note_kernel("head_warehouse_bwd_kernel");
    hipLaunchKernelGGL(head_warehouse_bwd_kernel, dim3(nic::ceil_div(n_scenarios, kBlock), Wn < 4 ? Wn : 4), dim3(kBlock), 0,
                       nic::as_stream(stream), Z, wh_inv, adjacency, upper_bound, transshipment, g_store_orders, g_wh_orders,
                       dZ, g_wh_inv, S, Wn, Ww, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_warehouse_bwd");
}

int nic_head_data_driven_fwd(const float* Z, const float* wh_inv, const float* mask, float* store_orders, float* wh_orders,
                             int32_t S, int32_t Wn, int32_t Ww, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && store_orders && (Wn == 0 || (wh_inv && mask && wh_orders)), "nic_head_data_driven_fwd: null buffer");
    NIC_REQUIRE(S > 0 && Wn >= 0 && (Wn == 0 || Ww > 0) && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_data_driven_fwd: bad sizes");
    nic::note_kernel("head_data_driven_fwd_kernel");
    hipLaunchKernelGGL(head_data_driven_fwd_kernel, dim3(nic::ceil_div(n_scenarios, 64), Wn > 0 ? Wn : 1), dim3(64), 0,
                       nic::as_stream(stream), Z, wh_inv, mask, store_orders, wh_orders, S, Wn, Ww, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_head_data_driven_fwd");
}

int nic_head_data_driven_bwd(const float* Z, const float* wh_inv, const float* mask, const float* g_store_orders,
                             const float* g_wh_orders, float* dZ, float* g_wh_inv, int32_t S, int32_t Wn, int32_t Ww,
                             int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(Z && g_store_orders && dZ && (Wn == 0 || (wh_inv && mask && g_wh_orders && g_wh_inv)),
                "nic_head_data_driven_bwd: null buffer");
    NIC_REQUIRE(S > 0 && Wn >= 0 && (Wn == 0 || Ww > 0) && n_scenarios > 0 && ldb >= n_scenarios, "nic_head_data_driven_bwd: bad sizes");
    nic::note_kernel("head_data_driven_bwd_kernel");
    hipLaunchKernelGGL(head_data_driven_bwd_kernel, dim3(nic::ceil_div(n_scenarios, 64), Wn > 0 ? Wn : 1), dim3(64), 0,
                       nic::as_stream(stream), Z, wh_inv, mask, g_store_orders, g_wh_orders, dZ, g_wh_inv, S, Wn, Ww, n_scenarios,
                       (int64_t)ldb);
    return nic::check_launch("nic_head_data_driven_bwd");
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
