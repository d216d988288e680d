// Feasibility head + one period of inventory dynamics in ONE launch per direction (round 4) - the vanilla_warehouse policy's
// `apply_softmax_feasibility_function` (neural_networks.py:140-166, 393-426) followed by Simulator.step (environment.py:110-299),
// and their adjoints in reverse order.
//
// Both halves already share a thread layout: a workgroup is 64 scenarios x 4 lanes (a "quad"), lane q owns stores q, q+4, ... and
// warehouse w is finished by lane w & 3.  Every order the head writes is therefore read back by THE SAME LANE in the env step
// (forward), and every order gradient the env adjoint writes is read back by the same lane in the head adjoint (backward): the
// fused kernels run the two bodies back to back with a workgroup barrier in between, the hand-off goes through the lane's own
// global stores (L1 / L2 hits; the rows are written once because the backward sweep and the tests read them).  What is saved
// per period and direction: a launch (~5-8 us of dispatch at the batch sizes the reference ships), the HBM re-read of the
// orders, and the second kernel's cold start (its first loads wait for nothing but the barrier).
// Arithmetic: the NIC_HD pieces of policy_heads_body.h / env_step_body.h, unchanged - results are bit-identical to the two
// separate launches (tests/test_gpu_kernels.py::test_head_env_fused_equals_the_two_launches).
// HBM-bound: algorithmic bytes per scenario-period forward = 4 [2 (S Ws + Wn Ww) + S + (S Wn + 2 Wn) + (S Wn + Wn) + 1]
// (state read + write, demand, logits, orders written, reward).
#include "env_step_body.h"
#include "nic_common.h"
#include "policy_heads_body.h"

namespace {

constexpr int kLanes = 64;
constexpr int kChunk = 8;  // warehouses whose shipment partials are exchanged per barrier round (as env_step.hip)

template <int MAXW, int MAXSQ>
__global__ __launch_bounds__(kLanes * nic::kQuad) void head_env_fwd_kernel(NicEnvStepIO io, const float* Z, const int32_t* __restrict__ adj,
                                                                            float ub, int trans, float* store_out, float* wh_out,
                                                                            float* __restrict__ reward, const int32_t* __restrict__ zrow,
                                                                            int first_wh_row) {
    __shared__ float xm[nic::kQuad][kLanes], xd[nic::kQuad][kLanes];
    __shared__ int xn[nic::kQuad][kLanes];
    __shared__ float part[kChunk][nic::kQuad][kLanes];
    __shared__ float rq[nic::kQuad][kLanes];
    __shared__ float cw[NIC_MAX_WAREHOUSES][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const int B = io.dims.n_scenarios, S = io.dims.n_stores, Wn = io.dims.n_warehouses, Ww = io.dims.warehouse_slots;
    const int64_t ldb = io.dims.ldb;
    const bool live = b < B;
    const int64_t bb = live ? b : B - 1;   // dead lanes shadow the last scenario (loads only)
    float* so = const_cast<float*>(io.store_orders.p);   // [S][Wn][ldb]: written here, read back below by the same lane
    float* wo = const_cast<float*>(io.wh_orders.p);      // [Wn][ldb]

    // ---- head: logits -> feasible orders (head_warehouse_fwd_quad_kernel's body, every warehouse in this workgroup) ----
    for (int w = 0; w < Wn; ++w) {
        if (w > 0) __syncthreads();   // the exchange arrays are reused
        nic::HeadLane<MAXSQ> L;
        int nc;
        xm[q][x] = nic::head_quad_load<MAXSQ, false>(L, Z, nullptr, adj, S, Wn, ldb, bb, w, q, nc, zrow);
        xn[q][x] = nc;
        const float stock = io.wh_inv[(int64_t)w * Ww * ldb + bb];
        __syncthreads();
        const float m = nic::head_quad_max(xm[0][x], xm[1][x], xm[2][x], xm[3][x], trans);
        const int n_conn = xn[0][x] + xn[1][x] + xn[2][x] + xn[3][x];
        xd[q][x] = nic::head_quad_exp<MAXSQ>(L, m);
        __syncthreads();
        const float denom = nic::head_quad_denom(xd[0][x], xd[1][x], xd[2][x], xd[3][x], m, trans);
        if (live) {
            nic::head_quad_fwd_store<MAXSQ>(L, denom, stock, n_conn, so, S, Wn, ldb, b, w, q);
            if (q == (w & 3)) nic::head_wh_order_fwd(Z, ub, wo, S, Wn, ldb, b, w, first_wh_row);
        }
    }
    __syncthreads();   // (every order row a lane reads below was written by that lane; the barrier orders the global accesses)

    // ---- env step (env_step_fwd_kernel's body) ----
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

template <int MAXW, int MAXSQ>
__global__ __launch_bounds__(kLanes * nic::kQuad) void head_env_bwd_kernel(
    NicEnvStepIO io, const float* Z, const int32_t* __restrict__ adj, float ub, int trans, const float* g_store_out,
    const float* g_wh_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_store_orders, float* g_wh_orders,
    float* dZ, const int32_t* __restrict__ zrow, int first_wh_row) {
    __shared__ float part[kChunk][nic::kQuad][kLanes];
    __shared__ float gwa[NIC_MAX_WAREHOUSES][kLanes];
    __shared__ float xm[nic::kQuad][kLanes], xd[nic::kQuad][kLanes], xt[nic::kQuad][kLanes], xs[nic::kQuad][kLanes];
    const int x = threadIdx.x & (kLanes - 1), q = threadIdx.x / kLanes;
    const int64_t b = (int64_t)blockIdx.x * kLanes + x;
    const int B = io.dims.n_scenarios, S = io.dims.n_stores, Wn = io.dims.n_warehouses, Ww = io.dims.warehouse_slots;
    const int64_t ldb = io.dims.ldb;
    const bool live = b < B;
    const int64_t bb = live ? b : B - 1;
    const float gr = live ? g_reward.p[b * g_reward.scn_stride] : 0.f;

    // ---- env adjoint (env_step_bwd_kernel's body, no echelons) ----
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
    __syncthreads();   // order gradients and the warehouse state gradient written above are read back below by the same lanes

    // ---- head adjoint (head_warehouse_bwd_quad_kernel's body) ----
    for (int w = 0; w < Wn; ++w) {
        if (w > 0) __syncthreads();
        nic::HeadLane<MAXSQ> L;
        int nc;
        xm[q][x] = nic::head_quad_load<MAXSQ, true>(L, Z, g_store_orders, adj, S, Wn, ldb, bb, w, q, nc, zrow);
        const float stock = io.wh_inv[(int64_t)w * Ww * ldb + bb];
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
            nic::head_quad_bwd_store<MAXSQ>(L, dot, stock, dZ, S, Wn, ldb, b, w, q, zrow);
            if (q == (w & 3)) {
                g_wh_in[(int64_t)w * Ww * ldb + b] += nic::combine4(xs[0][x], xs[1][x], xs[2][x], xs[3][x]);
                nic::head_wh_order_bwd(Z, ub, g_wh_orders, dZ, S, Wn, ldb, b, w, first_wh_row);
            }
        }
    }
}

int validate(const NicEnvStepIO* io, const float* Z, const int32_t* adj, const char* who) {
    NIC_REQUIRE(io != nullptr && Z != nullptr && adj != nullptr, "%s: null argument", who);
    const NicEnvDims& d = io->dims;
    NIC_REQUIRE(d.n_scenarios > 0 && d.ldb >= d.n_scenarios, "%s: bad n_scenarios/ldb (%d/%d)", who, d.n_scenarios, d.ldb);
    NIC_REQUIRE(d.n_stores > 0 && d.n_stores <= 64, "%s: 1..64 stores (the quad head keeps a lane's logits in registers)", who);
    NIC_REQUIRE(d.n_warehouses >= 1 && d.n_warehouses <= NIC_MAX_WAREHOUSES && d.n_echelons == 0,
                "%s: the vanilla_warehouse head needs 1..%d warehouses and no extra echelons", who, NIC_MAX_WAREHOUSES);
    NIC_REQUIRE(d.store_slots >= 2 && d.store_slots <= NIC_MAX_SLOTS && d.warehouse_slots >= 2 && d.warehouse_slots <= NIC_MAX_SLOTS,
                "%s: pipeline lengths outside [2,%d]", who, NIC_MAX_SLOTS);
    NIC_REQUIRE(io->store_inv && io->wh_inv && io->demand.p && io->store_orders.p && io->wh_orders.p && io->underage.p &&
                    io->holding.p && io->lead_times.p && io->wh_holding.p && io->wh_lead_times.p,
                "%s: null buffer in io", who);
    // the head writes the orders in its own [S][Wn][ldb] / [Wn][ldb] layout: the tables of `io` must describe exactly that
    NIC_REQUIRE(io->store_orders.scn_stride == 1 && io->store_orders.sup_stride == d.ldb &&
                    io->store_orders.loc_stride == (int64_t)d.n_warehouses * d.ldb && io->wh_orders.scn_stride == 1 &&
                    io->wh_orders.loc_stride == d.ldb,
                "%s: store_orders / wh_orders must be dense [S][Wn][ldb] / [Wn][ldb] blocks", who);
    return 0;
}

int max_slots(const NicEnvDims& d) { return d.store_slots > d.warehouse_slots ? d.store_slots : d.warehouse_slots; }

}  // namespace

extern "C" {

#define NIC_HEAD_ENV_DISPATCH(LAUNCH)            \
    do {                                         \
        const int m_ = max_slots(d);             \
        if (d.n_stores <= 16) {                  \
            if (m_ <= 4) LAUNCH(4, 4);           \
            else if (m_ <= 8) LAUNCH(8, 4);      \
            else LAUNCH(NIC_MAX_SLOTS, 4);       \
        } else {                                 \
            if (m_ <= 4) LAUNCH(4, 16);          \
            else if (m_ <= 8) LAUNCH(8, 16);     \
            else LAUNCH(NIC_MAX_SLOTS, 16);      \
        }                                        \
    } while (0)

int nic_head_env_fwd(const NicEnvStepIO* io, const float* Z, const int32_t* adjacency, const int32_t* logit_rows,
                     int32_t first_wh_row, float upper_bound, int32_t transshipment, float* store_inv_out, float* wh_inv_out,
                     float* reward, void* stream) {
    if (int e = validate(io, Z, adjacency, "nic_head_env_fwd")) return e;
    NIC_REQUIRE(store_inv_out && wh_inv_out && reward, "nic_head_env_fwd: null output");
    NIC_REQUIRE((logit_rows == nullptr) == (first_wh_row < 0), "nic_head_env_fwd: logit_rows and first_wh_row go together");
    const NicEnvDims& d = io->dims;
    const dim3 grid(nic::ceil_div(d.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernelf("head_env_fwd_kernel<%d,%d>", max_slots(d) <= 4 ? 4 : (max_slots(d) <= 8 ? 8 : NIC_MAX_SLOTS),
                      d.n_stores <= 16 ? 4 : 16);
#define NIC_L(MW, SQ)                                                                                                          \
    hipLaunchKernelGGL((head_env_fwd_kernel<MW, SQ>), grid, block, 0, s, *io, Z, adjacency, upper_bound, transshipment, store_inv_out, \
                       wh_inv_out, reward, logit_rows, first_wh_row)
    NIC_HEAD_ENV_DISPATCH(NIC_L);
#undef NIC_L
    return nic::check_launch("nic_head_env_fwd");
}

int nic_head_env_bwd(const NicEnvStepIO* io, const float* Z, const int32_t* adjacency, const int32_t* logit_rows,
                     int32_t first_wh_row, float upper_bound, int32_t transshipment, const float* g_store_out,
                     const float* g_wh_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_store_orders,
                     float* g_wh_orders, float* dZ, void* stream) {
    if (int e = validate(io, Z, adjacency, "nic_head_env_bwd")) return e;
    NIC_REQUIRE(g_reward.p && g_store_in && g_wh_in && g_store_orders && g_wh_orders && dZ, "nic_head_env_bwd: null buffer");
    NIC_REQUIRE((logit_rows == nullptr) == (first_wh_row < 0), "nic_head_env_bwd: logit_rows and first_wh_row go together");
    const NicEnvDims& d = io->dims;
    const dim3 grid(nic::ceil_div(d.n_scenarios, kLanes)), block(kLanes * nic::kQuad);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernelf("head_env_bwd_kernel<%d,%d>", max_slots(d) <= 4 ? 4 : (max_slots(d) <= 8 ? 8 : NIC_MAX_SLOTS),
                      d.n_stores <= 16 ? 4 : 16);
#define NIC_L(MW, SQ)                                                                                                          \
    hipLaunchKernelGGL((head_env_bwd_kernel<MW, SQ>), grid, block, 0, s, *io, Z, adjacency, upper_bound, transshipment, g_store_out, \
                       g_wh_out, g_reward, g_store_in, g_wh_in, g_store_orders, g_wh_orders, dZ, logit_rows, first_wh_row)
    NIC_HEAD_ENV_DISPATCH(NIC_L);
#undef NIC_L
    return nic::check_launch("nic_head_env_bwd");
}
}
