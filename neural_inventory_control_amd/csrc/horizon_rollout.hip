// Whole-horizon rollout of the data_driven policy (DataDrivenNet, neural_networks.py:430-515) for the batches the reference
// trains it on: 72 .. 288 products per batch x 21 stores x 3 warehouses x 95 weeks (many_warehouses_real_data_lost_demand.yml).
//
// Such a batch cannot fill the chip period by period: every one of the ~11 launches of a period runs for 10-30 us on a handful
// of workgroups, and a training step is ~1,050 dependent launches (14 ms eager, 9.7 ms replayed from a HIP graph).  Here ONE
// forward launch and ONE backward launch walk all T periods; a workgroup owns 16 scenarios (the N of v_mfma_f32_16x16x4_f32) for
// the whole horizon and nothing but the histories the backward / the weight-gradient GEMMs need ever leaves the CU:
//   * what the policy's first layer does with the OBSERVATION rows of its input (past-demand window, costs, days from
//     christmas, lead times: 442 of 597 rows at the reference's sizes) does not depend on the rollout - the caller contracts it
//     for all periods at once with one ordinary GEMM over T x ldb columns (z1_obs, bias included);  the serial chain only
//     contracts the STATE rows (S Ws + Wn Ww, 135) per period;
//   * weights live in registers as MFMA A fragments for the whole horizon (wave w owns rows 16 w .. 16 w + 15 of every layer);
//     activations cross the four wavefronts through 4 KB of LDS per layer in a b128-friendly order;
//   * state, logits, orders, demand of the period and every static table (costs, lead times) sit in LDS as [row][16 scenarios];
//     head and env step give every (scenario, store) pair its own lane (256 lanes for 16 x 21 pairs): the one-location bodies
//     of env_step_body.h (`env_*_t`: the arithmetic of a store / warehouse of the per-period kernels, bit for bit, read through an
//     accessor over 32-bit LDS offsets), the cross-store sums in the per-period kernels' Sum4 order, the data_driven head's
//     per-warehouse sums in store order between two barriers.  (The first version ran them on ONE wavefront, four lanes per
//     scenario: the env step alone was 6 us of a 13.7 us period.)
//   * tape modes (NicHorizonDesc.head_mode): the same env phases driven by ORDERS (1: just-in-time) or order-up-to LEVELS (2: the
//     quantile policies; the backward then returns d loss / d level) that the caller computed for all periods in one batched pass;
//   * the backward kernel walks the periods in reverse over the stored histories (state rows, hidden activations, logits,
//     orders), next period's loads issued a whole period ahead, and leaves the three pre-activation gradients per period; the
//     weight gradients are then ordinary contractions over (period x scenario) columns (nic_linear_wgrad).
//   * global memory: `vmcnt` retires loads and stores in issue order, so all global traffic of a period is issued in one burst at
//     the top of the period (next period's loads first, then the previous period's results from registers / LDS) and nothing is
//     waited for until the top of the next period; barriers publish LDS only (nic::lds_barrier).
// Latency-bound by construction (5 of 256 CUs for a 72-product batch): the figure of merit is microseconds per period of the
// dependent chain (9.5 forward / 12.4 backward at the reference's real-data shape), not a roofline fraction.  History layout: element (row, t, b) at row * hist_stride + t * ldb + b.
// Built with -ffp-contract=off (head / env arithmetic rounds like the reference's separate aten ops); the MFMA chains are fma
// by construction.  Same arithmetic as the per-period route except the summation order inside a layer's contraction.
#include "env_step_body.h"
#include "nic_common.h"
#include "policy_heads_body.h"

#ifndef HZ_FORCE_OCC2
#define HZ_FORCE_OCC2 0   // (experiment switch: every variant held to 256 registers = two workgroups per CU)
#endif

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int NB = 16;          // scenarios per workgroup
constexpr int kThreads = 256;   // four wavefronts
constexpr int kMaxH = 64;       // widest hidden layer
constexpr int kMaxOut = 128;    // most logits rows
constexpr int kMaxStores = 64;
constexpr int kMaxFD = 256;     // most state rows

// In-kernel timestamps (tuning build only: tools/horizon_stamp_probe.py compiles its own copy of the library with
// -DNIC_TUNING_BUILD; the product library contains none of this): the 100 MHz wall clock of workgroup 0's four wavefronts at up to
// 16 points of every period, [t][wave][point].
#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_hz_stamps = nullptr;
#define HZ_STAMP(t, point)                                                                                       \
    do {                                                                                                         \
        if (g_hz_stamps != nullptr && blockIdx.x == 0 && I.lane == 0) {                                           \
            g_hz_stamps[((t) * 4 + I.wave) * 16 + (point)] = wall_clock64();                                      \
            if ((point) == 0) g_hz_stamps[((t) * 4 + I.wave) * 16 + 15] = __builtin_readcyclecounter();           \
        }                                                                                                        \
    } while (0)
#else
#define HZ_STAMP(t, point) do { } while (0)
#endif

__device__ __forceinline__ float elu_f(float x) {   // (csrc/linear_mfma.hip::elu_f: the GEMM epilogues' ELU)
    const float xn = fminf(x, 0.f);
    const float series =
        xn * fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float viaexp = __expf(xn) - 1.f;
    const float neg = xn > -0.35f ? series : viaexp;
    return x > 0.f ? x : neg;
}
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

__host__ __device__ inline int up(int n, int m) { return (n + m - 1) / m * m; }

// ---- template variants, chosen from the shapes (host and device agree through these two functions) ----
// forward: MFMA steps of the first layer's state-row contraction (4 rows each); backward: 16-row tiles of state rows per wavefront
// and MFMA steps over the logits rows.  The MFMA loops are BRANCH-FREE over these compile-time counts (zero fragments beyond the
// real sizes) so that the compiler can issue every LDS read of a layer ahead of its MFMAs.
__host__ __device__ inline int fwd_steps(int FD) { return FD <= 64 ? 16 : (FD <= 160 ? 40 : 64); }
__host__ __device__ inline int bwd_variant(int FD, int n_out) { return (FD <= 64 && n_out <= 16) ? 0 : ((FD <= 192 && n_out <= 80) ? 1 : 2); }
__host__ __device__ inline int bwd_tiles(int v) { return v == 0 ? 1 : (v == 1 ? 3 : 4); }
__host__ __device__ inline int bwd_out_steps(int v) { return v == 0 ? 4 : (v == 1 ? 20 : 32); }

// LDS carve-up (float offsets), identical on host (size) and device.  Every per-scenario array is [row][16 scenarios].
struct HzLayout {
    int st0, st1;            // forward: state before / after the period [rows_st][16]; backward: st0 = state before the period
    int hxa, hxb;            // layer-to-layer exchange, 64 rows each, element (row, j) at ((row >> 2) * 16 + j) * 4 + (row & 3)
    int z, ord;              // logits [rows_z][16]; orders [n_ord][16] + what each warehouse ships [Wn][16]
    int tab_u, tab_h, tab_l, tab_wh, tab_wl, tab_we;   // static tables
    int dem, mask;           // demand of the period [S][16]; adjacency mask [S][Wn] (no scenario index)
    int scl, cst, rq, cw;    // allocation scale [Wn][16], per-store cost [S][16], Sum4 partials [4][16], warehouse cost [Wn][16]
    int g0, g1, gord, dz3, araw, prod, cmn;   // backward: state gradients x 2, order gradients, logits gradient, head adjoint pieces
    int total;
};
__host__ __device__ inline HzLayout hz_layout(const NicEnvDims& d, int n_out, bool bwd) {
    const int S = d.n_stores, Wn = d.n_warehouses, nsup = Wn > 0 ? Wn : 1;
    const int FD = S * d.store_slots + Wn * d.warehouse_slots, FDp = up(FD, 16), n_ord = S * nsup + Wn;
    const int rows_st = bwd ? FDp : (FDp > 4 * fwd_steps(FD) ? FDp : 4 * fwd_steps(FD));
    const int rows_z = bwd ? 4 * bwd_out_steps(bwd_variant(FD, n_out)) : up(n_out, 16);
    HzLayout L;
    int o = 0;
    auto take = [&](int floats) { const int at = o; o += up(floats, 4); return at; };
    L.st0 = take(rows_st * NB);
    L.st1 = bwd ? L.st0 : take(rows_st * NB);
    L.hxa = take(kMaxH * NB);
    L.hxb = take(kMaxH * NB);
    L.z = take(rows_z * NB);
    L.ord = take((n_ord + Wn) * NB);
    L.tab_u = take(S * NB);
    L.tab_h = take(S * NB);
    L.tab_l = take(S * nsup * NB);
    L.tab_wh = take(Wn * NB);
    L.tab_wl = take(Wn * NB);
    L.tab_we = take(Wn * NB);
    L.dem = take(S * NB);
    L.mask = take(S * Wn);
    L.scl = take(Wn * NB);
    L.cst = take(S * NB);
    L.rq = take(nic::kQuad * NB);
    L.cw = take(Wn * NB);
    L.g0 = L.g1 = L.gord = L.dz3 = L.araw = L.prod = L.cmn = 0;
    if (bwd) {
        L.g0 = take(FDp * NB);
        L.g1 = take(FDp * NB);
        L.gord = take(n_ord * NB);
        L.dz3 = take(rows_z * NB);
        L.araw = take(S * nsup * NB);
        L.prod = take(S * nsup * NB);
        L.cmn = take(Wn * NB);
    }
    L.total = o;
    return L;
}

struct HzIds {
    int tid, wave, lane, j, g, r0;
};
__device__ __forceinline__ HzIds hz_ids() {
    HzIds I;
    I.tid = threadIdx.x;
    I.wave = I.tid >> 6;
    I.lane = I.tid & 63;
    I.j = I.lane & 15;    // scenario of the block (MFMA column n; every per-scenario array uses the same column)
    I.g = I.lane >> 4;    // lane group (MFMA k / output-row group)
    I.r0 = I.tid >> 4;    // 0..15: this thread's first row in [row][16] arrays (rows r0, r0 + 16, ...) = its first store / task
    return I;
}

// One scenario's view of the workgroup's LDS blocks for the accessor-generic env bodies (env_step_body.h): 32-bit float offsets with
// the scenario column folded in, row stride NB known at compile time.
struct LdsEnv {
    float* l;                                                        // the workgroup's LDS
    int o_st, o_nx, o_ord, o_dem, o_tu, o_th, o_tl, o_twh, o_twl, o_twe;   // state before / after the period, orders, demand, static tables (+ column)
    int o_gout, o_gin, o_gord;                                        // backward: incoming / outgoing state gradients, order gradients
    int S_, Wn_, nsup_, Ws_, Ww_, whs, flags;                         // whs = first warehouse row of a state block; flags: 1 lost, 2 profit, 4 edge
    __device__ __forceinline__ int S() const { return S_; }
    __device__ __forceinline__ int Wn() const { return Wn_; }
    __device__ __forceinline__ int nsup() const { return nsup_; }
    __device__ __forceinline__ int Ws() const { return Ws_; }
    __device__ __forceinline__ int Ww() const { return Ww_; }
    __device__ __forceinline__ bool lost() const { return flags & 1; }
    __device__ __forceinline__ bool profit() const { return flags & 2; }
    __device__ __forceinline__ bool has_edge() const { return flags & 4; }
    __device__ __forceinline__ float inv(int s, int k) const { return l[o_st + (s * Ws_ + k) * NB]; }
    __device__ __forceinline__ float dem(int s) const { return l[o_dem + s * NB]; }
    __device__ __forceinline__ float under(int s) const { return l[o_tu + s * NB]; }
    __device__ __forceinline__ float hold(int s) const { return l[o_th + s * NB]; }
    __device__ __forceinline__ float ord(int s, int w) const { return l[o_ord + (s * nsup_ + w) * NB]; }
    __device__ __forceinline__ float lead(int s, int w) const { return l[o_tl + (s * nsup_ + w) * NB]; }
    __device__ __forceinline__ void put_inv(int s, int k, float v) const { l[o_nx + (s * Ws_ + k) * NB] = v; }
    __device__ __forceinline__ float wh_inv(int w, int k) const { return l[o_st + (whs + w * Ww_ + k) * NB]; }
    __device__ __forceinline__ float wh_hold(int w) const { return l[o_twh + w * NB]; }
    __device__ __forceinline__ float wh_lead(int w) const { return l[o_twl + w * NB]; }
    __device__ __forceinline__ float wh_edge(int w) const { return l[o_twe + w * NB]; }
    __device__ __forceinline__ float wh_ord(int w) const { return l[o_ord + (S_ * nsup_ + w) * NB]; }
    __device__ __forceinline__ void put_wh(int w, int k, float v) const { l[o_nx + (whs + w * Ww_ + k) * NB] = v; }
    __device__ __forceinline__ float g_out(int s, int k) const { return l[o_gout + (s * Ws_ + k) * NB]; }
    __device__ __forceinline__ float gwh_out(int w, int k) const { return l[o_gout + (whs + w * Ww_ + k) * NB]; }
    __device__ __forceinline__ void put_g_in(int s, int k, float v) const { l[o_gin + (s * Ws_ + k) * NB] = v; }
    __device__ __forceinline__ void put_gwh_in(int w, int k, float v) const { l[o_gin + (whs + w * Ww_ + k) * NB] = v; }
    __device__ __forceinline__ void put_g_ord(int s, int w, float v) const { l[o_gord + (s * nsup_ + w) * NB] = v; }
    __device__ __forceinline__ void put_gwh_ord(int w, float v) const { l[o_gord + (S_ * nsup_ + w) * NB] = v; }
};

// the static tables of the block's 16 scenarios -> LDS ([row][16], scenario stride 1), and the accessor over them
__device__ __forceinline__ LdsEnv hz_stage_tables(const NicHorizonDesc& d, const HzLayout& L, float* lds, const HzIds& I, int64_t b) {
    const NicEnvDims& D = d.io.dims;
    const int S = D.n_stores, Wn = D.n_warehouses, nsup = Wn > 0 ? Wn : 1;
    auto fill2 = [&](int at, const NicTable2& tb, int n) {
        if (!tb.p) return;
        for (int r = I.r0; r < n; r += 16) lds[at + r * NB + I.j] = tb.p[r * tb.loc_stride + b * tb.scn_stride];
    };
    fill2(L.tab_u, d.io.underage, S);
    fill2(L.tab_h, d.io.holding, S);
    fill2(L.tab_wh, d.io.wh_holding, Wn);
    fill2(L.tab_wl, d.io.wh_lead_times, Wn);
    fill2(L.tab_we, d.io.wh_edge_costs, Wn);
    const NicTable3& lt = d.io.lead_times;
    for (int r = I.r0; r < S * nsup; r += 16)
        lds[L.tab_l + r * NB + I.j] = lt.p[(r / nsup) * lt.loc_stride + (r % nsup) * lt.sup_stride + b * lt.scn_stride];
    for (int r = I.tid; r < S * Wn; r += kThreads) lds[L.mask + r] = d.mask[r];
    LdsEnv a;
    a.l = lds;
    a.o_st = a.o_nx = L.st0 + I.j;
    a.o_ord = L.ord + I.j;
    a.o_dem = L.dem + I.j;
    a.o_tu = L.tab_u + I.j;
    a.o_th = L.tab_h + I.j;
    a.o_tl = L.tab_l + I.j;
    a.o_twh = L.tab_wh + I.j;
    a.o_twl = L.tab_wl + I.j;
    a.o_twe = L.tab_we + I.j;
    a.o_gout = a.o_gin = L.g0 + I.j;
    a.o_gord = L.gord + I.j;
    a.S_ = S;
    a.Wn_ = Wn;
    a.nsup_ = nsup;
    a.Ws_ = D.store_slots;
    a.Ww_ = D.warehouse_slots;
    a.whs = S * D.store_slots;
    a.flags = (D.lost_demand ? 1 : 0) | (D.maximize_profit ? 2 : 0) | ((Wn && d.io.wh_edge_costs.p) ? 4 : 0);
    return a;
}

// rows r = r0, r0 + 16, ... < n of a [row][16] LDS array or of a global history.  The trip count is UNIFORM (scalar branch) and the
// ragged last group gets a clamped row index plus a `valid` flag: loads stay unconditional (no exec-mask branch per access).
template <int MAXU, class F>
__device__ __forceinline__ void for_rows(int n, int r0, F&& f) {
    const int n_u = (n + 15) >> 4;
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
        if (u < n_u) {
            const int r = r0 + 16 * u;
            f(u, r < n ? r : n - 1, r < n);
        }
}

// second output tile of wave w (logits rows beyond 64)
__device__ __forceinline__ int out_tile(int wave, int u) { return u == 0 ? wave : 4 + (3 - wave); }

using nic::lds_barrier;   // (nic_common.h: s_waitcnt lgkmcnt(0); s_barrier - nothing the wavefronts exchange goes through global memory)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// one 64-deep contraction of resident A fragments with an exchange block (k = 16 jj + 4 g + i); two accumulator chains
__device__ __forceinline__ f32x4 layer64(const float (&aW)[16], const float* hx, const HzIds& I, f32x4 acc) {
    f32x4 x[4], acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) x[jj] = *reinterpret_cast<const f32x4*>(hx + ((4 * jj + I.g) * NB + I.j) * 4);
#pragma unroll
    for (int jj = 0; jj < 4; jj += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = mfma4(aW[4 * jj + i], x[jj][i], acc);
            acc1 = mfma4(aW[4 * jj + 4 + i], x[jj + 1][i], acc1);
        }
    }
    return acc + acc1;
}

// ------------------------------------------------------------------------------------------------------------------------
// Global memory discipline of both kernels: vmcnt counts loads and stores in issue order, so waiting for ANY prefetched value also
// drains every store issued before it.  All global traffic of a period is therefore issued in ONE burst at the top of the period
// - the loads for the NEXT period first, then the stores of the PREVIOUS period's results (kept in registers / LDS until then) - and
// nothing is waited for until the top of the next period, a whole period of the chain later.
// (the smallest variant - one-store settings, which the reference trains in batches of 8,192 = 512 workgroups - is held to 256
// registers so that two workgroups share a CU: one round instead of two)
template <int MAXW, int MAXS1>
__global__ __launch_bounds__(kThreads, (MAXS1 == 16 || HZ_FORCE_OCC2) ? 2 : 1) void horizon_fwd_kernel(NicHorizonDesc d, const float* __restrict__ z1_obs,
                                                                const float* __restrict__ state0, float* __restrict__ rewards,
                                                                float* __restrict__ state_final, float* __restrict__ state_hist,
                                                                float* __restrict__ h1_hist, float* __restrict__ h2_hist,
                                                                float* __restrict__ logits_hist, float* __restrict__ orders_hist) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const NicEnvDims& D = d.io.dims;
    const HzIds I = hz_ids();
    // (32-bit element offsets into every global buffer: the launcher checks that none reaches 2^31)
    const uint32_t b = blockIdx.x * NB + I.j, ld = D.ldb, hs = (uint32_t)d.hist_stride;
    const bool live = b < D.n_scenarios;
    const int S = D.n_stores, Wn = D.n_warehouses, nsup = Wn > 0 ? Wn : 1, Ww = D.warehouse_slots;
    const int FD = S * D.store_slots + Wn * Ww, n_ord = S * nsup + Wn, NOp = up(d.n_out, 16);
    const int H1 = d.head_mode == 0 ? d.H1 : 0, H2 = d.head_mode == 0 ? d.H2 : 0;   // (tape modes: no layers, every weight guard false)
    const HzLayout L = hz_layout(D, d.n_out, false);
    LdsEnv env = hz_stage_tables(d, L, lds, I, b);
    const int rows_st = (L.st1 - L.st0) / NB;

    // ---- resident weight fragments: A lane (g, j) holds W[tile row j][k(step, g)] ----
    float aW1[MAXS1], aW2[16], aW3[2][16], bias2[4], bias3[2][4];
    const int row_w = 16 * I.wave + I.j;
#pragma unroll
    for (int s = 0; s < MAXS1; ++s) {
        const int k = 4 * s + I.g;
        aW1[s] = (row_w < H1 && k < FD) ? d.W1[(int64_t)row_w * d.ldw1 + k] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = 16 * (e >> 2) + 4 * I.g + (e & 3);
        aW2[e] = (row_w < H2 && k < H1) ? d.W2[(int64_t)row_w * d.ldw2 + k] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 16 * out_tile(I.wave, u) + I.j;
            aW3[u][e] = (row < d.n_out && k < H2) ? d.W3[(int64_t)row * d.ldw3 + k] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * I.wave + 4 * I.g + i;
        bias2[i] = (r < H2 && d.head_mode == 0) ? d.b2[r] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ro = 16 * out_tile(I.wave, u) + 4 * I.g + i;
            bias3[u][i] = (ro < d.n_out && d.head_mode == 0) ? d.b3[ro] : 0.f;
        }
    }

    // ---- period 0: state; the first z1_obs rows and demand into the prefetch registers ----
    for (int r = I.r0; r < rows_st; r += 16) {
        lds[L.st0 + r * NB + I.j] = (r < FD && live) ? state0[(uint32_t)r * ld + b] : 0.f;
        lds[L.st1 + r * NB + I.j] = 0.f;
    }
    for (int r = I.r0; r < NOp; r += 16) lds[L.z + r * NB + I.j] = 0.f;
    // head_mode 0: the policy MLP + data_driven head.  1: the orders of every period come from a tape (policies that do not read the
    // state: just-in-time).  2: order-up-to levels from a tape (the quantile policies: order = clip(level - pipeline total, 0))
    const int mode = d.head_mode, n_tape = mode == 1 ? n_ord : (mode == 2 ? S : 0);
    // (zero-initialised: a head mode's prefetch only fills the registers that mode reads - z1 with the MLP, tape_pf with a tape -
    // and the period loop copies all of them forward)
    float z1[4] = {0.f, 0.f, 0.f, 0.f}, dem_pf[kMaxStores / 16], tape_pf[kMaxOut / 16];
#pragma unroll
    for (int i = 0; i < kMaxOut / 16; ++i) tape_pf[i] = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxStores / 16; ++i) dem_pf[i] = 0.f;
    auto prefetch = [&](int t) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * I.wave + 4 * I.g + i;
                z1[i] = (r < H1 && live) ? z1_obs[(uint32_t)r * hs + (uint32_t)t * ld + b] : 0.f;
            }
        } else {
            for_rows<kMaxOut / 16>(n_tape, I.r0, [&](int u, int r, bool) { tape_pf[u] = d.tape[(uint32_t)r * hs + (uint32_t)t * ld + b]; });
        }
        for_rows<kMaxStores / 16>(S, I.r0, [&](int u, int r, bool) { dem_pf[u] = d.demand[((uint32_t)(d.t0 + t) * S + r) * ld + b]; });
    };
    prefetch(0);
    // results of the previous period, stored at the top of the next one
    f32x4 h1_p = {0.f, 0.f, 0.f, 0.f}, h2_p = h1_p, z_p[2] = {h1_p, h1_p};
    float reward_p = 0.f, ship_p[2] = {0.f, 0.f};
    auto flush = [&](int t) {   // histories of period t (the order block in LDS still holds period t's orders)
        const uint32_t at = (uint32_t)t * ld + b;
        if (!live) return;
        if (I.r0 == 0) rewards[at] = reward_p;
        if (!state_hist) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (mode != 0) break;
            const int r = 16 * I.wave + 4 * I.g + i;
            if (r < H1) h1_hist[(uint32_t)r * hs + at] = h1_p[i];
            if (r < H2) h2_hist[(uint32_t)r * hs + at] = h2_p[i];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ro = 16 * out_tile(I.wave, u) + 4 * I.g + i;
                if (ro < d.n_out) logits_hist[(uint32_t)ro * hs + at] = z_p[u][i];
            }
        }
        for_rows<kMaxOut / 16 + 2>(n_ord + Wn, I.r0, [&](int, int r, bool ok) {
            if (ok) orders_hist[(uint32_t)r * hs + at] = lds[L.ord + r * NB + I.j];
        });
    };
    lds_barrier();

    for (int t = 0; t < d.T; ++t) {
        float* cur = lds + ((t & 1) ? L.st1 : L.st0);
        float* nxt = lds + ((t & 1) ? L.st0 : L.st1);
        HZ_STAMP(t, 0);
        // ---- top of the period: consume the prefetch, then ONE burst of global traffic ----
        f32x4 acc0 = {z1[0], z1[1], z1[2], z1[3]}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for_rows<kMaxStores / 16>(S, I.r0, [&](int u, int r, bool ok) {
            if (ok) lds[L.dem + r * NB + I.j] = dem_pf[u];   // (read next in the env phase, four barriers down)
        });
        float tape_now[kMaxOut / 16];
#pragma unroll
        for (int u = 0; u < kMaxOut / 16; ++u) tape_now[u] = tape_pf[u];
        prefetch(t + 1 < d.T ? t + 1 : t);
        if (t > 0) flush(t - 1);
        if (state_hist && live)
            for_rows<kMaxFD / 16>(FD, I.r0, [&](int, int r, bool ok) {
                if (ok) state_hist[(uint32_t)r * hs + (uint32_t)t * ld + b] = cur[r * NB + I.j];
            });
        HZ_STAMP(t, 1);
        env.o_st = ((t & 1) ? L.st1 : L.st0) + I.j;
        env.o_nx = ((t & 1) ? L.st0 : L.st1) + I.j;
        const float* wh_now = cur + S * D.store_slots * NB;
        float* so = lds + L.ord;
        float* wo = lds + L.ord + S * nsup * NB;
        const float* Z = lds + L.z;
        if (mode != 0) {   // the period's tape rows: orders as they are (mode 1), or levels parked in the logits block (mode 2)
            for_rows<kMaxOut / 16>(n_tape, I.r0, [&](int u, int r, bool ok) {
                if (ok) lds[(mode == 1 ? L.ord : L.z) + r * NB + I.j] = (mode == 1 && d.round_orders) ? rintf(tape_now[u]) : tape_now[u];
            });
            lds_barrier();
        } else {
        // ---- layer 1: z1_obs (observation rows + bias, contracted outside) + W1[state rows] x state ----
        {
            float x[MAXS1];
#pragma unroll
            for (int s = 0; s < MAXS1; ++s) x[s] = cur[(4 * s + I.g) * NB + I.j];   // (rows beyond the state are zero)
#pragma unroll
            for (int s = 0; s < MAXS1; s += 2) {
                acc0 = mfma4(aW1[s], x[s], acc0);
                acc1 = mfma4(aW1[s + 1], x[s + 1], acc1);
            }
        }
        f32x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = elu_f(acc0[i] + acc1[i]);
        *reinterpret_cast<f32x4*>(lds + L.hxa + ((4 * I.wave + I.g) * NB + I.j) * 4) = h;
        h1_p = h;
        lds_barrier();
        HZ_STAMP(t, 2);
        // ---- layer 2 ----
        const f32x4 a2 = layer64(aW2, lds + L.hxa, I, f32x4{bias2[0], bias2[1], bias2[2], bias2[3]});
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = elu_f(a2[i]);
        *reinterpret_cast<f32x4*>(lds + L.hxb + ((4 * I.wave + I.g) * NB + I.j) * 4) = h;
        h2_p = h;
        lds_barrier();
        HZ_STAMP(t, 3);
        // ---- layer 3: logits (no activation here: the head applies the ReLU) ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = out_tile(I.wave, u);
            if (16 * tile < NOp) {
                const f32x4 z = layer64(aW3[u], lds + L.hxb, I, f32x4{bias3[u][0], bias3[u][1], bias3[u][2], bias3[u][3]});
                z_p[u] = z;
#pragma unroll
                for (int i = 0; i < 4; ++i) lds[L.z + (16 * tile + 4 * I.g + i) * NB + I.j] = z[i];
            }
        }
        lds_barrier();
        HZ_STAMP(t, 4);
        // ---- head (DataDrivenNet.forward :474-515): thread (store s = r0 + 16 u, scenario j) ----
        if (Wn > 0) {
            // masked ReLU outputs of this thread's stores; then per warehouse the allocation scale (serial sum in store order)
            for (int s = I.r0; s < S; s += 16)
                for (int w = 0; w < Wn; ++w)
                    so[(s * Wn + w) * NB + I.j] = fmaxf(Z[(Wn + s * Wn + w) * NB + I.j], 0.f) * lds[L.mask + s * Wn + w];
            lds_barrier();
            for (int w = I.r0; w < Wn; w += 16) {
                float avail = 0.f;
                for (int k = 0; k < Ww; ++k) avail += wh_now[(w * Ww + k) * NB + I.j];
                float sum = 0.f;
                for (int s0 = 0; s0 < S; s0 += nic::kHeadBatch) {
                    float a[nic::kHeadBatch];
#pragma unroll
                    for (int u = 0; u < nic::kHeadBatch; ++u) a[u] = so[((s0 + u < S ? s0 + u : S - 1) * Wn + w) * NB + I.j];
#pragma unroll
                    for (int u = 0; u < nic::kHeadBatch; ++u)
                        if (s0 + u < S) sum += a[u];
                }
                lds[L.scl + w * NB + I.j] = fminf(avail / (sum + 1e-10f), 1.f);
                const float own = fmaxf(Z[w * NB + I.j], 0.f);
                wo[w * NB + I.j] = d.round_orders ? rintf(own) : own;
            }
            lds_barrier();
        }
        }   // (head_mode 0)
        HZ_STAMP(t, 5);
        // ---- this thread's stores: final orders, then cost + pipeline update (env_fwd_one_store = a store of env_fwd_stores) ----
        for (int s = I.r0; s < S; s += 16) {
            if (mode == 2) {   // order up to the level (QuantilePolicy.forecast_base_stock_allocation, neural_networks.py:560-575)
                float pos = 0.f;
                for (int k = 0; k < D.store_slots; ++k) pos += cur[(s * D.store_slots + k) * NB + I.j];
                float a = Z[s * NB + I.j] - pos;
                if (!d.allow_negative) a = fmaxf(a, 0.f);
                so[s * NB + I.j] = d.round_orders ? rintf(a) : a;
            } else if (mode == 1) {
            } else if (Wn > 0) {
                for (int w = 0; w < Wn; ++w) {
                    const float a = so[(s * Wn + w) * NB + I.j] * lds[L.scl + w * NB + I.j];
                    so[(s * Wn + w) * NB + I.j] = d.round_orders ? rintf(a) : a;   // (discrete allocation, trainer.py:201-202)
                }
            } else {
                const float a = fmaxf(Z[s * NB + I.j], 0.f);
                so[s * NB + I.j] = d.round_orders ? rintf(a) : a;
            }
            lds[L.cst + s * NB + I.j] = nic::env_fwd_store_t<MAXW>(env, s);
        }
        lds_barrier();
        HZ_STAMP(t, 6);
        // ---- warehouses (row groups 0 .. Wn-1, ...) and the Sum4 partials of the store costs (the LAST four row groups: another
        // wavefront than the first warehouses', so that neither walks through the other's branch) ----
        for (int w = I.r0; w < Wn; w += 16) {
            const float shipped = nic::env_shipped_t(env, w);
            lds[L.cw + w * NB + I.j] = nic::env_fwd_warehouse_t<MAXW>(env, w, shipped);
            lds[L.ord + (n_ord + w) * NB + I.j] = shipped;   // (history: the backward does not re-sum the orders)
        }
        if (I.r0 >= 16 - nic::kQuad) {
            const int q = I.r0 - (16 - nic::kQuad);
            float r = 0.f;
            for (int s = q; s < S; s += nic::kQuad) r += lds[L.cst + s * NB + I.j];
            lds[L.rq + q * NB + I.j] = r;
        }
        lds_barrier();
        HZ_STAMP(t, 7);
        if (I.r0 == 0) {
            const float* rq = lds + L.rq;
            float total = nic::combine4(rq[I.j], rq[NB + I.j], rq[2 * NB + I.j], rq[3 * NB + I.j]);
            if (Wn > 0) {
                float r_wh = 0.f;
                for (int w = 0; w < Wn; ++w) r_wh += lds[L.cw + w * NB + I.j];
                total += r_wh;
            }
            reward_p = total;
        }
        HZ_STAMP(t, 8);
    }
    flush(d.T - 1);
    if (state_final && live) {
        const float* fin = lds + ((d.T & 1) ? L.st1 : L.st0);
        for (int r = I.r0; r < FD; r += 16) state_final[(uint32_t)r * ld + b] = fin[r * NB + I.j];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward sweep.  Per period (last to first): env-step adjoint (one lane per (scenario, store); the warehouse lanes beside them),
// the data_driven head's adjoint (per-warehouse reductions in store order between two barriers), then the three input-gradient
// contractions on resident TRANSPOSED weight fragments - the first layer's only over the state rows (nothing else of the input
// carries a gradient back in time).
template <int MAXW, int VAR>
__global__ __launch_bounds__(kThreads, (VAR == 0 || HZ_FORCE_OCC2) ? 2 : 1) void horizon_bwd_kernel(NicHorizonDesc d, const float* __restrict__ state_hist,
                                                                const float* __restrict__ h1_hist, const float* __restrict__ h2_hist,
                                                                const float* __restrict__ logits_hist,
                                                                const float* __restrict__ orders_hist, NicTable2 g_reward,
                                                                float* __restrict__ dz1_hist, float* __restrict__ dz2_hist,
                                                                float* __restrict__ dz3_hist) {
    constexpr int MAXT1 = VAR == 0 ? 1 : (VAR == 1 ? 3 : 4), MAXS3 = VAR == 0 ? 4 : (VAR == 1 ? 20 : 32);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const NicEnvDims& D = d.io.dims;
    const HzIds I = hz_ids();
    // (32-bit element offsets into every global buffer: the launcher checks that none reaches 2^31)
    const uint32_t b = blockIdx.x * NB + I.j, ld = D.ldb, hs = (uint32_t)d.hist_stride;
    const bool live = b < D.n_scenarios;
    const int S = D.n_stores, Wn = D.n_warehouses, nsup = Wn > 0 ? Wn : 1, Ww = D.warehouse_slots;
    const int FD = S * D.store_slots + Wn * Ww, FDp = up(FD, 16), n_ord = S * nsup + Wn;
    const int mode = d.head_mode;   // 0: policy MLP + data_driven head; 2: order-up-to levels from a tape (1 has no gradient)
    const int H1 = mode == 0 ? d.H1 : 0, H2 = mode == 0 ? d.H2 : 0, n_t1 = FDp / 16;
    const HzLayout L = hz_layout(D, d.n_out, true);
    LdsEnv env = hz_stage_tables(d, L, lds, I, b);
    const float* wh_now = lds + L.st0 + S * D.store_slots * NB;

    // ---- resident transposed fragments: A lane (g, j) holds W[k(step, g)][tile row j] ----
    float aW3t[MAXS3], aW2t[16], aW1t[MAXT1][16];
    const int row_w = 16 * I.wave + I.j;
#pragma unroll
    for (int s = 0; s < MAXS3; ++s) {
        const int k = 4 * s + I.g;
        aW3t[s] = (row_w < H2 && k < d.n_out) ? d.W3[(int64_t)k * d.ldw3 + row_w] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = 16 * (e >> 2) + 4 * I.g + (e & 3);
        aW2t[e] = (row_w < H1 && k < H2) ? d.W2[(int64_t)k * d.ldw2 + row_w] : 0.f;
#pragma unroll
        for (int u = 0; u < MAXT1; ++u) {
            const int row = 16 * (I.wave + 4 * u) + I.j;   // state row
            aW1t[u][e] = (row < FD && k < H1) ? d.W1[(int64_t)k * d.ldw1 + row] : 0.f;
        }
    }
    const float gr = live ? g_reward.p[b * g_reward.scn_stride] : 0.f;
    for (int r = I.r0; r < FDp; r += 16) lds[L.g0 + r * NB + I.j] = lds[L.g1 + r * NB + I.j] = lds[L.st0 + r * NB + I.j] = 0.f;
    for (int r = I.r0; r < 4 * MAXS3; r += 16) lds[L.dz3 + r * NB + I.j] = lds[L.z + r * NB + I.j] = 0.f;

    // what a period needs from the histories, fetched one period ahead into registers
    float p_st[kMaxFD / 16], p_z[kMaxOut / 16], p_ord[kMaxOut / 16 + 2], p_dem[kMaxStores / 16], p_h1[4], p_h2[4];
    auto fetch = [&](int t) {
        const uint32_t at = (uint32_t)t * ld + b;
        const float* zsrc = mode == 0 ? logits_hist : d.tape;   // (mode 2: the levels)
        for_rows<kMaxFD / 16>(FD, I.r0, [&](int u, int r, bool) { p_st[u] = state_hist[(uint32_t)r * hs + at]; });
        for_rows<kMaxOut / 16>(d.n_out, I.r0, [&](int u, int r, bool) { p_z[u] = zsrc[(uint32_t)r * hs + at]; });
        // orders (n_ord == n_out rows) + what the warehouses shipped (Wn <= 32 rows)
        for_rows<kMaxOut / 16 + 2>(n_ord + Wn, I.r0, [&](int u, int r, bool) { p_ord[u] = orders_hist[(uint32_t)r * hs + at]; });
        for_rows<kMaxStores / 16>(S, I.r0, [&](int u, int r, bool) { p_dem[u] = d.demand[((uint32_t)(d.t0 + t) * S + r) * ld + b]; });
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * I.wave + 4 * I.g + i;
            p_h1[i] = r < H1 ? h1_hist[(uint32_t)r * hs + at] : 0.f;
            p_h2[i] = r < H2 ? h2_hist[(uint32_t)r * hs + at] : 0.f;
        }
    };
    fetch(d.T - 1);
    f32x4 dz2_p = {0.f, 0.f, 0.f, 0.f}, dz1_p = dz2_p;
    auto flush = [&](int t) {   // pre-activation gradients of period t (the dz3 block in LDS still holds period t's)
        if (!live) return;
        const uint32_t at = (uint32_t)t * ld + b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * I.wave + 4 * I.g + i;
            if (r < H2) dz2_hist[(uint32_t)r * hs + at] = dz2_p[i];   // (tape modes: H1 = H2 = 0)
            if (r < H1) dz1_hist[(uint32_t)r * hs + at] = dz1_p[i];
        }
        for_rows<kMaxOut / 16>(d.n_out, I.r0, [&](int, int r, bool ok) {
            if (ok) dz3_hist[(uint32_t)r * hs + at] = lds[L.dz3 + r * NB + I.j];
        });
    };
    lds_barrier();

    for (int t = d.T - 1; t >= 0; --t) {
        float* g_next = lds + (((d.T - 1 - t) & 1) ? L.g1 : L.g0);   // gradient w.r.t. the state AFTER period t (zeros for t = T-1)
        float* g_cur = lds + (((d.T - 1 - t) & 1) ? L.g0 : L.g1);
        HZ_STAMP(t, 0);
        // ---- top of the period: this period's history out of the prefetch registers, then ONE burst of global traffic ----
        float h1v[4], h2v[4];
        for_rows<kMaxFD / 16>(FD, I.r0, [&](int u, int r, bool ok) { if (ok) lds[L.st0 + r * NB + I.j] = p_st[u]; });
        for_rows<kMaxOut / 16>(d.n_out, I.r0, [&](int u, int r, bool ok) { if (ok) lds[L.z + r * NB + I.j] = p_z[u]; });
        for_rows<kMaxOut / 16 + 2>(n_ord + Wn, I.r0, [&](int u, int r, bool ok) { if (ok) lds[L.ord + r * NB + I.j] = p_ord[u]; });
        for_rows<kMaxStores / 16>(S, I.r0, [&](int u, int r, bool ok) { if (ok) lds[L.dem + r * NB + I.j] = p_dem[u]; });
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h1v[i] = p_h1[i];
            h2v[i] = p_h2[i];
        }
        if (t > 0) fetch(t - 1);
        if (t < d.T - 1) flush(t + 1);
        lds_barrier();
        HZ_STAMP(t, 1);

        // ---- env-step adjoint: warehouses (threads r0 < Wn) beside the stores (every thread; each recomputes the gradient of its
        // suppliers' post-shipping on-hand instead of waiting for the warehouse lanes), then the head's per-store pieces ----
        float* gso = lds + L.gord;
        float* gwo = lds + L.gord + S * nsup * NB;
        const float* shipped = lds + L.ord + n_ord * NB;
        float* g_next_wh = g_next + S * D.store_slots * NB;
        float* g_cur_wh = g_cur + S * D.store_slots * NB;
        const float* Z = lds + L.z;
        env.o_gout = (((d.T - 1 - t) & 1) ? L.g1 : L.g0) + I.j;
        env.o_gin = (((d.T - 1 - t) & 1) ? L.g0 : L.g1) + I.j;
        for (int w = I.r0; w < Wn; w += 16) (void)nic::env_bwd_warehouse_t<MAXW>(env, gr, w, shipped[w * NB + I.j]);
        for (int s = I.r0; s < S; s += 16) {
            nic::env_bwd_store_t<MAXW>(env, gr, [&](int w) { return nic::env_bwd_wh_g_after_t(env, gr, w, shipped[w * NB + I.j]); }, s);
            if (mode == 2) {   // order = clip(level - pipeline total, 0): clamp(min = 0) passes the gradient where its input is >= 0
                float pos = 0.f;
                for (int k = 0; k < D.store_slots; ++k) pos += lds[L.st0 + (s * D.store_slots + k) * NB + I.j];
                const float a = Z[s * NB + I.j] - pos;
                const float gl = (d.allow_negative || a >= 0.f) ? gso[s * NB + I.j] : 0.f;
                lds[L.dz3 + s * NB + I.j] = gl;                                   // d loss / d level
                for (int k = 0; k < D.store_slots; ++k) g_cur[(s * D.store_slots + k) * NB + I.j] -= gl;
            } else if (Wn > 0) {
                for (int w = 0; w < Wn; ++w) {
                    const int r = s * Wn + w;
                    const float a = fmaxf(Z[(Wn + r) * NB + I.j], 0.f) * lds[L.mask + r];
                    lds[L.araw + r * NB + I.j] = a;
                    lds[L.prod + r * NB + I.j] = gso[r * NB + I.j] * a;
                }
            } else {
                lds[L.dz3 + s * NB + I.j] = Z[s * NB + I.j] > 0.f ? gso[s * NB + I.j] : 0.f;
            }
        }
        lds_barrier();
        HZ_STAMP(t, 2);
        if (mode != 0) continue;   // (no layers: the state gradient is complete, the level gradient waits in the dz3 block)
        if (Wn > 0) {
            // per warehouse: sums in store order (head_data_driven_bwd_one's), scale / common term for its stores' logits
            for (int w = I.r0; w < Wn; w += 16) {
                float avail = 0.f;
                for (int k = 0; k < Ww; ++k) avail += wh_now[(w * Ww + k) * NB + I.j];
                float sum = 0.f, dot = 0.f;
                for (int s0 = 0; s0 < S; s0 += nic::kHeadBatch) {
                    float a[nic::kHeadBatch], pr[nic::kHeadBatch];
#pragma unroll
                    for (int u = 0; u < nic::kHeadBatch; ++u) {
                        const int r = (s0 + u < S ? s0 + u : S - 1) * Wn + w;
                        a[u] = lds[L.araw + r * NB + I.j];
                        pr[u] = lds[L.prod + r * NB + I.j];
                    }
#pragma unroll
                    for (int u = 0; u < nic::kHeadBatch; ++u)
                        if (s0 + u < S) {
                            sum += a[u];
                            dot += pr[u];
                        }
                }
                const float den = sum + 1e-10f, ratio = avail / den;
                const float d_scale = ratio <= 1.f ? dot : 0.f;   // torch.clip(max = 1) passes the gradient where ratio <= 1
                lds[L.scl + w * NB + I.j] = fminf(ratio, 1.f);
                lds[L.cmn + w * NB + I.j] = -(d_scale * avail / (den * den));
                const float g_av = d_scale / den;
                for (int k = 0; k < Ww; ++k) g_cur_wh[(w * Ww + k) * NB + I.j] += g_av;
                lds[L.dz3 + w * NB + I.j] = Z[w * NB + I.j] > 0.f ? gwo[w * NB + I.j] : 0.f;
            }
            lds_barrier();
            for (int s = I.r0; s < S; s += 16)
                for (int w = 0; w < Wn; ++w) {
                    const int r = s * Wn + w;
                    const float da = gso[r * NB + I.j] * lds[L.scl + w * NB + I.j] + lds[L.cmn + w * NB + I.j];
                    lds[L.dz3 + (Wn + r) * NB + I.j] = Z[(Wn + r) * NB + I.j] > 0.f ? da * lds[L.mask + r] : 0.f;
                }
            lds_barrier();
        }
        HZ_STAMP(t, 3);
        // ---- dZ2 = (W3^T dZ3) * ELU'(h2) ----
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        {
            float x[MAXS3];
#pragma unroll
            for (int s = 0; s < MAXS3; ++s) x[s] = lds[L.dz3 + (4 * s + I.g) * NB + I.j];   // (rows beyond the logits are zero)
#pragma unroll
            for (int s = 0; s < MAXS3; s += 2) {
                a0 = mfma4(aW3t[s], x[s], a0);
                a1 = mfma4(aW3t[s + 1], x[s + 1], a1);
            }
        }
        f32x4 dz;
#pragma unroll
        for (int i = 0; i < 4; ++i) dz[i] = (a0[i] + a1[i]) * elu_grad_from_out(h2v[i]);
        *reinterpret_cast<f32x4*>(lds + L.hxa + ((4 * I.wave + I.g) * NB + I.j) * 4) = dz;
        dz2_p = dz;
        lds_barrier();
        HZ_STAMP(t, 4);
        // ---- dZ1 = (W2^T dZ2) * ELU'(h1) ----
        const f32x4 d1 = layer64(aW2t, lds + L.hxa, I, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int i = 0; i < 4; ++i) dz[i] = d1[i] * elu_grad_from_out(h1v[i]);
        *reinterpret_cast<f32x4*>(lds + L.hxb + ((4 * I.wave + I.g) * NB + I.j) * 4) = dz;
        dz1_p = dz;
        lds_barrier();
        HZ_STAMP(t, 5);
        // ---- state rows of the first layer's input gradient, added to what the env / head adjoints left in g_cur ----
#pragma unroll
        for (int u = 0; u < MAXT1; ++u) {
            const int tile = I.wave + 4 * u;
            if (tile < n_t1) {
                const f32x4 gx = layer64(aW1t[u], lds + L.hxb, I, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * tile + 4 * I.g + i;
                    if (r < FD) g_cur[r * NB + I.j] += gx[i];
                }
            }
        }
        HZ_STAMP(t, 6);
        lds_barrier();
    }
    flush(0);
}

int validate(const NicHorizonDesc* d, const char* who) {
    NIC_REQUIRE(d != nullptr, "%s: null descriptor", who);
    const NicEnvDims& D = d->io.dims;
    NIC_REQUIRE(D.n_scenarios > 0 && D.ldb >= D.n_scenarios && D.ldb % NB == 0, "%s: bad n_scenarios / ldb (%d / %d)", who,
                D.n_scenarios, D.ldb);
    NIC_REQUIRE(D.n_echelons == 0, "%s: settings with extra echelons take the per-period kernels", who);
    NIC_REQUIRE(D.n_stores > 0 && D.n_stores <= kMaxStores, "%s: 1..%d stores", who, kMaxStores);
    NIC_REQUIRE(D.n_warehouses >= 0 && D.n_warehouses <= NIC_MAX_WAREHOUSES, "%s: at most %d warehouses", who, NIC_MAX_WAREHOUSES);
    NIC_REQUIRE(D.store_slots >= 2 && D.store_slots <= 8 && (D.n_warehouses == 0 || (D.warehouse_slots >= 2 && D.warehouse_slots <= 8)),
                "%s: pipelines of 2..8 slots", who);
    const int FD = D.n_stores * D.store_slots + D.n_warehouses * D.warehouse_slots;
    NIC_REQUIRE(FD <= kMaxFD, "%s: %d state rows (at most %d)", who, FD, kMaxFD);
    const int n_out = D.n_warehouses ? D.n_warehouses + D.n_stores * D.n_warehouses : D.n_stores;
    NIC_REQUIRE(d->n_out == n_out && n_out <= kMaxOut, "%s: %d logits rows (the setting has %d; at most %d)", who, d->n_out, n_out,
                kMaxOut);
    NIC_REQUIRE(d->head_mode >= 0 && d->head_mode <= 2, "%s: head_mode 0 (MLP), 1 (order tape) or 2 (level tape)", who);
    if (d->head_mode != 0) {
        NIC_REQUIRE(d->tape != nullptr && d->demand != nullptr, "%s: null tape / demand", who);
        NIC_REQUIRE(d->head_mode == 1 || D.n_warehouses == 0, "%s: order-up-to levels are a one-supplier policy (no warehouses)", who);
        NIC_REQUIRE(d->T > 0 && d->t0 >= 0 && d->hist_stride >= (int64_t)d->T * D.ldb, "%s: bad T / t0 / hist_stride", who);
        NIC_REQUIRE(D.n_warehouses == 0 || d->mask, "%s: null adjacency mask", who);
        NIC_REQUIRE(d->io.underage.p && d->io.holding.p && d->io.lead_times.p, "%s: null store table", who);
        NIC_REQUIRE(D.n_warehouses == 0 || (d->io.wh_holding.p && d->io.wh_lead_times.p), "%s: null warehouse table", who);
        return 0;
    }
    NIC_REQUIRE(d->H1 > 0 && d->H1 <= kMaxH && d->H2 > 0 && d->H2 <= kMaxH, "%s: hidden widths 1..%d", who, kMaxH);
    NIC_REQUIRE(d->T > 0 && d->t0 >= 0 && d->hist_stride >= (int64_t)d->T * D.ldb, "%s: bad T / t0 / hist_stride", who);
    NIC_REQUIRE(d->W1 && d->W2 && d->W3 && d->b2 && d->b3 && d->demand, "%s: null weight / demand pointer", who);
    NIC_REQUIRE(d->ldw1 >= FD && d->ldw2 >= d->H1 && d->ldw3 >= d->H2, "%s: weight row strides too short", who);
    NIC_REQUIRE(D.n_warehouses == 0 || d->mask, "%s: null adjacency mask", who);
    NIC_REQUIRE(d->io.underage.p && d->io.holding.p && d->io.lead_times.p, "%s: null store table", who);
    NIC_REQUIRE(D.n_warehouses == 0 || (d->io.wh_holding.p && d->io.wh_lead_times.p), "%s: null warehouse table", who);
    return 0;
}

// the kernels index every global buffer with 32-bit element offsets
int check_offsets(const NicHorizonDesc* d, const char* who) {
    const NicEnvDims& D = d->io.dims;
    const int64_t FD = D.n_stores * D.store_slots + D.n_warehouses * D.warehouse_slots;
    int64_t rows = FD > d->n_out + D.n_warehouses ? FD : d->n_out + D.n_warehouses;
    if (d->head_mode == 0 && rows < kMaxH) rows = kMaxH;
    NIC_REQUIRE(rows * d->hist_stride + (int64_t)d->T * D.ldb < (1ll << 31), "%s: histories of %lld rows x stride %lld exceed 32-bit offsets",
                who, (long long)rows, (long long)d->hist_stride);
    NIC_REQUIRE((int64_t)(d->t0 + d->T + 1) * D.n_stores * D.ldb < (1ll << 31), "%s: demand trace exceeds 32-bit offsets", who);
    return 0;
}

int lds_bytes(const NicHorizonDesc* d, bool bwd) { return hz_layout(d->io.dims, d->n_out, bwd).total * (int)sizeof(float); }

template <typename K>
int allow_lds(K kernel, int bytes, const char* who) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return nic::fail("%s: %d bytes of LDS refused: %s", who, bytes, hipGetErrorString(e));
    }
    return 0;
}

}  // namespace

#ifdef NIC_TUNING_BUILD
extern "C" int nic_tuning_set_horizon_stamps(void* buf) {   // buf: device memory, [T][4][16] u64 (or null)
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_hz_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

int nic_horizon_rollout_ok(const NicHorizonDesc* d) {
    if (validate(d, "nic_horizon_rollout_ok") || check_offsets(d, "nic_horizon_rollout_ok")) return 0;
    return lds_bytes(d, true) <= 160 * 1024 ? 1 : 0;
}

int nic_horizon_rollout_fwd(const NicHorizonDesc* d, const float* z1_obs, const float* state0, float* rewards, float* state_final,
                            float* state_hist, float* h1_hist, float* h2_hist, float* logits_hist, float* orders_hist,
                            void* stream) {
    if (int e = validate(d, "nic_horizon_rollout_fwd")) return e;
    if (int e = check_offsets(d, "nic_horizon_rollout_fwd")) return e;
    NIC_REQUIRE((z1_obs || d->head_mode != 0) && state0 && rewards, "nic_horizon_rollout_fwd: null z1_obs / state0 / rewards");
    NIC_REQUIRE(!state_hist || (orders_hist && (d->head_mode != 0 || (h1_hist && h2_hist && logits_hist))),
                "nic_horizon_rollout_fwd: histories come together");
    const NicEnvDims& D = d->io.dims;
    const int FD = D.n_stores * D.store_slots + D.n_warehouses * D.warehouse_slots;
    const int bytes = lds_bytes(d, false);
    NIC_REQUIRE(bytes <= 160 * 1024, "nic_horizon_rollout_fwd: %d bytes of LDS", bytes);
    const dim3 grid(nic::ceil_div(D.n_scenarios, NB)), block(kThreads);
    hipStream_t s = nic::as_stream(stream);
    const int steps = fwd_steps(FD);
    nic::note_kernelf("horizon_fwd_kernel<8,%d>", steps);
#define NIC_HZ_FWD(S1)                                                                                                      \
    do {                                                                                                                    \
        if (int e = allow_lds(horizon_fwd_kernel<8, S1>, bytes, "nic_horizon_rollout_fwd")) return e;                        \
        hipLaunchKernelGGL((horizon_fwd_kernel<8, S1>), grid, block, bytes, s, *d, z1_obs, state0, rewards, state_final,     \
                           state_hist, h1_hist, h2_hist, logits_hist, orders_hist);                                         \
    } while (0)
    if (steps == 16) NIC_HZ_FWD(16);
    else if (steps == 40) NIC_HZ_FWD(40);
    else NIC_HZ_FWD(64);
#undef NIC_HZ_FWD
    return nic::check_launch("nic_horizon_rollout_fwd");
}

int nic_horizon_rollout_bwd(const NicHorizonDesc* d, const float* state_hist, const float* h1_hist, const float* h2_hist,
                            const float* logits_hist, const float* orders_hist, NicTable2 g_reward, float* dz1_hist,
                            float* dz2_hist, float* dz3_hist, void* stream) {
    if (int e = validate(d, "nic_horizon_rollout_bwd")) return e;
    if (int e = check_offsets(d, "nic_horizon_rollout_bwd")) return e;
    NIC_REQUIRE(d->head_mode != 1, "nic_horizon_rollout_bwd: an order tape has no gradient");
    NIC_REQUIRE(state_hist && orders_hist && g_reward.p && dz3_hist, "nic_horizon_rollout_bwd: null buffer");
    NIC_REQUIRE(d->head_mode != 0 || (h1_hist && h2_hist && logits_hist && dz1_hist && dz2_hist), "nic_horizon_rollout_bwd: null buffer");
    NIC_REQUIRE(!d->round_orders, "nic_horizon_rollout_bwd: rounded orders have no gradient (evaluation only)");
    const NicEnvDims& D = d->io.dims;
    const int FD = D.n_stores * D.store_slots + D.n_warehouses * D.warehouse_slots;
    const int bytes = lds_bytes(d, true);
    NIC_REQUIRE(bytes <= 160 * 1024, "nic_horizon_rollout_bwd: %d bytes of LDS", bytes);
    const dim3 grid(nic::ceil_div(D.n_scenarios, NB)), block(kThreads);
    hipStream_t s = nic::as_stream(stream);
    const int var = bwd_variant(FD, d->n_out);
    nic::note_kernelf("horizon_bwd_kernel<8,%d>", var);
#define NIC_HZ_BWD(V)                                                                                                       \
    do {                                                                                                                    \
        if (int e = allow_lds(horizon_bwd_kernel<8, V>, bytes, "nic_horizon_rollout_bwd")) return e;                         \
        hipLaunchKernelGGL((horizon_bwd_kernel<8, V>), grid, block, bytes, s, *d, state_hist, h1_hist, h2_hist, logits_hist, \
                           orders_hist, g_reward, dz1_hist, dz2_hist, dz3_hist);                                            \
    } while (0)
    if (var == 0) NIC_HZ_BWD(0);
    else if (var == 1) NIC_HZ_BWD(1);
    else NIC_HZ_BWD(2);
#undef NIC_HZ_BWD
    return nic::check_launch("nic_horizon_rollout_bwd");
}
}
