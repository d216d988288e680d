// Whole-horizon rollout + forward-mode gradient of the closed-form policies (body: closed_form_body.h).  One lane = one
// store chain for all T periods; 64-lane workgroups (one wavefront) so that 32k scenarios already spread over every CU.
// HBM traffic: 4 B of demand per chain-period, nothing else in the period loop unless reward_hist is requested.
// Latency-bound below ~10^6 chains (one dependent ~60-instruction chain per period per lane); the next period's demand is
// always in flight.  The echelon chain (4 levels) is bound by vector issue: ~265 vector instructions per wave-period, tangents as
// packed pairs (closed_form_body.h).
#include "nic_common.h"
#include "closed_form_body.h"

namespace {
constexpr int kBlock = 64;    // one wavefront per workgroup (32k chains already spread over every CU; 256-thread workgroups measured
                              // the same at 10^6 chains: the CU's workgroup slots do not cap the resident wavefronts)
constexpr int kWave = 64;

template <int NP, int MF, bool CHAIN, int WC = 0>
__global__ __launch_bounds__(kBlock) void closed_form_kernel(NicClosedFormDesc d, const float* __restrict__ levels,
                                                              const float* __restrict__ demand,
                                                              const float* __restrict__ state0, float* __restrict__ reward_hist,
                                                              float* __restrict__ totals, float* __restrict__ state_final,
                                                              float* __restrict__ partial, int partial_stride, int sums_at) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int s = blockIdx.y;
    d.levels = levels;
    d.demand = demand;
    d.state0 = state0;
    float g[NP > 0 ? NP : 1];
    float sums[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NP > 0 ? NP : 1); ++j) g[j] = 0.f;
    if (b < d.n_scenarios) nic::closed_form_chain<NP, MF, CHAIN, WC>(d, reward_hist, totals, state_final, s, b, g, sums);
    if (partial) {   // one row per wavefront of chains: [d total / d level_j ...][total, reported from column sums_at on, if >= 0]
        const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
        const int64_t n_waves = (d.n_scenarios + kWave - 1) / kWave;
        float* row = partial + ((int64_t)blockIdx.y * n_waves + wave_id) * partial_stride;
        const bool writer = (threadIdx.x & (kWave - 1)) == 0 && wave_id < n_waves;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float v = g[j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (writer) row[j] = v;
        }
        if (sums_at >= 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = sums[j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
                if (writer) row[sums_at + j] = v;
            }
        }
    }
}

int validate(const NicClosedFormDesc* d) {
    NIC_REQUIRE(d, "nic_closed_form_rollout: null descriptor");
    NIC_REQUIRE(d->levels && d->demand && d->state0, "nic_closed_form_rollout: null levels / demand / state0");
    NIC_REQUIRE(d->n_scenarios > 0 && d->ldb >= d->n_scenarios && d->T > 0 && d->t0 >= 0 && d->S >= 1,
                "nic_closed_form_rollout: bad sizes");
    NIC_REQUIRE(d->policy >= NIC_CF_BASE_STOCK && d->policy <= NIC_CF_ECHELON, "nic_closed_form_rollout: unknown policy %d", d->policy);
    const int F = d->Ws + d->Wn * d->Ww + d->E * d->We;
    NIC_REQUIRE(d->Ws >= 2 && d->Ws <= NIC_MAX_SLOTS && d->Ww <= NIC_MAX_SLOTS && d->We <= NIC_MAX_SLOTS,
                "nic_closed_form_rollout: pipelines need 2..%d slots", NIC_MAX_SLOTS);
    (void)F;
    if (d->policy == NIC_CF_ECHELON) {
        NIC_REQUIRE(d->S == 1 && d->Wn == 1 && d->E >= 1 && d->E <= 3 && d->Ww >= 2 && d->We >= 2,
                    "nic_closed_form_rollout: echelon_stock needs the serial system (S = 1, Wn = 1, 1 <= E <= 3)");
        NIC_REQUIRE(d->n_levels == d->E + 2, "nic_closed_form_rollout: echelon_stock takes E + 2 levels");
    } else {
        NIC_REQUIRE(d->Wn == 0 && d->E == 0, "nic_closed_form_rollout: base-stock policies run on settings without warehouses");
        NIC_REQUIRE(d->n_levels == (d->policy == NIC_CF_CAPPED ? 2 : 1), "nic_closed_form_rollout: wrong number of levels");
    }
    NIC_REQUIRE(d->underage.p && d->holding.p && d->lead.p, "nic_closed_form_rollout: null store table");
    return 0;
}
}  // namespace

extern "C" {

int nic_closed_form_num_partials(int32_t n_scenarios, int32_t S) {
    if (n_scenarios <= 0 || S <= 0) return 0;
    return nic::ceil_div(n_scenarios, kWave) * S;
}

int nic_closed_form_rollout_sums(const NicClosedFormDesc* d, float* reward_hist, float* totals, float* state_final, float* partial,
                                 int32_t partial_stride, int32_t with_grad, int32_t with_sums, void* stream) {
    float* g_levels_partial = with_grad ? partial : nullptr;
    if (int e = validate(d)) return e;
    NIC_REQUIRE(!(d->round_orders && g_levels_partial), "nic_closed_form_rollout: rounded orders have no gradient");
    NIC_REQUIRE(!(with_grad || with_sums) || (partial && partial_stride >= (with_grad ? d->n_levels : 0) + (with_sums ? 2 : 0)),
                "nic_closed_form_rollout: partial buffer missing or its rows too short");
    const int sums_at = with_sums ? (with_grad ? d->n_levels : 0) : -1;
    float* part = (with_grad || with_sums) ? partial : nullptr;
    const dim3 grid(nic::ceil_div(d->n_scenarios, kBlock), d->S), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
    const int np = g_levels_partial ? d->n_levels : 0;
    const bool chain = d->policy == NIC_CF_ECHELON;
    int w = d->Ws;  // register slots per pipeline: the longest live pipeline, rounded up to 4 / 8 / 16
    if (chain) w = w > d->Ww ? w : d->Ww, w = w > d->We ? w : d->We;
    const int mf = w <= 4 ? 4 : (w <= 8 ? 8 : 16);
    // single-store chains with pipelines of 2..4 slots whose lead time does not vary over the scenarios (table stride 0: every
    // shipped setting): the variant with the pipeline length compiled in and the order placed by a wave-uniform branch
    const int wc = (!chain && mf == 4 && d->lead.scn_stride == 0) ? d->Ws : 0;
    if (wc) nic::note_kernelf("closed_form_kernel<%d,%d,false,%d>", np, mf, wc);
    else nic::note_kernelf("closed_form_kernel<%d,%d,%s>", np, mf, chain ? "true" : "false");
#define NIC_CF_LAUNCH(NP, MF, CH)                                                                                          \
    hipLaunchKernelGGL((closed_form_kernel<NP, MF, CH>), grid, block, 0, s, *d, d->levels, d->demand, d->state0, reward_hist, \
                       totals, state_final, part, partial_stride, sums_at)
    if (chain) {  // echelon_stock on the serial system: E + 2 = 3..5 levels
#define NIC_CF_CHAIN(MF)                                        \
    do {                                                        \
        if (np == 0) NIC_CF_LAUNCH(0, MF, true);                \
        else if (np == 3) NIC_CF_LAUNCH(3, MF, true);           \
        else if (np == 4) NIC_CF_LAUNCH(4, MF, true);           \
        else NIC_CF_LAUNCH(5, MF, true);                        \
    } while (0)
        if (mf == 4) NIC_CF_CHAIN(4);
        else if (mf == 8) NIC_CF_CHAIN(8);
        else NIC_CF_CHAIN(16);
#undef NIC_CF_CHAIN
    } else {      // base_stock (1 level) / capped_base_stock (2 levels) on independent stores
#define NIC_CF_STORE(MF)                                        \
    do {                                                        \
        if (np == 0) NIC_CF_LAUNCH(0, MF, false);               \
        else if (np == 1) NIC_CF_LAUNCH(1, MF, false);          \
        else NIC_CF_LAUNCH(2, MF, false);                       \
    } while (0)
#define NIC_CF_LAUNCH_C(NP, WCV)                                                                                               \
    hipLaunchKernelGGL((closed_form_kernel<NP, 4, false, WCV>), grid, block, 0, s, *d, d->levels, d->demand, d->state0, reward_hist, \
                       totals, state_final, part, partial_stride, sums_at)
#define NIC_CF_STORE_C(WCV)                                     \
    do {                                                        \
        if (np == 0) NIC_CF_LAUNCH_C(0, WCV);                   \
        else if (np == 1) NIC_CF_LAUNCH_C(1, WCV);              \
        else NIC_CF_LAUNCH_C(2, WCV);                           \
    } while (0)
        if (wc == 2) NIC_CF_STORE_C(2);
        else if (wc == 3) NIC_CF_STORE_C(3);
        else if (wc == 4) NIC_CF_STORE_C(4);
        else if (mf == 4) NIC_CF_STORE(4);
        else if (mf == 8) NIC_CF_STORE(8);
        else NIC_CF_STORE(16);
#undef NIC_CF_STORE
#undef NIC_CF_STORE_C
#undef NIC_CF_LAUNCH_C
    }
#undef NIC_CF_LAUNCH
    return nic::check_launch("nic_closed_form_rollout");
}
}
