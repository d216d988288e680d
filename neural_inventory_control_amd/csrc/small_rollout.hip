// Whole-horizon rollout kernels for the small (one-store chain, 32-wide MLP) policies: ONE launch runs all T periods.
// One lane = one scenario (64-lane workgroups), state / activations / gradients in registers across the horizon (body:
// small_rollout_body.h).  The weight pointer is a separate `const __restrict__` kernel argument so that hipcc proves the
// (wave-uniform) weight reads invariant and issues them on the scalar path (s_load) instead of 64-lane vector loads.
// VALU-bound: ~2 * params FMAs per scenario-period forward; HBM traffic is 4 B (demand) + 4 B (reward) per scenario-period
// plus, when training, the stored activations (4 * (F + 32 * n_hidden + n_out) B).
#include "nic_common.h"
#include "small_rollout_body.h"

namespace {
constexpr int kBlock = 64;
// packed weights staged in LDS once per workgroup: every lane reads the same address (LDS broadcast, conflict-free), the
// reads vectorise to ds_read_b128 and pipeline behind counted lgkmcnt waits.  (Streaming the ~2.3k weights through the
// scalar cache every period left the single wave per SIMD stalled on s_load latency: 72k cycles per period measured.)
constexpr int kMaxPacked = (NIC_SR_HIDDEN * NIC_SR_MAX_INPUTS + NIC_SR_HIDDEN) + 2 * (NIC_SR_HIDDEN * NIC_SR_HIDDEN + NIC_SR_HIDDEN) +
                           (NIC_SR_MAX_OUTPUTS * NIC_SR_HIDDEN + NIC_SR_MAX_OUTPUTS);

__device__ __forceinline__ void stage_weights(float* wlds, const float* __restrict__ weights, const NicSmallRolloutDesc& d) {
    const int n = (NIC_SR_HIDDEN * d.F + NIC_SR_HIDDEN) + (d.n_hidden - 1) * (NIC_SR_HIDDEN * NIC_SR_HIDDEN + NIC_SR_HIDDEN) +
                  (d.n_out * NIC_SR_HIDDEN + d.n_out);
    for (int i = threadIdx.x; i < n; i += kBlock) wlds[i] = weights[i];
    __syncthreads();
}

template <int NL>
__global__ __launch_bounds__(kBlock) void small_rollout_fwd_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                   const float* __restrict__ demand,
                                                                   const float* __restrict__ state0, float* __restrict__ rewards,
                                                                   float* __restrict__ state_final, float* __restrict__ states_hist,
                                                                   float* __restrict__ hidden_hist, float* __restrict__ logits_hist) {
    __shared__ __attribute__((aligned(16))) float wlds[kMaxPacked];
    stage_weights(wlds, weights, d);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= d.n_scenarios) return;
    d.weights = wlds;
    d.demand = demand;
    d.state0 = state0;
    nic::small_rollout_fwd_scenario<NL>(d, rewards, state_final, states_hist, hidden_hist, logits_hist, b);
}

template <int NL>
__global__ __launch_bounds__(kBlock) void small_rollout_bwd_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                   const float* __restrict__ demand,
                                                                   const float* __restrict__ states_hist,
                                                                   const float* __restrict__ hidden_hist,
                                                                   const float* __restrict__ logits_hist, NicTable2 g_reward,
                                                                   float* __restrict__ dz_hidden, float* __restrict__ dz_out) {
    __shared__ __attribute__((aligned(16))) float wlds[kMaxPacked];
    stage_weights(wlds, weights, d);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= d.n_scenarios) return;
    d.weights = wlds;
    d.demand = demand;
    nic::small_rollout_bwd_scenario<NL>(d, states_hist, hidden_hist, logits_hist, g_reward, dz_hidden, dz_out, b);
}

int validate(const NicSmallRolloutDesc* d, const char* who) {
    NIC_REQUIRE(d != nullptr, "%s: null descriptor", who);
    NIC_REQUIRE(d->n_scenarios > 0 && d->ldb >= d->n_scenarios && d->T > 0 && d->t0 >= 0, "%s: bad sizes", who);
    NIC_REQUIRE(d->n_hidden >= 1 && d->n_hidden <= 3, "%s: n_hidden must be 1..3", who);
    NIC_REQUIRE(d->n_out >= 1 && d->n_out <= NIC_SR_MAX_OUTPUTS, "%s: n_out out of range", who);
    NIC_REQUIRE(d->head == 0 || d->head == 1, "%s: unknown head", who);
    NIC_REQUIRE(d->Ws >= 2 && d->Wn >= 0 && d->Wn <= 1 && d->E >= 0 && d->E <= 3, "%s: unsupported chain", who);
    NIC_REQUIRE(d->E == 0 || d->Wn == 1, "%s: echelons need the warehouse", who);
    NIC_REQUIRE(d->F == d->Ws + d->Wn * d->Ww + d->E * d->We && d->F <= NIC_SR_MAX_INPUTS, "%s: F mismatch / too large", who);
    NIC_REQUIRE(d->head == 0 ? d->n_out == 1 : d->n_out == d->E + 2, "%s: n_out does not match the head", who);
    NIC_REQUIRE(d->weights && d->demand && d->underage.p && d->holding.p && d->lead.p, "%s: null buffer", who);
    NIC_REQUIRE(d->Wn == 0 || (d->wh_holding.p && d->wh_lead.p), "%s: null warehouse table", who);
    NIC_REQUIRE(d->E == 0 || (d->ech_holding.p && d->ech_lead.p), "%s: null echelon table", who);
    return 0;
}
}  // namespace

extern "C" {

int nic_small_rollout_fwd(const NicSmallRolloutDesc* d, float* rewards, float* state_final, float* states_hist,
                          float* hidden_hist, float* logits_hist, void* stream) {
    if (int e = validate(d, "nic_small_rollout_fwd")) return e;
    NIC_REQUIRE(d->state0 && rewards && state_final, "nic_small_rollout_fwd: null buffer");
    NIC_REQUIRE(!states_hist || (hidden_hist && logits_hist), "nic_small_rollout_fwd: incomplete history buffers");
    const dim3 grid(nic::ceil_div(d->n_scenarios, kBlock)), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
#define NIC_SR_FWD(NL)                                                                                                     \
    hipLaunchKernelGGL(small_rollout_fwd_kernel<NL>, grid, block, 0, s, *d, d->weights, d->demand, d->state0, rewards,    \
                       state_final, states_hist, hidden_hist, logits_hist)
    if (d->n_hidden == 1) NIC_SR_FWD(1);
    else if (d->n_hidden == 2) NIC_SR_FWD(2);
    else NIC_SR_FWD(3);
#undef NIC_SR_FWD
    return nic::check_launch("nic_small_rollout_fwd");
}

int nic_small_rollout_bwd(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                          const float* logits_hist, NicTable2 g_reward, float* dz_hidden, float* dz_out, void* stream) {
    if (int e = validate(d, "nic_small_rollout_bwd")) return e;
    NIC_REQUIRE(states_hist && hidden_hist && logits_hist && g_reward.p && dz_hidden && dz_out,
                "nic_small_rollout_bwd: null buffer");
    const dim3 grid(nic::ceil_div(d->n_scenarios, kBlock)), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
#define NIC_SR_BWD(NL)                                                                                                     \
    hipLaunchKernelGGL(small_rollout_bwd_kernel<NL>, grid, block, 0, s, *d, d->weights, d->demand, states_hist,           \
                       hidden_hist, logits_hist, g_reward, dz_hidden, dz_out)
    if (d->n_hidden == 1) NIC_SR_BWD(1);
    else if (d->n_hidden == 2) NIC_SR_BWD(2);
    else NIC_SR_BWD(3);
#undef NIC_SR_BWD
    return nic::check_launch("nic_small_rollout_bwd");
}
}
