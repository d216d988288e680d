// Whole-horizon rollout kernels for the small (one-store chain, 32-wide MLP) policies: ONE launch runs all T periods.
// One wavefront = 32 scenarios; the 32-wide layers run on the matrix cores with the weights resident in VGPRs, the env
// step / head are per-lane code shared with the host build (small_rollout_body.h, which also holds the scalar per-scenario
// restatement of the whole kernel that tests/hostsim checks against the golden vectors).
// HBM traffic is 4 B (demand) + 4 B (reward) per scenario-period plus, when training, the stored activations
// (4 * (F + 32 * n_hidden + n_out) B).  The backward sweep contracts the weight gradients itself (template WG: accumulators in
// registers over the whole horizon, one partial gradient per wavefront), and the two chains the reference ships are compiled with
// their structure as constants (template SHAPE) - a kernel whose duration is the instruction stream of ONE wavefront per SIMD
// cannot afford the select chains of run-time pipeline offsets.
#include <stdlib.h>

#include "nic_common.h"
#include "small_rollout16.h"
#include "small_rollout_body.h"

namespace {
constexpr int kBlock = 64;
// ---------------------------------------------------------------------------------------------------------------
// MFMA form of the whole-horizon FORWARD: one wavefront = 32 scenarios; the 32-wide layers run on the matrix cores with
// the weights RESIDENT IN VGPRs for the whole horizon (no per-period weight traffic at all — the per-lane VALU form above
// re-reads ~2.3k weights from LDS every period and is bound by that latency).
//   layer output D[neuron i][scenario j] = sum_k W[i][k] H[k][j]:  A = W (lane l holds A[i = l&31][kk = l>>5]),
//   B = activations (lane l holds B[kk = l>>5][j = l&31]), C/D: lane (j, h) holds rows row(r, h) = (r&3) + 8(r>>2) + 4h.
//   MFMA step s of the NEXT layer is defined to contract over k = row(s, h): its B operand is then exactly register s of
//   the previous layer's accumulator — no data movement between layers; the A fragments are loaded in that k order once.
//   The bias is the initial accumulator.  Both halves of the wave carry the same 32 scenarios (different neuron rows), so
//   the per-scenario state and the env step are simply replicated in both halves.
// ---------------------------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ELU of an accumulator: the series of nic::expm1_neg on PAIRS of elements with packed FP32 FMAs (v_pk_fma_f32: two lanes' worth
// of work per instruction; each component rounds exactly like the scalar fmaf chain, so the values are those of nic::elu1).  The
// activations are more than half of the forward kernel's instruction stream (48 per scenario-period).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void elu16(const f32x16& z, float (&out)[16]) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 x = {z[r], z[r + 1]};
        f32x2 p = __builtin_elementwise_fma(x, (f32x2)(1.f / 720.f), (f32x2)(1.f / 120.f));
        p = __builtin_elementwise_fma(x, p, (f32x2)(1.f / 24.f));
        p = __builtin_elementwise_fma(x, p, (f32x2)(1.f / 6.f));
        p = __builtin_elementwise_fma(x, p, (f32x2)(0.5f));
        p = __builtin_elementwise_fma(x, p, (f32x2)(1.f));
        const f32x2 sp = x * p;
        const float e0 = __expf(x.x) - 1.f, e1 = __expf(x.y) - 1.f;
        const float n0 = x.x > -0.35f ? sp.x : e0, n1 = x.y > -0.35f ? sp.y : e1;
        out[r] = x.x > 0.f ? x.x : n0;
        out[r + 1] = x.y > 0.f ? x.y : n1;
    }
}

// Shape specialisation.  The per-lane env step / head bodies (small_rollout_body.h) index the st[16] register array with offsets
// built from the descriptor's pipeline lengths; with those as run-time values every access is a 16-way select chain whose 64-bit
// lane masks live in (spilled) SGPRs - 4,100 instructions per period, a quarter of them v_readlane / s_nop spill traffic, on a
// kernel whose duration IS the instruction count of one wavefront (one wave per SIMD, T sequential periods).  The two chains the
// reference ships are therefore compiled with their structure as constants: overwriting the structural fields of the kernel's
// own copy of the descriptor lets constant propagation fold every offset, loop bound and select through the always-inline bodies.
//   SHAPE 0: any supported chain (run-time structure)
//   SHAPE 1: one store, Ws = 4, softplus head            (one_store_lost.yml / one_store_backlogged.yml + vanilla_one_store)
//   SHAPE 2: store + warehouse + 2 echelons, 4 / 3 / 4, serial head   (serial_system.yml + vanilla_serial)
template <int SHAPE>
__device__ __forceinline__ void fix_shape(NicSmallRolloutDesc& d, int n_hidden) {
    d.n_hidden = n_hidden;
    if (SHAPE == 1) {
        d.Ws = 4; d.Ww = 0; d.We = 0; d.Wn = 0; d.E = 0; d.head = 0; d.F = 4; d.n_out = 1;
    } else if (SHAPE == 2) {
        d.Ws = 4; d.Ww = 3; d.We = 4; d.Wn = 1; d.E = 2; d.head = 1; d.F = 15; d.n_out = 4;
    }
}
int shape_of(const NicSmallRolloutDesc& d) {
    if (d.Ws == 4 && d.Wn == 0 && d.E == 0 && d.head == 0 && d.F == 4 && d.n_out == 1) return 1;
    if (d.Ws == 4 && d.Wn == 1 && d.Ww == 3 && d.E == 2 && d.We == 4 && d.head == 1 && d.F == 15 && d.n_out == 4) return 2;
    return 0;
}

template <int NL, int SHAPE>
__global__ __launch_bounds__(64) void small_rollout_fwd_mfma_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                    const float* __restrict__ demand,
                                                                    const float* __restrict__ state0, float* __restrict__ rewards,
                                                                    float* __restrict__ state_final, float* __restrict__ states_hist,
                                                                    float* __restrict__ hidden_hist, float* __restrict__ logits_hist) {
    using namespace nic;
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int64_t b_raw = (int64_t)blockIdx.x * 32 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;  // dead lanes shadow scenario 0 (they take part in the MFMAs but never store)
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    d.weights = weights;
    d.demand = demand;
    d.state0 = state0;
    fix_shape<SHAPE>(d, NL);

    // ---- weight fragments, resident for the whole horizon
    const int i = j;  // A-operand row owned by this lane
    float aW1[8], cB[NL + 1][16];
    float aWh[(NL > 1 ? NL - 1 : 1)][16], aWo[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) aW1[s] = (2 * s + h < d.F) ? weights[i * d.F + 2 * s + h] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) cB[0][r] = weights[SR_H * d.F + crow(r, h)];
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float* Wl = weights + sr_hidden_offset(d, l);
#pragma unroll
        for (int s = 0; s < 16; ++s) aWh[l - 1][s] = Wl[i * SR_H + crow(s, h)];
#pragma unroll
        for (int r = 0; r < 16; ++r) cB[l][r] = Wl[SR_H * SR_H + crow(r, h)];
    }
    {
        const float* Wo = weights + sr_out_offset(d);
#pragma unroll
        for (int s = 0; s < 16; ++s) aWo[s] = (i < d.n_out) ? Wo[i * SR_H + crow(s, h)] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cB[NL][r] = (crow(r, h) < d.n_out) ? Wo[d.n_out * SR_H + crow(r, h)] : 0.f;
    }

    const SrStatics c = sr_load_statics(d, b);
    float st[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) st[k] = k < d.F ? state0[(int64_t)k * ldb + b] : 0.f;

    // the demand of period t+1 is fetched at the END of period t, after that period's stores: s_waitcnt counts in issue
    // order, so a load issued before the (conditional) stores of a period could only be waited for together with them,
    // and a load issued right before its use exposes its whole latency every period
    float dem = demand[(int64_t)d.t0 * ldb + b];
    for (int t = 0; t < d.T; ++t) {
        f32x16 acc;
        float hcur[16];
        // layer 1: contraction over the state slots, k = 2s + h
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = cB[0][r];
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aW1[s], h ? st[2 * s + 1] : st[2 * s], acc, 0, 0, 0);
        elu16(acc, hcur);
        if (states_hist && live) {
            if (h == 0) {
#pragma unroll
                for (int k = 0; k < SR_MAXF; ++k)
                    if (k < d.F) states_hist[k * tl + t * ldb + b] = st[k];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) hidden_hist[(int64_t)crow(r, h) * tl + t * ldb + b] = hcur[r];
        }
#pragma unroll
        for (int l = 1; l < NL; ++l) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = cB[l][r];
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aWh[l - 1][s], hcur[s], acc, 0, 0, 0);
            elu16(acc, hcur);
            if (states_hist && live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hidden_hist[(int64_t)(l * SR_H + crow(r, h)) * tl + t * ldb + b] = hcur[r];
            }
        }
        // output layer: rows 0..3 land in half 0 (registers 0..3), rows 4..7 in half 1
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = cB[NL][r];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aWo[s], hcur[s], acc, 0, 0, 0);
        float z[SR_MAXOUT];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float mine = acc[n], other = __shfl_xor(mine, 32);
            z[n] = h == 0 ? mine : other;
            z[n + 4] = h == 0 ? other : mine;
        }
        if (logits_hist && live && h == 0) {
#pragma unroll
            for (int n = 0; n < SR_MAXOUT; ++n)
                if (n < d.n_out) logits_hist[n * tl + t * ldb + b] = z[n];
        }
        const SrOrders o = sr_head(d, z, st);
        float nx[SR_MAXF];
        const float cost = sr_env_fwd(d, c, st, nx, dem, o);
        if (live && h == 0) rewards[(int64_t)t * ldb + b] = cost;
        dem = demand[(int64_t)(t + 1 < d.T ? t + 1 + d.t0 : t + d.t0) * ldb + b];
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) st[k] = nx[k];
    }
    if (live && h == 0) {
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k)
            if (k < d.F) state_final[(int64_t)k * ldb + b] = st[k];
    }
}

// MFMA form of the whole-horizon BACKWARD sweep: same wave shape as the forward.  The transposed weights are the resident
// A fragments (dH_{l-1} = W_l^T dZ_l), contraction order over neurons n = crow(s, h) so the previous layer's dZ accumulator
// registers are again the B operands as they stand.  elu'(H) comes from the stored activations, read in the C layout the
// forward wrote them in.  Emits the same dZ history as the per-lane form for the weight-gradient GEMMs.
//
// WG = true: the weight gradients are contracted HERE instead of in four GEMM launches over a dZ history.  Per period and layer
// the pre-activation gradient (column-owner accumulator registers) goes through a wave-private LDS tile into the row-owner A
// operand, the layer input (the stored activation already in registers for elu') through a second tile into the row-owner B
// operand, and 16 MFMAs add dZ X^T over the wave's 32 scenarios into an accumulator that lives in registers for the whole
// horizon (one wavefront per SIMD: the register file is otherwise idle, and these MFMA chains are independent of the
// latency-bound dH chain, whose bubbles they fill).  No dz history is written (97 rows x T x ldb x 4 B for cfg2) or read back.
// At the end the wavefront stores its partial gradient, in the layout of the packed weights, to `slab[blockIdx.x]`.
template <int NL, bool WG, int SHAPE>
__global__ __launch_bounds__(64) void small_rollout_bwd_mfma_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                    const float* __restrict__ demand,
                                                                    const float* __restrict__ states_hist,
                                                                    const float* __restrict__ hidden_hist,
                                                                    const float* __restrict__ logits_hist, NicTable2 g_reward,
                                                                    float* __restrict__ dz_hidden, float* __restrict__ dz_out,
                                                                    float* __restrict__ slab, int64_t slab_stride) {
    using namespace nic;
    __shared__ float tiles[WG ? 2 * 32 * 33 : 1];
    float* const tA = tiles;
    float* const tB = tiles + (WG ? 32 * 33 : 0);
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int64_t b_raw = (int64_t)blockIdx.x * 32 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;  // dead lanes shadow scenario 0: they compute (and, where stores are unconditional, write) exactly what its own lane does
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    d.weights = weights;
    d.demand = demand;
    fix_shape<SHAPE>(d, NL);

    const int i = j;  // A-operand row: the INPUT feature of the layer being back-propagated through
    float aWoT[4], aWhT[(NL > 1 ? NL - 1 : 1)][16], aW1T[16];
    {
        const float* Wo = weights + sr_out_offset(d);
#pragma unroll
        for (int s = 0; s < 4; ++s) aWoT[s] = (2 * s + h < d.n_out) ? Wo[(2 * s + h) * SR_H + i] : 0.f;
    }
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float* Wl = weights + sr_hidden_offset(d, l);
#pragma unroll
        for (int s = 0; s < 16; ++s) aWhT[l - 1][s] = Wl[crow(s, h) * SR_H + i];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) aW1T[s] = (i < d.F) ? weights[crow(s, h) * d.F + i] : 0.f;

    const SrStatics c = sr_load_statics(d, b);
    // WG: a dead lane (shadowing scenario 0) gets a zero cost gradient, so every dz it produces is exactly zero and it adds
    // nothing to the weight gradients - no per-element masking of the A operands.  (The dz-history form must NOT do this: its
    // dead lanes store what scenario 0's own lane stores.)
    const float gr = (WG && !live) ? 0.f : g_reward.p[b * g_reward.scn_stride];
    float gn[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) gn[k] = 0.f;

    // The sweep is a chain of dependent HBM round trips if each period loads its own history (one wavefront per SIMD: nothing
    // else to switch to), so period t-1's history is fetched into registers while period t is computed.
    float st[SR_MAXF], z[SR_MAXOUT], dem, hh[NL][16];
    float st_n[SR_MAXF], z_n[SR_MAXOUT], dem_n, hh_n[NL][16];
    // in-kernel weight gradients: dW of the output layer / hidden layers 1.. / first layer, bias sums of row i = j
    f32x16 gO, gH[(NL > 1 ? NL - 1 : 1)], g1;
    float sbO = 0.f, sbH[(NL > 1 ? NL - 1 : 1)], sb1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gO[r] = 0.f;
        g1[r] = 0.f;
#pragma unroll
        for (int l = 0; l < (NL > 1 ? NL - 1 : 1); ++l) gH[l][r] = 0.f;
    }
#pragma unroll
    for (int l = 0; l < (NL > 1 ? NL - 1 : 1); ++l) sbH[l] = 0.f;
    // column-owner registers (rows crow(r, h) of scenario column j) -> row-owner operand (row j, columns h*16 + s); dead lanes
    // shadow scenario 0 and must not be counted twice: their dz are zero (see gr)
    auto to_A = [&](const float (&v)[16], float (&out)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tA[crow(r, h) * 33 + j] = v[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) out[s2] = tA[j * 33 + h * 16 + s2];
        __builtin_amdgcn_wave_barrier();
    };
    auto to_B = [&](const float (&v)[16], float (&out)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tB[crow(r, h) * 33 + j] = v[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) out[s2] = tB[j * 33 + h * 16 + s2];
        __builtin_amdgcn_wave_barrier();
    };
    auto wgrad = [&](f32x16& g, float& sb, const float (&at)[16], const float (&bt)[16]) {
        float s_ = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            g = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s2], bt[s2], g, 0, 0, 0);
            s_ += at[s2];
        }
        sb += s_ + __shfl_xor(s_, 32);
    };
    // every load of a fetch is unconditional (rows beyond F / n_out re-read the last valid row and are zeroed by a select) and
    // the fetch sits AFTER the period's only conditional stores (dz_out), followed by unconditional dz_hidden stores: the
    // number of memory operations issued after it is then a compile-time constant and the wait for it at the top of the
    // next period (s_waitcnt vmcnt(N), issue order) does not also wait for those stores to complete
    auto fetch = [&](int t, float (&fs)[SR_MAXF], float (&fz)[SR_MAXOUT], float& fd, float (&fh)[NL][16]) {
        const int64_t at = (int64_t)t * ldb + b;
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) {
            const float v = states_hist[(k < d.F ? k : d.F - 1) * tl + at];
            fs[k] = k < d.F ? v : 0.f;
        }
#pragma unroll
        for (int n = 0; n < SR_MAXOUT; ++n) {
            const float v = logits_hist[(n < d.n_out ? n : d.n_out - 1) * tl + at];
            fz[n] = n < d.n_out ? v : 0.f;
        }
        fd = demand[(int64_t)(t + d.t0) * ldb + b];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) fh[l][r] = hidden_hist[(int64_t)(l * SR_H + crow(r, h)) * tl + at];
    };
    fetch(d.T - 1, st, z, dem, hh);
    for (int t = d.T - 1; t >= 0; --t) {
        const int64_t at = (int64_t)t * ldb + b;
        const SrOrders o = sr_head(d, z, st);
        float go[SR_MAXF], dz[SR_MAXOUT];
        const SrOrders g = sr_env_bwd(d, c, st, gn, go, dem, o, gr);
        sr_head_bwd(d, z, st, g, dz, go);
        if (!WG) {
            if (live && h == 0) {
#pragma unroll
                for (int n = 0; n < SR_MAXOUT; ++n)
                    if (n < d.n_out) dz_out[n * tl + at] = dz[n];
            }
        }
        fetch(t > 0 ? t - 1 : 0, st_n, z_n, dem_n, hh_n);
        if (WG) {  // output layer: dWout += dz_out H_last^T (both halves hold the same per-scenario dz: half 0 writes the tile)
            float a_[16], b_[16];
            if (h == 0) {
#pragma unroll
                for (int n = 0; n < SR_MAXOUT; ++n) tA[n * 33 + j] = n < d.n_out ? dz[n] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) a_[s2] = j < SR_MAXOUT ? tA[j * 33 + h * 16 + s2] : 0.f;
            __builtin_amdgcn_wave_barrier();
            to_B(hh[NL - 1], b_);
            wgrad(gO, sbO, a_, b_);
        }
        // output layer -> last hidden layer: contraction over the n_out logits, n = 2s + h
        f32x16 acc;
        float dh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aWoT[s], h ? dz[2 * s + 1] : dz[2 * s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dh[r] = acc[r] * elu1_grad_from_out(hh[NL - 1][r]);
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            if (WG) {
                float a_[16], b_[16];
                to_A(dh, a_);
                to_B(hh[l - 1], b_);
                wgrad(gH[l - 1], sbH[l - 1], a_, b_);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) dz_hidden[(int64_t)(l * SR_H + crow(r, h)) * tl + at] = dh[r];  // (dead lanes: see b)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aWhT[l - 1][s], dh[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dh[r] = acc[r] * elu1_grad_from_out(hh[l - 1][r]);
        }
        if (WG) {  // first layer: dW1 += dz1 X^T, X = the period's state rows (per-lane scalars, rows >= F are zero)
            float a_[16], b_[16];
            to_A(dh, a_);
            if (h == 0) {
#pragma unroll
                for (int k = 0; k < SR_MAXF; ++k) tB[k * 33 + j] = st[k];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) b_[s2] = j < SR_MAXF ? tB[j * 33 + h * 16 + s2] : 0.f;
            __builtin_amdgcn_wave_barrier();
            wgrad(g1, sb1, a_, b_);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dz_hidden[(int64_t)crow(r, h) * tl + at] = dh[r];
        }
        // first layer -> state (the reference detaches vanilla_serial's MLP input, neural_networks.py:329)
        if (!d.detach_input) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aW1T[s], dh[s], acc, 0, 0, 0);
            // state rows 0..15 live in registers 0..7: half 0 holds rows 0-3, 8-11; half 1 rows 4-7, 12-15
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float mine = acc[r], other = __shfl_xor(mine, 32);
                const int lo = (r & 3) + 8 * (r >> 2);  // row held by half 0 in register r; half 1 holds lo + 4
                go[lo] += h == 0 ? mine : other;
                go[lo + 4] += h == 0 ? other : mine;
            }
        }
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) {
            gn[k] = go[k];
            st[k] = st_n[k];
        }
#pragma unroll
        for (int n = 0; n < SR_MAXOUT; ++n) z[n] = z_n[n];
        dem = dem_n;
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) hh[l][r] = hh_n[l][r];
    }
    if (WG) {  // this wavefront's partial gradient, in the packed-weight layout
        float* S = slab + (int64_t)blockIdx.x * slab_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = crow(r, h);
            if (j < d.F) S[n * d.F + j] = g1[r];
#pragma unroll
            for (int l = 1; l < NL; ++l) S[sr_hidden_offset(d, l) + n * SR_H + j] = gH[l - 1][r];
            if (n < d.n_out) S[sr_out_offset(d) + n * SR_H + j] = gO[r];
        }
        if (h == 0) {
            S[SR_H * d.F + i] = sb1;
#pragma unroll
            for (int l = 1; l < NL; ++l) S[sr_hidden_offset(d, l) + SR_H * SR_H + i] = sbH[l - 1];
            if (i < d.n_out) S[sr_out_offset(d) + d.n_out * SR_H + i] = sbO;
        }
    }
}

// scenarios per wavefront: 32 unless the descriptor asks for the 16-scenario form (whose hidden-activation history is in a
// wave-native order private to its forward / backward pair, so the choice is the caller's and the same for both launches)
int lane_width(const NicSmallRolloutDesc& d, bool) { return d.lane_scenarios == 16 ? 16 : 32; }

int validate(const NicSmallRolloutDesc* d, const char* who) {
    NIC_REQUIRE(d != nullptr, "%s: null descriptor", who);
    NIC_REQUIRE(d->n_scenarios > 0 && d->ldb >= d->n_scenarios && d->T > 0 && d->t0 >= 0, "%s: bad sizes", who);
    NIC_REQUIRE(d->n_hidden >= 1 && d->n_hidden <= 3, "%s: n_hidden must be 1..3", who);
    NIC_REQUIRE(d->n_out >= 1 && d->n_out <= NIC_SR_MAX_OUTPUTS, "%s: n_out out of range", who);
    NIC_REQUIRE(d->head == 0 || d->head == 1, "%s: unknown head", who);
    NIC_REQUIRE(d->lane_scenarios == 0 || d->lane_scenarios == 16 || d->lane_scenarios == 32, "%s: lane_scenarios must be 0, 16 or 32", who);
    NIC_REQUIRE(d->lane_scenarios != 16 || d->ldb % 16 == 0, "%s: the 16-scenario form needs ldb to be a multiple of 16", who);
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
    hipStream_t s = nic::as_stream(stream);
    if (lane_width(*d, false) == 16) {
        nic::small_rollout16_fwd(*d, shape_of(*d), rewards, state_final, states_hist, hidden_hist, logits_hist, s);
        return nic::check_launch("nic_small_rollout_fwd");
    }
    {  // matrix-core form: 32 scenarios per wavefront
        const int shape = shape_of(*d);
        nic::note_kernelf("small_rollout_fwd_mfma_kernel<%d,%s>", d->n_hidden, shape == 1 ? "one_store" : (shape == 2 ? "serial" : "any"));
        const dim3 g32(nic::ceil_div(d->n_scenarios, 32)), b64(64);
#define NIC_SR_FWD_MFMA(NL, SH)                                                                                             \
    hipLaunchKernelGGL((small_rollout_fwd_mfma_kernel<NL, SH>), g32, b64, 0, s, *d, d->weights, d->demand, d->state0, rewards, \
                       state_final, states_hist, hidden_hist, logits_hist)
        if (shape == 1 && d->n_hidden == 3) NIC_SR_FWD_MFMA(3, 1);
        else if (shape == 1 && d->n_hidden == 2) NIC_SR_FWD_MFMA(2, 1);
        else if (shape == 2 && d->n_hidden == 2) NIC_SR_FWD_MFMA(2, 2);
        else if (shape == 2 && d->n_hidden == 3) NIC_SR_FWD_MFMA(3, 2);
        else if (d->n_hidden == 1) NIC_SR_FWD_MFMA(1, 0);
        else if (d->n_hidden == 2) NIC_SR_FWD_MFMA(2, 0);
        else NIC_SR_FWD_MFMA(3, 0);
#undef NIC_SR_FWD_MFMA
        return nic::check_launch("nic_small_rollout_fwd");
    }
}

int nic_small_rollout_bwd(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                          const float* logits_hist, NicTable2 g_reward, float* dz_hidden, float* dz_out, void* stream) {
    if (int e = validate(d, "nic_small_rollout_bwd")) return e;
    NIC_REQUIRE(d->lane_scenarios != 16, "nic_small_rollout_bwd: the dz-history sweep reads the [row][t][ldb] history of the 32-scenario forward");
    NIC_REQUIRE(states_hist && hidden_hist && logits_hist && g_reward.p && dz_hidden && dz_out,
                "nic_small_rollout_bwd: null buffer");
    hipStream_t s = nic::as_stream(stream);
    {
        const int shape = shape_of(*d);
        nic::note_kernelf("small_rollout_bwd_mfma_kernel<%d,%s>", d->n_hidden, shape == 1 ? "one_store" : (shape == 2 ? "serial" : "any"));
        const dim3 g32(nic::ceil_div(d->n_scenarios, 32)), b64(64);
#define NIC_SR_BWD_MFMA(NL, SH)                                                                                              \
    hipLaunchKernelGGL((small_rollout_bwd_mfma_kernel<NL, false, SH>), g32, b64, 0, s, *d, d->weights, d->demand, states_hist, \
                       hidden_hist, logits_hist, g_reward, dz_hidden, dz_out, (float*)nullptr, (int64_t)0)
        if (shape == 1 && d->n_hidden == 3) NIC_SR_BWD_MFMA(3, 1);
        else if (shape == 2 && d->n_hidden == 2) NIC_SR_BWD_MFMA(2, 2);
        else if (d->n_hidden == 1) NIC_SR_BWD_MFMA(1, 0);
        else if (d->n_hidden == 2) NIC_SR_BWD_MFMA(2, 0);
        else NIC_SR_BWD_MFMA(3, 0);
#undef NIC_SR_BWD_MFMA
        return nic::check_launch("nic_small_rollout_bwd");
    }
}

/* (one partial-gradient row per wavefront; sized for the 16-scenario form - the 32-scenario form fills the first half and the
 * caller's buffer is zero elsewhere) */
int nic_small_rollout_bwd_wgrad_slots(int32_t n_scenarios) { return nic::ceil_div(n_scenarios, 16); }

int nic_small_rollout_bwd_wgrad(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                                const float* logits_hist, NicTable2 g_reward, float* slab, int64_t slab_stride, void* stream) {
    if (int e = validate(d, "nic_small_rollout_bwd_wgrad")) return e;
    NIC_REQUIRE(states_hist && hidden_hist && logits_hist && g_reward.p && slab, "nic_small_rollout_bwd_wgrad: null buffer");
    const int n_packed = (nic::SR_H * d->F + nic::SR_H) + (d->n_hidden - 1) * (nic::SR_H * nic::SR_H + nic::SR_H) + (d->n_out * nic::SR_H + d->n_out);
    NIC_REQUIRE(slab_stride >= n_packed, "nic_small_rollout_bwd_wgrad: slab rows (%lld) shorter than the packed weights (%d)",
                (long long)slab_stride, n_packed);
    hipStream_t s = nic::as_stream(stream);
    const int shape = shape_of(*d);
    if (lane_width(*d, true) == 16) {
        nic::small_rollout16_bwd_wgrad(*d, shape, states_hist, hidden_hist, logits_hist, g_reward, slab, slab_stride, s);
        return nic::check_launch("nic_small_rollout_bwd_wgrad");
    }
    nic::note_kernelf("small_rollout_bwd_mfma_kernel<%d,wgrad,%s>", d->n_hidden, shape == 1 ? "one_store" : (shape == 2 ? "serial" : "any"));
    const dim3 g32(nic::ceil_div(d->n_scenarios, 32)), b64(64);
#define NIC_SR_BWD_WG(NL, SH)                                                                                                 \
    hipLaunchKernelGGL((small_rollout_bwd_mfma_kernel<NL, true, SH>), g32, b64, 0, s, *d, d->weights, d->demand, states_hist, \
                       hidden_hist, logits_hist, g_reward, (float*)nullptr, (float*)nullptr, slab, slab_stride)
    if (shape == 1 && d->n_hidden == 3) NIC_SR_BWD_WG(3, 1);
    else if (shape == 1 && d->n_hidden == 2) NIC_SR_BWD_WG(2, 1);
    else if (shape == 2 && d->n_hidden == 2) NIC_SR_BWD_WG(2, 2);
    else if (shape == 2 && d->n_hidden == 3) NIC_SR_BWD_WG(3, 2);
    else if (d->n_hidden == 1) NIC_SR_BWD_WG(1, 0);
    else if (d->n_hidden == 2) NIC_SR_BWD_WG(2, 0);
    else NIC_SR_BWD_WG(3, 0);
#undef NIC_SR_BWD_WG
    return nic::check_launch("nic_small_rollout_bwd_wgrad");
}
}
