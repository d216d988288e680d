// Whole-horizon rollout of the small policies with SIXTEEN scenarios per wavefront on v_mfma_f32_16x16x4_f32.
//
// The 32-scenario form (small_rollout.hip, v_mfma_f32_32x32x2_f32) gives BASELINE cfg4's 16,384 scenarios per GPU 512
// wavefronts for 1,024 SIMDs, and each wavefront is one serial chain (layer -> ELU -> layer -> ... -> head -> env step) at one
// wavefront per SIMD: half the chip idles and nothing overlaps.  With 16 scenarios per wavefront
//   * there are twice as many wavefronts (cfg4: one per SIMD; cfg2's 32,768 scenarios: two per SIMD, so one wavefront's MFMA
//     phase runs under the other's ELU / env-step phase),
//   * a 32-wide layer is 16 MFMAs of 8 passes instead of 16 of 16 passes (the tile is half as wide), and a lane holds 8 instead
//     of 16 activations per layer (half the ELUs, history loads / stores and accumulator reads per lane).
// Layout (16x16x4: A lane l holds A[m = l & 15][k = l >> 4], B lane l holds B[k = l >> 4][n = l & 15], C/D lane l holds
// D[4 (l >> 4) + i][l & 15] in register i): lane (j, g) = (l & 15, l >> 4) owns scenario j of the wavefront's block; a layer's 32
// outputs are two tiles, element e = 4 * tile + i of a lane is output row R(e, g) = 16 (e >> 2) + 4 g + (e & 3).  MFMA step e of
// the NEXT layer is defined to contract over k = R(e, g): its B operand is then register e of this layer's activations as it
// stands (the same trick as the 32-wide kernels), and the resident A fragments are loaded in that k order once.  The four lane
// groups g replicate the per-scenario state, head and env step (small_rollout_body.h).
// Hidden-activation history: the forward / backward pair of THIS file keeps it in a wave-native order - element e of lane l of
// layer y, period t, 16-scenario block q at ((t * n_blocks + q) * n_hidden + y) * 512 + l * 8 + e (n_blocks = ldb / 16; round 4:
// lane-major, two 16-byte accesses per lane and layer - 2 KB contiguous per wavefront - instead of eight 4-byte ones: in-kernel
// stamps had the history stores at a third of a forward period, ~100 cycles per store instruction) - so that
// every history store / load of a wavefront is one contiguous access (in the [row][t][ldb] order of the 32-wide
// kernels a 16-scenario wavefront would touch four 64-byte pieces per instruction: measured 0.26 of the HBM roofline against
// 0.40).  States and logits go the same way (round 4): a (period, block)'s state slots are [scenario j][Fp] (Fp = F rounded up to
// 4: one 16-byte store per lane - lane group g owns slots 4 g .. 4 g + 3 -, Fp / 4 16-byte loads per lane in the backward,
// instead of F 4-byte accesses of 64 bytes per wavefront each, which also fetched every 128-byte line twice: cfg4's backward moved
// 1.25 x its algorithmic bytes), its logits [scenario j][NOp] (NOp = n_out rounded up to 4; n_out = 1: [scenario j], the old
// position) - `sr16_state_rows` / `sr16_logit_rows` give the row counts the caller allocates ([rows][T][ldb] floats as before).
// Same arithmetic as the 32-wide kernels except the summation order inside a layer's contraction (k order differs): results
// agree to rounding, parity is against the golden vectors / the per-period route as before.
#include "nic_common.h"
#include "small_rollout16.h"
#include "small_rollout_body.h"

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;

// In-kernel timestamps of the forward kernel (tuning build only - tools/small_rollout_stamp_probe.py compiles its own copy of the
// library with -DNIC_TUNING_BUILD; the product library contains none of this): the 100 MHz wall clock of workgroups 0 and 1 at 10
// points of every period, [t][block < 2][point].
#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_sr16_stamps = nullptr;
#define SR_STAMP(t, point)                                                                                          \
    do {                                                                                                            \
        if (g_sr16_stamps != nullptr && blockIdx.x < 2 && lane == 0) g_sr16_stamps[((t) * 2 + blockIdx.x) * 16 + (point)] = wall_clock64(); \
    } while (0)
#else
#define SR_STAMP(t, point) do { } while (0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int row16(int e, int g) { return 16 * (e >> 2) + 4 * g + (e & 3); }
__device__ __forceinline__ float sel4(int g, float a, float b, float c, float d) {
    const float lo = (g & 1) ? b : a, hi = (g & 1) ? d : c;
    return (g & 2) ? hi : lo;
}

// ELU of 8 activations: the packed-FMA series of small_rollout.hip's elu16 (same values as nic::elu1)
__device__ __forceinline__ void elu8(const f32x4& z0, const f32x4& z1, float (&out)[8]) {
    const float z[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
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

template <int SHAPE>
__device__ __forceinline__ void fix_shape16(NicSmallRolloutDesc& d, int n_hidden) {
    d.n_hidden = n_hidden;
    if (SHAPE == 1) {
        d.Ws = 4; d.Ww = 0; d.We = 0; d.Wn = 0; d.E = 0; d.head = 0; d.F = 4; d.n_out = 1;
    } else if (SHAPE == 2) {
        d.Ws = 4; d.Ww = 3; d.We = 4; d.Wn = 1; d.E = 2; d.head = 1; d.F = 15; d.n_out = 4;
    }
}

// one 32-wide layer on resident fragments: acc[tile] = bias + sum over the 8 steps of A[tile][e] * x[e]
__device__ __forceinline__ void layer32(const float (&aW)[2][8], const float (&bias)[8], const float (&x)[8], f32x4 (&acc)[2]) {
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        acc[ot] = f32x4{bias[4 * ot], bias[4 * ot + 1], bias[4 * ot + 2], bias[4 * ot + 3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(aW[ot][e], x[e], acc[ot], 0, 0, 0);
    }
}

template <int NL, int SHAPE>
__global__ __launch_bounds__(64, 2) void small_rollout16_fwd_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                    const float* __restrict__ demand,
                                                                    const float* __restrict__ state0, float* __restrict__ rewards,
                                                                    float* __restrict__ state_final, float* __restrict__ states_hist,
                                                                    float* __restrict__ hidden_hist, float* __restrict__ logits_hist) {
    using namespace nic;
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int64_t b_raw = (int64_t)blockIdx.x * 16 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;  // dead lanes shadow scenario 0 (they take part in the MFMAs but never store)
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    d.weights = weights;
    d.demand = demand;
    d.state0 = state0;
    fix_shape16<SHAPE>(d, NL);

    // ---- weight fragments, resident for the whole horizon (A row m = j of each 16-row tile)
    float aW1[2][4], cB[NL + 1][8];
    float aWh[(NL > 1 ? NL - 1 : 1)][2][8], aWo[8];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int s = 0; s < 4; ++s) aW1[ot][s] = (4 * s + g < d.F) ? weights[(16 * ot + j) * d.F + 4 * s + g] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) cB[0][e] = weights[SR_H * d.F + row16(e, g)];
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float* Wl = weights + sr_hidden_offset(d, l);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int e = 0; e < 8; ++e) aWh[l - 1][ot][e] = Wl[(16 * ot + j) * SR_H + row16(e, g)];
#pragma unroll
        for (int e = 0; e < 8; ++e) cB[l][e] = Wl[SR_H * SR_H + row16(e, g)];
    }
    {
        const float* Wo = weights + sr_out_offset(d);
#pragma unroll
        for (int e = 0; e < 8; ++e) aWo[e] = (j < d.n_out) ? Wo[j * SR_H + row16(e, g)] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) cB[NL][i] = (4 * g + i < d.n_out) ? Wo[d.n_out * SR_H + 4 * g + i] : 0.f;
    }

    const SrStatics c = sr_load_statics(d, b);
    float st[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) st[k] = k < d.F ? state0[(int64_t)k * ldb + b] : 0.f;

    const int64_t n_blk = ldb / 16;
    const int Fp = sr16_state_rows(d.F), NOp = sr16_logit_rows(d.n_out);
    // (hidden history of a (period, block, layer): 512 floats, LANE-major - lane l's eight activations are 32 contiguous bytes, so a
    // layer is two 16-byte stores per lane (2 KB contiguous per wavefront) instead of eight 4-byte ones: round 4, see the header)
    auto hstore = [&](int t, int layer, const float (&h)[8]) {
        f32x4* p = reinterpret_cast<f32x4*>(hidden_hist + (((int64_t)t * n_blk + blockIdx.x) * NL + layer) * 512 + lane * 8);
        p[0] = f32x4{h[0], h[1], h[2], h[3]};
        p[1] = f32x4{h[4], h[5], h[6], h[7]};
    };
    float dem = demand[(int64_t)d.t0 * ldb + b];
    for (int t = 0; t < d.T; ++t) {
        f32x4 acc[2];
        float hcur[8];
        SR_STAMP(t, 0);
        // layer 1: contraction over the state slots, k = 4 s + g
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            acc[ot] = f32x4{cB[0][4 * ot], cB[0][4 * ot + 1], cB[0][4 * ot + 2], cB[0][4 * ot + 3]};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(aW1[ot][s], sel4(g, st[4 * s], st[4 * s + 1], st[4 * s + 2], st[4 * s + 3]),
                                                               acc[ot], 0, 0, 0);
        }
        SR_STAMP(t, 1);
        elu8(acc[0], acc[1], hcur);
        SR_STAMP(t, 2);
        if (states_hist) {
            // (dead lanes store too: their slots of the block's wave-native history are read back by the backward's dead lanes)
            if (4 * g < Fp) {
                static_assert(SR_MAXF == 16, "lane group g owns state slots 4 g .. 4 g + 3");
                *reinterpret_cast<f32x4*>(states_hist + (((int64_t)t * n_blk + blockIdx.x) * 16 + j) * Fp + 4 * g) =
                    f32x4{sel4(g, st[0], st[4], st[8], st[12]), sel4(g, st[1], st[5], st[9], st[13]),
                          sel4(g, st[2], st[6], st[10], st[14]), sel4(g, st[3], st[7], st[11], st[15])};
            }
            hstore(t, 0, hcur);
        }
        SR_STAMP(t, 3);
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            layer32(aWh[l - 1], cB[l], hcur, acc);
            if (l == 1) SR_STAMP(t, 4);
            elu8(acc[0], acc[1], hcur);
            if (l == 1) SR_STAMP(t, 5);
            if (states_hist) {
                hstore(t, l, hcur);
            }
        }
        SR_STAMP(t, 6);
        // output layer: one 16-row tile; logit n lives in lane group n >> 2, register n & 3
        f32x4 zo = f32x4{cB[NL][0], cB[NL][1], cB[NL][2], cB[NL][3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) zo = __builtin_amdgcn_mfma_f32_16x16x4f32(aWo[e], hcur[e], zo, 0, 0, 0);
        float z[SR_MAXOUT];
#pragma unroll
        for (int n = 0; n < SR_MAXOUT; ++n) z[n] = __shfl(zo[n & 3], j + 16 * (n >> 2));
        if (logits_hist) {   // lane (j, g) holds logits 4 g .. 4 g + 3 of scenario j as they leave the matrix core
            if (d.n_out == 1) {
                if (live && g == 0) logits_hist[t * ldb + b] = zo[0];
            } else if (4 * g < NOp) {
                *reinterpret_cast<f32x4*>(logits_hist + (((int64_t)t * n_blk + blockIdx.x) * 16 + j) * NOp + 4 * g) = zo;
            }
        }
        SR_STAMP(t, 7);
        const SrOrders o = sr_head(d, z, st);
        SR_STAMP(t, 8);
        float nx[SR_MAXF];
        const float cost = sr_env_fwd(d, c, st, nx, dem, o);
        if (live && g == 0) rewards[(int64_t)t * ldb + b] = cost;
        dem = demand[(int64_t)(t + 1 < d.T ? t + 1 + d.t0 : t + d.t0) * ldb + b];
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k) st[k] = nx[k];
        SR_STAMP(t, 9);
    }
    if (live && g == 0) {
#pragma unroll
        for (int k = 0; k < SR_MAXF; ++k)
            if (k < d.F) state_final[(int64_t)k * ldb + b] = st[k];
    }
}

// Backward sweep with in-kernel weight gradients, 16 scenarios per wavefront.  Transposed weights are the resident A fragments;
// per period and layer the pre-activation gradient and the layer input go through two wave-private LDS tiles ([32 rows][16
// scenarios], row stride 17) into row-owner operands, and 16 MFMAs (4 output tiles x 4 steps of 4 scenarios) add dZ X^T into
// accumulators that stay in registers for the whole horizon.
template <int NL, int SHAPE>
__global__ __launch_bounds__(64, 2) void small_rollout16_bwd_kernel(NicSmallRolloutDesc d, const float* __restrict__ weights,
                                                                 const float* __restrict__ demand,
                                                                 const float* __restrict__ states_hist,
                                                                 const float* __restrict__ hidden_hist,
                                                                 const float* __restrict__ logits_hist, NicTable2 g_reward,
                                                                 float* __restrict__ slab, int64_t slab_stride) {
    using namespace nic;
    constexpr int NH = NL > 1 ? NL - 1 : 1;
    __shared__ float tiles[2 * 32 * 17];
    float* const tA = tiles;
    float* const tB = tiles + 32 * 17;
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int64_t b_raw = (int64_t)blockIdx.x * 16 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;
    const int64_t ldb = d.ldb, tl = (int64_t)d.T * ldb;
    d.weights = weights;
    d.demand = demand;
    fix_shape16<SHAPE>(d, NL);

    // transposed weight fragments: A row m = j of tile ot is INPUT feature 16 ot + j of the layer being back-propagated through
    float aWoT[2][2], aWhT[NH][2][8], aW1T[8];
    {
        const float* Wo = weights + sr_out_offset(d);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int s = 0; s < 2; ++s) aWoT[ot][s] = (4 * s + g < d.n_out) ? Wo[(4 * s + g) * SR_H + 16 * ot + j] : 0.f;
    }
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float* Wl = weights + sr_hidden_offset(d, l);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int e = 0; e < 8; ++e) aWhT[l - 1][ot][e] = Wl[row16(e, g) * SR_H + 16 * ot + j];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) aW1T[e] = (j < d.F) ? weights[row16(e, g) * d.F + j] : 0.f;

    const SrStatics c = sr_load_statics(d, b);
    const float gr = live ? g_reward.p[b * g_reward.scn_stride] : 0.f;  // dead lanes: zero cost gradient -> every dz is zero
    float gn[SR_MAXF];
#pragma unroll
    for (int k = 0; k < SR_MAXF; ++k) gn[k] = 0.f;

    // what a period's backward step reads back: two sets, used alternately (the next period's loads are issued while this period
    // computes; the loop is unrolled by two so that no set is ever copied into the other - that copy was 71 v_mov per period)
    struct Per { float st[SR_MAXF], z[SR_MAXOUT], dem, hh[NL][8]; };
    Per P0, P1;
    float gn1[SR_MAXF];
    // weight-gradient accumulators: tile (ot, kt) of dW = rows 16 ot + 4 g + i, column 16 kt + j; bias sums of row 16 ot + j
    f32x4 gO[2], gH[NH][2][2], g1[2];
    float sbO = 0.f, sbH[NH][2], sb1[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        gO[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        g1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < NH; ++l) {
            sbH[l][q] = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) gH[l][q][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // column-owner registers (rows row16(e, g) of scenario column j) -> tile [row][scenario]
    auto put = [&](float* tile, const float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[row16(e, g) * 17 + j] = v[e];
    };
    // row-owner operand of tile `blk` (rows 16 blk + j): scenarios 4 s + g, s = 0..3
    auto get = [&](const float* tile, int blk, float (&out)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) out[s] = tile[(16 * blk + j) * 17 + 4 * s + g];
    };
    auto rowsum = [&](const float (&a)[4]) {
        float s_ = (a[0] + a[1]) + (a[2] + a[3]);
        s_ += __shfl_xor(s_, 16);
        s_ += __shfl_xor(s_, 32);
        return s_;
    };
    const int64_t n_blk = ldb / 16;
    const int Fp = sr16_state_rows(d.F), NOp = sr16_logit_rows(d.n_out);
    auto fetch = [&](int t, float (&fs)[SR_MAXF], float (&fz)[SR_MAXOUT], float& fd, float (&fh)[NL][8]) {
        const int64_t blk = ((int64_t)t * n_blk + blockIdx.x) * 16 + j;
        const f32x4* sp = reinterpret_cast<const f32x4*>(states_hist + blk * Fp);
#pragma unroll
        for (int s = 0; s < SR_MAXF / 4; ++s) {
            const f32x4 v = sp[4 * s < Fp ? s : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) fs[4 * s + i] = 4 * s + i < d.F ? v[i] : 0.f;
        }
        if (d.n_out == 1) {
            fz[0] = logits_hist[(int64_t)t * ldb + b];
#pragma unroll
            for (int n = 1; n < SR_MAXOUT; ++n) fz[n] = 0.f;
        } else {
            const f32x4* zp = reinterpret_cast<const f32x4*>(logits_hist + blk * NOp);
#pragma unroll
            for (int s = 0; s < SR_MAXOUT / 4; ++s) {
                const f32x4 v = zp[4 * s < NOp ? s : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i) fz[4 * s + i] = 4 * s + i < d.n_out ? v[i] : 0.f;
            }
        }
        fd = demand[(int64_t)(t + d.t0) * ldb + b];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const f32x4* p = reinterpret_cast<const f32x4*>(hidden_hist + (((int64_t)t * n_blk + blockIdx.x) * NL + l) * 512 + lane * 8);
            const f32x4 lo = p[0], hi = p[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                fh[l][e] = lo[e];
                fh[l][4 + e] = hi[e];
            }
        }
    };
    fetch(d.T - 1, P0.st, P0.z, P0.dem, P0.hh);
    // one period: C = this period's set, N = the set the next (earlier) period's loads go to; gin / go = the state gradient
    // arriving from period t + 1 / leaving for period t - 1
    auto step = [&](int t, Per& C, Per& N, const float (&gin)[SR_MAXF], float (&go)[SR_MAXF]) {
        float (&st)[SR_MAXF] = C.st;
        float (&z)[SR_MAXOUT] = C.z;
        float (&hh)[NL][8] = C.hh;
        const float dem = C.dem;
        const SrOrders o = sr_head(d, z, st);
        float dz[SR_MAXOUT];
        const SrOrders gord = sr_env_bwd(d, c, st, gin, go, dem, o, gr);
        sr_head_bwd(d, z, st, gord, dz, go);
        fetch(t > 0 ? t - 1 : 0, N.st, N.z, N.dem, N.hh);
        {   // output layer: dWout (rows n < n_out <= 8 in tile 0) += dz_out H_last^T
            if (g < 2) {   // rows 0..7 hold the logit gradients, rows 8..15 zeros
#pragma unroll
                for (int n = 0; n < SR_MAXOUT; ++n) tA[(8 * g + n) * 17 + j] = (g == 0 && n < d.n_out) ? dz[n] : 0.f;
            }
            put(tB, hh[NL - 1]);
            __builtin_amdgcn_wave_barrier();
            float a_[4], b0[4], b1[4];
            get(tA, 0, a_);
            get(tB, 0, b0);
            get(tB, 1, b1);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                gO[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[s], b0[s], gO[0], 0, 0, 0);
                gO[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[s], b1[s], gO[1], 0, 0, 0);
            }
            sbO += rowsum(a_);
        }
        // output layer -> last hidden layer: contraction over the logits, n = 4 s + g
        f32x4 acc[2];
        float dh[8];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(aWoT[ot][s], sel4(g, dz[4 * s], dz[4 * s + 1], dz[4 * s + 2], dz[4 * s + 3]),
                                                               acc[ot], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dh[e] = acc[e >> 2][e & 3] * elu1_grad_from_out(hh[NL - 1][e]);
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            {   // dW_l += dz_l H_{l-1}^T
                put(tA, dh);
                put(tB, hh[l - 1]);
                __builtin_amdgcn_wave_barrier();
                float a0[4], a1[4], b0[4], b1[4];
                get(tA, 0, a0);
                get(tA, 1, a1);
                get(tB, 0, b0);
                get(tB, 1, b1);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    gH[l - 1][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], gH[l - 1][0][0], 0, 0, 0);
                    gH[l - 1][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b1[s], gH[l - 1][0][1], 0, 0, 0);
                    gH[l - 1][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b0[s], gH[l - 1][1][0], 0, 0, 0);
                    gH[l - 1][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], gH[l - 1][1][1], 0, 0, 0);
                }
                sbH[l - 1][0] += rowsum(a0);
                sbH[l - 1][1] += rowsum(a1);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(aWhT[l - 1][ot][e], dh[e], acc[ot], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dh[e] = acc[e >> 2][e & 3] * elu1_grad_from_out(hh[l - 1][e]);
        }
        {   // first layer: dW1 += dz1 X^T, X = the period's state rows (per-lane scalars, rows >= F are zero)
            put(tA, dh);
            if (g == 0) {
#pragma unroll
                for (int k = 0; k < SR_MAXF; ++k) tB[k * 17 + j] = st[k];
            }
            __builtin_amdgcn_wave_barrier();
            float a0[4], a1[4], b0[4];
            get(tA, 0, a0);
            get(tA, 1, a1);
            get(tB, 0, b0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                g1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], g1[0], 0, 0, 0);
                g1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b0[s], g1[1], 0, 0, 0);
            }
            sb1[0] += rowsum(a0);
            sb1[1] += rowsum(a1);
        }
        // first layer -> state (the reference detaches vanilla_serial's MLP input, neural_networks.py:329)
        if (!d.detach_input) {
            f32x4 gs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; ++e) gs = __builtin_amdgcn_mfma_f32_16x16x4f32(aW1T[e], dh[e], gs, 0, 0, 0);
            // state row k lives in lane group k >> 2, register k & 3
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k) go[k] += __shfl(gs[k & 3], j + 16 * (k >> 2));
        }
    };
    if constexpr (SHAPE != 0 && !(SHAPE == 2 && NL == 3)) {   // shapes compiled in (that fit): two periods per iteration, the sets swap roles
        int t = d.T - 1;
        for (; t >= 1; t -= 2) {
            step(t, P0, P1, gn, gn1);
            step(t - 1, P1, P0, gn1, gn);
        }
        if (t == 0) step(0, P0, P1, gn, gn1);
    } else {   // run-time shapes (and the serial chain with three hidden layers): twice the loop body does not fit the register file
        for (int t = d.T - 1; t >= 0; --t) {
            step(t, P0, P1, gn, gn1);
            P0 = P1;
#pragma unroll
            for (int k = 0; k < SR_MAXF; ++k) gn[k] = gn1[k];
        }
    }
    // this wavefront's partial gradient, in the packed-weight layout: tile (ot, kt) register i = dW[16 ot + 4 g + i][16 kt + j]
    float* S = slab + (int64_t)blockIdx.x * slab_stride;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 16 * ot + 4 * g + i;
            if (j < d.F) S[n * d.F + j] = g1[ot][i];
#pragma unroll
            for (int l = 1; l < NL; ++l)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) S[sr_hidden_offset(d, l) + n * SR_H + 16 * kt + j] = gH[l - 1][ot][kt][i];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = 4 * g + i;
        if (n < d.n_out) {
            S[sr_out_offset(d) + n * SR_H + j] = gO[0][i];
            S[sr_out_offset(d) + n * SR_H + 16 + j] = gO[1][i];
        }
    }
    if (g == 0) {   // bias sums of rows 16 ot + j (every lane group holds the same sums)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            S[SR_H * d.F + 16 * ot + j] = sb1[ot];
#pragma unroll
            for (int l = 1; l < NL; ++l) S[sr_hidden_offset(d, l) + SR_H * SR_H + 16 * ot + j] = sbH[l - 1][ot];
        }
        if (j < d.n_out) S[sr_out_offset(d) + d.n_out * SR_H + j] = sbO;
    }
}
}  // namespace

#ifdef NIC_TUNING_BUILD
extern "C" int nic_tuning_set_small_rollout_stamps(void* buf) {   // buf: device memory, [T][2][16] u64 (or null)
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sr16_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

namespace nic {

void small_rollout16_fwd(const NicSmallRolloutDesc& d, int shape, float* rewards, float* state_final, float* states_hist,
                         float* hidden_hist, float* logits_hist, hipStream_t s) {
    nic::note_kernelf("small_rollout16_fwd_kernel<%d,%s>", d.n_hidden, shape == 1 ? "one_store" : (shape == 2 ? "serial" : "any"));
    const dim3 grid(nic::ceil_div(d.n_scenarios, 16)), block(64);
#define NIC_SR16_FWD(NL, SH)                                                                                              \
    hipLaunchKernelGGL((small_rollout16_fwd_kernel<NL, SH>), grid, block, 0, s, d, d.weights, d.demand, d.state0, rewards,  \
                       state_final, states_hist, hidden_hist, logits_hist)
    if (shape == 1 && d.n_hidden == 3) NIC_SR16_FWD(3, 1);
    else if (shape == 1 && d.n_hidden == 2) NIC_SR16_FWD(2, 1);
    else if (shape == 2 && d.n_hidden == 2) NIC_SR16_FWD(2, 2);
    else if (shape == 2 && d.n_hidden == 3) NIC_SR16_FWD(3, 2);
    else if (d.n_hidden == 1) NIC_SR16_FWD(1, 0);
    else if (d.n_hidden == 2) NIC_SR16_FWD(2, 0);
    else NIC_SR16_FWD(3, 0);
#undef NIC_SR16_FWD
}

void small_rollout16_bwd_wgrad(const NicSmallRolloutDesc& d, int shape, const float* states_hist, const float* hidden_hist,
                               const float* logits_hist, NicTable2 g_reward, float* slab, int64_t slab_stride, hipStream_t s) {
    nic::note_kernelf("small_rollout16_bwd_kernel<%d,wgrad,%s>", d.n_hidden, shape == 1 ? "one_store" : (shape == 2 ? "serial" : "any"));
    const dim3 grid(nic::ceil_div(d.n_scenarios, 16)), block(64);
#define NIC_SR16_BWD(NL, SH)                                                                                               \
    hipLaunchKernelGGL((small_rollout16_bwd_kernel<NL, SH>), grid, block, 0, s, d, d.weights, d.demand, states_hist, hidden_hist, \
                       logits_hist, g_reward, slab, slab_stride)
    if (shape == 1 && d.n_hidden == 3) NIC_SR16_BWD(3, 1);
    else if (shape == 1 && d.n_hidden == 2) NIC_SR16_BWD(2, 1);
    else if (shape == 2 && d.n_hidden == 2) NIC_SR16_BWD(2, 2);
    else if (shape == 2 && d.n_hidden == 3) NIC_SR16_BWD(3, 2);
    else if (d.n_hidden == 1) NIC_SR16_BWD(1, 0);
    else if (d.n_hidden == 2) NIC_SR16_BWD(2, 0);
    else NIC_SR16_BWD(3, 0);
#undef NIC_SR16_BWD
}

}  // namespace nic
