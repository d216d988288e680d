// Whole-horizon rollout of the WIDE vanilla_warehouse policy (BASELINE cfg3: 51 -> 512 -> 512 -> 512 -> 17) - round 5.
//
// Every stage of a period is column-local: a scenario's hidden activations, logits, orders, state and cost depend on that
// scenario's column only (trainer.py:190-213, neural_networks.py:393-426, environment.py:110-299).  The per-period route still
// pays, per period, two tiled GEMM launches (prologue, epilogue and a barrier per k tile that nothing overlaps at the 8,192
// columns of an 8-GPU shard: 0.66 of the FP32 MFMA peak) and the tail launch(es) (latency chains of ~20-30 us on one wavefront
// per SIMD).  Here ONE launch per direction walks all T periods: a workgroup of four wavefronts owns a block of 32 scenarios for
// the whole horizon and nothing but the histories the backward / the weight-gradient GEMMs need leaves the CU.
//   * hidden activations [H][32] live in LDS in "B layout" (a lane reads the four rows 8 g + 2 j + h of a k group as one b128);
//     wave w owns rows [w H / 4, (w + 1) H / 4) of every hidden layer: RT = H / 128 row tiles of 32, 16 accumulator registers each;
//   * the weights of the H x H layers stream from L2 as PRE-PACKED MFMA A fragments ([row tile][k group][lane][4]: one coalesced
//     16-byte load per lane feeds four MFMAs), PF k groups ahead of the MFMAs that consume them; every CU streams the same 1 MB
//     per layer at about the same time (L2 hits): 16 B / clk / CU, half of what a CU can take from its XCD's L2;
//   * the last hidden layer never goes back to LDS: its post-activation accumulators ARE the B operands of the logits
//     contraction (D layout = B layout up to the k order, which the packed logits weights absorb); the four wavefronts' partial
//     logits are added through LDS once;
//   * head + env step on LDS tiles with the NIC_HD bodies (as csrc/period_tail.hip, stage B); the state tile stays in LDS from
//     period to period, the static tables are staged once per block;
//   * the next period's first layer runs from the new state tile (the bias as its row of ones), as period_tail.hip's stage C.
// MFMA-bound by construction: per period and block 2 (H / 128) (H / 2) + ... = 2,216 MFMAs per wavefront at cfg3 = 59 us at
// 2.4 GHz; what leaves the CU per period is 3 H x 32 floats of activation history (training) - 3 % of a CU's HBM share.
// The backward sweep (wide_bwd_kernel) mirrors it with the transposed weights: first layer's input gradient from the registers
// that hold dZ1, env / head adjoints on LDS tiles, the logits layer's input gradient, and the hidden layers' input gradients
// with ELU' read from the activation history; weight gradients stay ordinary (period x scenario) contractions
// (nic_linear_wgrad_periods) over the histories these kernels leave behind.
// Built with -ffp-contract=off (head / env arithmetic rounds like the reference's separate aten ops); MFMA chains are fma by
// construction.  Same arithmetic as the per-period route except the summation order inside the logits contraction.
#include "tail_pieces.h"
#include "../../include/nic_experiments.h"

namespace {

constexpr int kMaxHidden = 4;   // hidden layers (first layer included)

struct WideParams {
    NicEnvStepIO io;          // dims + static tables; the state / demand / order pointers inside are not used (histories below)
    const int32_t* adj;
    float ub;
    int trans;
    int F, T, H, n_hidden, n_out;
    int n_blocks;
    // histories: element (t, row, b) at base + t * period stride + row * ldb + b
    const float* demand;      // [T][S][ld_demand]
    int64_t ps_demand, ld_demand;
    float* states;            // [T + 1][F (+ 1)][ldb]  (block 0 = the initial state, written by the caller)
    float* orders;            // [T][S Wn + Wn][ldb]
    float* logits;            // [T][n_out][ldb]
    float* rewards;           // [T][ldb]
    float* hidden[kMaxHidden];   // [T][H][ldb] per hidden layer, or null (evaluation)
    int64_t ps_state, ps_orders, ps_logits, ps_hidden;
    // weights
    const float* Wt_in;       // first layer transposed [F + 1][ldwt], bias as row F
    int64_t ldwt;
    const float4* Wp[kMaxHidden];   // hidden layer l (1 <= l < n_hidden): packed [H / 32][H / 8][64] float4
    const float* bh[kMaxHidden];    // ... its bias [H]
    const float* Wq;          // logits layer packed for register-fed B operands [H / 32][16][64]
    const float* b_out;       // [n_out] or null
};

// element (row k, column c) of an activation block in LDS
__device__ __forceinline__ int bl(int k, int c) { return (((k >> 3) * 2 + (k & 1)) * 32 + c) * 4 + ((k & 7) >> 1); }

// One H x H layer for the block: acc[i] = rows 32 (wave RT + i) .. + 31 of W hin, K = H.  Weights as packed A fragments through a
// buffer descriptor (one per-lane offset, the (tile, group) position as a scalar offset), PF groups in flight; activations
// from LDS one group ahead.
template <int RT, int PF>
__device__ __forceinline__ void stream_layer(const float4* __restrict__ Wp, int K, int wave, const float* hin, f32x16 (&acc)[RT]) {
    // Two register sets of PF k groups each: while the MFMAs of one set run (PF x 4 RT x 64 cycles = 1.7 us at RT = 4), the
    // loads of the other set are in flight, all of them issued in FRONT of those MFMAs.  (hipcc waits for EVERYTHING outstanding
    // at the head of a loop whose loads cross the back edge - s_waitcnt vmcnt(0) - so loads issued group by group behind the
    // MFMAs that free their registers, the textbook rotation, expose the whole L2 latency once per trip: measured 0.46 of peak.)
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int ng = K / 8;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(reinterpret_cast<const float*>(Wp), (int64_t)K * K);
    const int vw = lane * 16;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int tile_off[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) tile_off[i] = (wave * RT + i) * ng * 1024;
    float4 a0[PF][RT], a1[PF][RT];
    auto fetch = [&](float4 (&a)[PF][RT], int g0) {   // groups g0 .. g0 + PF - 1 (past the end: the last groups again, unused)
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int g = g0 + d < ng ? g0 + d : ng - 1;
#pragma unroll
            for (int i = 0; i < RT; ++i) a[d][i] = buf_load4(rW, vw + tile_off[i] + g * 1024);
        }
    };
    const float4* hb = reinterpret_cast<const float4*>(hin) + h * 32 + c;
    auto run = [&](const float4 (&a)[PF][RT], int g0) {
        float4 bq[2];   // (two named sets: with one, hipcc issues each group's LDS read behind the previous group's last MFMA)
        bq[0] = hb[g0 * 64];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (d + 1 < PF) bq[(d + 1) & 1] = hb[(g0 + d + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);   // (the next group's LDS read stays in FRONT of this group's MFMAs)
            const float4 b = bq[d & 1];
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].w, b.w, acc[i], 0, 0, 0);
        }
    };
    fetch(a0, 0);
    for (int g0 = 0; g0 < ng; g0 += 2 * PF) {   // (ng % (2 PF) == 0)
        fetch(a1, g0 + PF);
        __builtin_amdgcn_sched_barrier(0);
        run(a0, g0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(a0, g0 + 2 * PF);
        __builtin_amdgcn_sched_barrier(0);
        run(a1, g0 + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bias + ELU of a wavefront's row tiles; the results replace the accumulators, go to LDS in B layout (hout) and / or to the
// history block of the period (gout: [H][ldb] at this block's first scenario; columns >= n_store are not stored)
template <int RT>
__device__ __forceinline__ void layer_epilogue(f32x16 (&acc)[RT], int wave, const float* bias_l, float* hout, float* gout, int ldb,
                                               int n_store, int H) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rG = make_rsrc(gout ? gout : hout, gout ? (int64_t)H * ldb : 0);   // (null history: every store dropped)
    const int vg = (4 * h * ldb + c) * 4;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row0 = (wave * RT + i) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ku = (r & 3) + 8 * (r >> 2);
            const int row = row0 + ku + 4 * h;
            const float y = elu_f(acc[i][r] + (bias_l ? bias_l[row] : 0.f));
            acc[i][r] = y;
            if (hout) hout[bl(row, c)] = y;
            if (c < n_store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rG, vg, (row0 + ku) * ldb * 4, 0);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// first layer of a period from the state tile `src` ([F + 1 rows (+ zeros)][LDT], row F = ones): this wavefront's RT row tiles
template <int RT, int KS>
__device__ __forceinline__ void first_layer(const WideParams& p, int wave, const float* src, f32x16 (&acc)[RT]) {
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    float xs[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xs[s] = src[(2 * s + h) * LDT + li];
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.Wt_in, (int64_t)(p.F + 1) * p.ldwt);
    const int lw4 = (int)p.ldwt * 4;
    float a[KS];
    {
        const int vw = h * lw4 + ((wave * RT) * 32 + li) * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = ldf(rW, vw, 2 * s * lw4);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int nxt = wave * RT + (i + 1 < RT ? i + 1 : i);
        const int vw = h * lw4 + (nxt * 32 + li) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xs[s], acc[i], 0, 0, 0);
            a[s] = ldf(rW, vw, 2 * s * lw4);
        }
    }
}

// ================================================================================================================================
// forward
// ================================================================================================================================
template <int MAXW, int KS, int RT>
__global__ __launch_bounds__(kThreads) void wide_fwd_kernel(WideParams p) {
    constexpr int MAXSQ = 4, PF = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = p.H;
    float* buf0 = lds;
    float* buf1 = lds + H * NB;
    float* stA = buf1 + H * NB;                 // state tiles (two, swapped every period)
    float* stB = stA + kStateRows * LDT;
    float* tb = stB + kStateRows * LDT;          // static tables
    int* adj_l = reinterpret_cast<int*>(tb + kTabRows * LDT);
    float* bias_l = reinterpret_cast<float*>(adj_l + 32);   // [n_hidden - 1][H]
    // scratch of the head / env stage: inside buf1 (free whenever it is used, see the period loop)
    float* zt = buf1;
    float* dm = zt + 32 * LDT;
    float* od = dm + 4 * MAXSQ * LDT;
    float* ex = od + 32 * LDT;
    float (*xm)[NB] = reinterpret_cast<float (*)[NB]>(ex);
    float (*xd)[NB] = reinterpret_cast<float (*)[NB]>(ex + 4 * NB);
    int (*xn)[NB] = reinterpret_cast<int (*)[NB]>(ex + 8 * NB);
    float (*rq)[NB] = reinterpret_cast<float (*)[NB]>(ex + 12 * NB);
    float (*cw)[NB] = reinterpret_cast<float (*)[NB]>(ex + 16 * NB);
    float (*part)[4][NB] = reinterpret_cast<float (*)[4][NB]>(ex + 16 * NB + kMaxWh * NB);
    float* red = ex + 16 * NB + kMaxWh * NB + kChunk * 4 * NB;   // 3 x 16 x 64

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const NicEnvDims& d = p.io.dims;
    const int B = d.n_scenarios, S = d.n_stores, Wn = d.n_warehouses, Ww = d.warehouse_slots;
    const int ldb = d.ldb;
    const int ncols = (B + 3) / 4 * 4;
    const int n_ord = S * Wn + Wn;
    const bool active = tid < 4 * NB;
    const int x = tid & (NB - 1), q = (tid >> 5) & 3;

    // logits weights for register-fed B operands: step (i, r) of this wavefront contracts rows 32 (wave RT + i) + crow(r, h)
    float aq[RT][16];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) aq[i][r] = p.Wq[((wave * RT + i) * 16 + r) * 64 + lane];

    for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
        const int c0 = blk * NB;
        const int nlive = min(NB, B - c0);
        const int n_store = min(NB, ncols - c0);   // columns the GEMM-like stages store (as nic_linear_fwd)
        const bool live = active && x < nlive;
        const int bb = x < nlive ? x : nlive - 1;
        float* st = stA;
        float* sn = stB;
        if (blk != (int)blockIdx.x) __syncthreads();
        {   // ---- per block: static tables, adjacency, biases, initial state
            TabRegs ptb;
            float4 pst[2];
            tables_fetch(p.io, c0, nlive, ptb);
            const int padj = p.adj[tid < S * Wn ? tid : 0];
            tile_fetch<2>(p.states, ldb, c0, p.F, pst);
            for (int l = 1; l < p.n_hidden; ++l)
                for (int i = tid; i < H; i += kThreads) bias_l[(l - 1) * H + i] = p.bh[l][i];
            for (int i = tid; i < kStateRows * LDT; i += kThreads) st[i] = (i / LDT == p.F) ? 1.f : 0.f;
            __syncthreads();
            tables_put(tb, p.io, ptb);
            if (tid < 32) adj_l[tid] = padj;
            tile_put<2>(st, p.F, pst);
            __syncthreads();
        }
        const NicEnvStepIO io0 = block_io(p.io, nlive, st, dm, od, tb);
        f32x16 acc[RT];
        // ---- first layer of period 0
        first_layer<RT, KS>(p, wave, st, acc);
        layer_epilogue<RT>(acc, wave, nullptr, buf0, p.hidden[0] ? p.hidden[0] + c0 : nullptr, ldb, n_store, H);
        __syncthreads();

        for (int t = 0; t < p.T; ++t) {
            // demand of the period: requested now, parked in registers until the layers are done
            float4 pdm[1];
            tile_fetch<1>(p.demand + t * p.ps_demand, p.ld_demand, c0, S, pdm);
            // ---- hidden layers 1 .. n_hidden - 1 (the last one stays in registers)
            float* hin = buf0;
            float* hout = buf1;
            for (int l = 1; l < p.n_hidden; ++l) {
                stream_layer<RT, PF>(p.Wp[l], H, wave, hin, acc);
                const bool last = l + 1 == p.n_hidden;
                layer_epilogue<RT>(acc, wave, bias_l + (l - 1) * H, last ? nullptr : hout,
                                   p.hidden[l] ? p.hidden[l] + t * p.ps_hidden + c0 : nullptr, ldb, n_store, H);
                __syncthreads();   // hout complete / every wavefront done with hin (the scratch in buf1 may be written after the last one)
                float* tmp = hin;
                hin = hout;
                hout = tmp;
            }
            // ---- logits: this wavefront's share of the contraction, B operands straight from the accumulators
            {
                f32x16 z[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[0][r] = 0.f;
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i][r], acc[i][r], z[0], 0, 0, 0);
                float aux[16];
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = crow(r, h);
                        aux[r] = (p.b_out != nullptr && row < p.n_out) ? p.b_out[row] : 0.f;
                    }
                }
                ksplit_publish<1>(red, z);
                __syncthreads();
                if (wave == 0) {
                    ksplit_collect<1>(red, z);
                    const __amdgpu_buffer_rsrc_t rZ = make_rsrc(p.logits + t * p.ps_logits + c0, (int64_t)p.n_out * ldb);   // rows >= n_out dropped
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ku = (r & 3) + 8 * (r >> 2);
                        const float y = z[0][r] + aux[r];
                        zt[(ku + 4 * h) * LDT + li] = y;
                        if (li < n_store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rZ, (4 * h * ldb + li) * 4, ku * ldb * 4, 0);
                    }
                }
                tile_put<1>(dm, S, pdm);
                for (int i = tid; i < kStateRows * LDT; i += kThreads) sn[i] = (i / LDT == p.F) ? 1.f : 0.f;
                __syncthreads();
            }
            // ---- head + env step (period_tail.hip, stage B)
            NicEnvStepIO io = io0;
            io.store_inv = st;
            io.wh_inv = st + S * d.store_slots * LDT;
            float* od_wh = od + S * Wn * LDT;
            for (int w = 0; w < Wn; ++w) {
                if (w > 0) nic::lds_barrier();
                nic::HeadLane<MAXSQ> L;
                int nc;
                const float mq = nic::head_quad_load<MAXSQ, false>(L, zt, nullptr, adj_l, S, Wn, LDT, bb, w, q, nc, nullptr);
                if (active) {
                    xm[q][x] = mq;
                    xn[q][x] = nc;
                }
                const float stock = io.wh_inv[w * Ww * LDT + bb];
                nic::lds_barrier();
                const float m = nic::head_quad_max(xm[0][x], xm[1][x], xm[2][x], xm[3][x], p.trans);
                const int n_conn = xn[0][x] + xn[1][x] + xn[2][x] + xn[3][x];
                const float dq = nic::head_quad_exp<MAXSQ>(L, m);
                if (active) xd[q][x] = dq;
                nic::lds_barrier();
                const float denom = nic::head_quad_denom(xd[0][x], xd[1][x], xd[2][x], xd[3][x], m, p.trans);
                if (live) {
                    nic::head_quad_fwd_store<MAXSQ>(L, denom, stock, n_conn, od, S, Wn, LDT, x, w, q);
                    if (q == (w & 3)) nic::head_wh_order_fwd(zt, p.ub, od_wh, S, Wn, LDT, x, w, -1);
                }
            }
            nic::lds_barrier();
            float* sn_wh = sn + S * d.store_slots * LDT;
            {
                const nic::IoAccess ac{io, x, sn, sn_wh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                float r = 0.f;
                if (live)
                    for (int s_ = q; s_ < S; s_ += nic::kQuad) r += nic::env_fwd_store_t<MAXW>(ac, s_);
                if (active) rq[q][x] = r;
            }
            for (int wc = 0; wc < Wn; wc += kChunk) {
                for (int i = 0; i < kChunk && wc + i < Wn; ++i) {
                    const float v = live ? nic::env_ship_partial(io, wc + i, x, q) : 0.f;
                    if (active) part[i][q][x] = v;
                }
                nic::lds_barrier();
                for (int i = q; i < kChunk && wc + i < Wn; i += nic::kQuad) {
                    const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
                    const float c = live ? nic::env_fwd_warehouse<MAXW>(io, sn_wh, wc + i, shipped, x) : 0.f;
                    if (active) cw[wc + i][x] = c;
                }
                nic::lds_barrier();
            }
            if (q == 0 && live) {
                const float total = nic::combine4(rq[0][x], rq[1][x], rq[2][x], rq[3][x]);
                float r_wh = 0.f;
                for (int w = 0; w < Wn; ++w) r_wh += cw[w][x];
                p.rewards[(int64_t)t * ldb + c0 + x] = total + r_wh;
            }
            tile_store(od, p.orders + t * p.ps_orders, ldb, c0, n_ord, nlive);
            tile_store(sn, p.states + (t + 1) * p.ps_state, ldb, c0, p.F, nlive);
            // ---- first layer of period t + 1 from the new state tile
            if (t + 1 < p.T) {
                first_layer<RT, KS>(p, wave, sn, acc);
                layer_epilogue<RT>(acc, wave, nullptr, buf0, p.hidden[0] ? p.hidden[0] + (t + 1) * p.ps_hidden + c0 : nullptr, ldb, n_store,
                                   H);
            }
            __syncthreads();
            float* tmp = st;
            st = sn;
            sn = tmp;
        }
    }
}

// ================================================================================================================================
// backward
// ================================================================================================================================
struct WideBwdParams {
    NicEnvStepIO io;
    const int32_t* adj;
    float ub;
    int trans;
    int F, T, H, n_hidden, n_out;
    int n_blocks;
    const float* demand;
    int64_t ps_demand, ld_demand;
    const float* states;      // histories the forward sweep left
    const float* orders;
    const float* logits;
    const float* hidden[kMaxHidden];
    int64_t ps_state, ps_orders, ps_logits, ps_hidden;
    NicTable2 g_reward;
    float* dZ[kMaxHidden];    // [T][H][ldb] pre-activation gradient history of hidden layer l (written)
    float* dZ_out;            // [T][n_out][ldb] logits gradient history (written)
    int64_t ps_dz, ps_dzout;
    const float4* WpT[kMaxHidden];   // hidden layer l (1 <= l < n_hidden): its TRANSPOSE packed like Wp
    const float* Wq_in;       // first layer for register-fed B operands: [2][H / 32][16][64]: lane (f, h) of (mt, tile, r) holds
                              // W_in^T[32 mt + f][32 tile + crow(r, h)] (0 for state rows >= F)
    const float* Wo_t;        // logits layer transposed fragments: [H / 32][NS][64]: lane (k, h) of (tile, s) holds W_out[2 s + h][32 tile + k]
};

// this wavefront's rows of an activation history block ([H][ldb] at the block's first scenario) in accumulator layout
template <int RT>
__device__ __forceinline__ void load_act(const float* g, int wave, int ldb, int H, float (&v)[RT][16]) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rG = make_rsrc(g, (int64_t)H * ldb);
    const int vg = (4 * h * ldb + c) * 4;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[i][r] = ldf(rG, vg, ((wave * RT + i) * 32 + (r & 3) + 8 * (r >> 2)) * ldb * 4);
}

// acc <- acc * ELU'(act) (act = the layer's post-activation output); to the gradient history (gout) and / or LDS in B layout
template <int RT>
__device__ __forceinline__ void dgrad_epilogue(f32x16 (&acc)[RT], const float (&act)[RT][16], int wave, float* hout, float* gout, int ldb,
                                               int n_store, int H) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rG = make_rsrc(gout, (int64_t)H * ldb);
    const int vg = (4 * h * ldb + c) * 4;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row0 = (wave * RT + i) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ku = (r & 3) + 8 * (r >> 2);
            const float y = acc[i][r] * elu_grad_from_out(act[i][r]);
            acc[i][r] = y;
            if (hout) hout[bl(row0 + ku + 4 * h, c)] = y;
            if (c < n_store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rG, vg, (row0 + ku) * ldb * 4, 0);
        }
    }
}

template <int MAXW, int NS, int RT>
__global__ __launch_bounds__(kThreads) void wide_bwd_kernel(WideBwdParams p) {
    constexpr int MAXSQ = 4, PF = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = p.H;
    float* buf0 = lds;
    float* buf1 = lds + H * NB;
    float* gi = buf1 + H * NB;                  // env part of d loss / d state(t): written by stage B', read by the NEXT iteration
    float* tb = gi + kStateRows * LDT;
    int* adj_l = reinterpret_cast<int*>(tb + kTabRows * LDT);
    // scratch of stages A' / B' (both activation buffers are free then); the logits gradient tile sits in buf1: stage C' reads it
    // while it writes buf0
    float* red = buf0;                           // 3 x 2 x 16 x 64
    float* st = red + 3 * 2 * 16 * 64;
    float* dm = st + kStateRows * LDT;
    float* od = dm + 4 * MAXSQ * LDT;
    float* zt = od + 32 * LDT;
    float* gs = zt + 32 * LDT;                   // [64][LDT]
    float* go = gs + 64 * LDT;
    float* ex = go + 32 * LDT;                   // (ends 16 floats inside buf1)
    float* dz = buf1 + 1024;
    float (*xm)[NB] = reinterpret_cast<float (*)[NB]>(ex);
    float (*xd)[NB] = reinterpret_cast<float (*)[NB]>(ex + 4 * NB);
    float (*xt)[NB] = reinterpret_cast<float (*)[NB]>(ex + 8 * NB);
    float (*xs)[NB] = reinterpret_cast<float (*)[NB]>(ex + 12 * NB);
    float (*gwa)[NB] = reinterpret_cast<float (*)[NB]>(ex + 16 * NB);
    float (*part)[4][NB] = reinterpret_cast<float (*)[4][NB]>(ex + 16 * NB + kMaxWh * NB);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const NicEnvDims& d = p.io.dims;
    const int B = d.n_scenarios, S = d.n_stores, Wn = d.n_warehouses, Ww = d.warehouse_slots;
    const int ldb = d.ldb;
    const int ncols = (B + 3) / 4 * 4;
    const int n_ord = S * Wn + Wn;
    const bool active = tid < 4 * NB;
    const int x = tid & (NB - 1), q = (tid >> 5) & 3;

    // resident fragments: first layer (register-fed contraction of stage A'), logits layer transposed (stage C')
    float aIn[2][RT][16], aO[RT][NS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) aIn[mt][i][r] = p.Wq_in[(((mt * (H / 32)) + wave * RT + i) * 16 + r) * 64 + lane];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s) aO[i][s] = p.Wo_t[((wave * RT + i) * NS + s) * 64 + lane];

    for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
        const int c0 = blk * NB;
        const int nlive = min(NB, B - c0);
        const int n_store = min(NB, ncols - c0);
        const bool live = active && x < nlive;
        const int bb = x < nlive ? x : nlive - 1;
        if (blk != (int)blockIdx.x) __syncthreads();
        {
            TabRegs ptb;
            tables_fetch(p.io, c0, nlive, ptb);
            const int padj = p.adj[tid < S * Wn ? tid : 0];
            tables_put(tb, p.io, ptb);
            if (tid < 32) adj_l[tid] = padj;
        }
        const float gr = live ? p.g_reward.p[(int64_t)(c0 + x) * p.g_reward.scn_stride] : 0.f;
        f32x16 dz1[RT];   // pre-activation gradient of the first hidden layer, period t + 1 (this wavefront's rows)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz1[i][r] = 0.f;

        for (int t = p.T - 1; t >= 0; --t) {
            const bool has_next = t + 1 < p.T;
            // ---- tiles of period t (parked in registers) and the last hidden layer's activations (for ELU' in stage C')
            float4 pst[2], pdm[1], pod[1], pzt[1];
            tile_fetch<2>(p.states + t * p.ps_state, ldb, c0, p.F, pst);
            tile_fetch<1>(p.demand + t * p.ps_demand, p.ld_demand, c0, S, pdm);
            tile_fetch<1>(p.orders + t * p.ps_orders, ldb, c0, n_ord, pod);
            tile_fetch<1>(p.logits + t * p.ps_logits, ldb, c0, p.n_out, pzt);
            float act[RT][16];
            load_act<RT>(p.hidden[p.n_hidden - 1] + t * p.ps_hidden + c0, wave, ldb, H, act);
            // ---- A': first layer's input gradient of period t + 1, contracted over this wavefront's rows of dZ1 from the registers
            if (has_next) {
                f32x16 g2[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) g2[mt][r] = 0.f;
#pragma unroll
                    for (int i = 0; i < RT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) g2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aIn[mt][i][r], dz1[i][r], g2[mt], 0, 0, 0);
                }
                ksplit_publish<2>(red, g2);
                __syncthreads();
                if (wave == 0) {
                    ksplit_collect<2>(red, g2);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gs[(32 * mt + crow(r, h)) * LDT + li] = g2[mt][r];
                }
                __syncthreads();
                // gs <- G + the env part the previous iteration left in gi
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (tid >> 3) + 32 * i;
                    if (r < p.F) {
                        float4* cell = reinterpret_cast<float4*>(gs + r * LDT + (tid & 7) * 4);
                        const float4 e = *reinterpret_cast<const float4*>(gi + r * LDT + (tid & 7) * 4);
                        float4 g = *cell;
                        g.x += e.x; g.y += e.y; g.z += e.z; g.w += e.w;
                        *cell = g;
                    }
                }
            } else {
                for (int i = tid; i < 64 * LDT; i += kThreads) gs[i] = 0.f;
            }
            tile_put<2>(st, p.F, pst);
            tile_put<1>(dm, S, pdm);
            tile_put<1>(od, n_ord, pod);
            tile_put<1>(zt, p.n_out, pzt);
            for (int i = tid; i < 32 * LDT; i += kThreads) dz[i] = 0.f;
            __syncthreads();

            // ---- B': env adjoint + head adjoint (period_tail.hip, stage B')
            const NicEnvStepIO io = block_io(p.io, nlive, st, dm, od, tb);
            float* gs_wh = gs + S * d.store_slots * LDT;
            float* gi_wh = gi + S * d.store_slots * LDT;
            float* go_wh = go + S * Wn * LDT;
            for (int wc = 0; wc < Wn; wc += kChunk) {
                for (int i = 0; i < kChunk && wc + i < Wn; ++i) {
                    const float v = live ? nic::env_ship_partial(io, wc + i, x, q) : 0.f;
                    if (active) part[i][q][x] = v;
                }
                nic::lds_barrier();
                for (int i = q; i < kChunk && wc + i < Wn; i += nic::kQuad) {
                    const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
                    const float v = live ? nic::env_bwd_warehouse<MAXW>(io, gs_wh, gr, 0.f, wc + i, shipped, gi_wh, go_wh, x) : 0.f;
                    if (active) gwa[wc + i][x] = v;
                }
                nic::lds_barrier();
            }
            if (live) {
                const nic::IoAccess ac{io, x, nullptr, nullptr, gs, gs_wh, gi, gi_wh, go, go_wh};
                for (int s_ = q; s_ < S; s_ += nic::kQuad) nic::env_bwd_store_t<MAXW>(ac, gr, [&](int w) { return gwa[w][x]; }, s_);
            }
            nic::lds_barrier();
            for (int w = 0; w < Wn; ++w) {
                if (w > 0) nic::lds_barrier();
                nic::HeadLane<MAXSQ> L;
                int nc;
                const float mq = nic::head_quad_load<MAXSQ, true>(L, zt, go, adj_l, S, Wn, LDT, bb, w, q, nc, nullptr);
                if (active) xm[q][x] = mq;
                const float stock = io.wh_inv[w * Ww * LDT + bb];
                nic::lds_barrier();
                const float m = nic::head_quad_max(xm[0][x], xm[1][x], xm[2][x], xm[3][x], p.trans);
                const float dq = nic::head_quad_exp<MAXSQ>(L, m);
                if (active) xd[q][x] = dq;
                nic::lds_barrier();
                const float denom = nic::head_quad_denom(xd[0][x], xd[1][x], xd[2][x], xd[3][x], m, p.trans);
                float tq, sq;
                nic::head_quad_bwd_dots<MAXSQ>(L, denom, stock, tq, sq);
                if (active) {
                    xt[q][x] = tq;
                    xs[q][x] = sq;
                }
                nic::lds_barrier();
                if (live) {
                    const float dot = nic::combine4(xt[0][x], xt[1][x], xt[2][x], xt[3][x]);
                    nic::head_quad_bwd_store<MAXSQ>(L, dot, stock, dz, S, Wn, LDT, x, w, q, nullptr);
                    if (q == (w & 3)) {
                        gi_wh[w * Ww * LDT + x] += nic::combine4(xs[0][x], xs[1][x], xs[2][x], xs[3][x]);
                        nic::head_wh_order_bwd(zt, p.ub, go_wh, dz, S, Wn, LDT, x, w, -1);
                    }
                }
            }
            __syncthreads();   // dz complete; every scratch tile in buf0 is dead
            tile_store(dz, p.dZ_out + t * p.ps_dzout, ldb, c0, p.n_out, nlive);

            // ---- C': input gradient of the logits layer for this wavefront's rows, ELU' from the activation history
            f32x16 acc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aO[i][s], dz[(2 * s + h) * LDT + li], acc[i], 0, 0, 0);
            }
            {
                const int l = p.n_hidden - 1;
                dgrad_epilogue<RT>(acc, act, wave, l >= 1 ? buf0 : nullptr, p.dZ[l] + t * p.ps_dz + c0, ldb, n_store, H);
            }
            __syncthreads();
            // ---- hidden layers' input gradients, last to first; the first layer's stays in registers for the next iteration
            float* hin = buf0;
            float* hout = buf1;
            for (int l = p.n_hidden - 1; l >= 1; --l) {
                load_act<RT>(p.hidden[l - 1] + t * p.ps_hidden + c0, wave, ldb, H, act);
                stream_layer<RT, PF>(p.WpT[l], H, wave, hin, acc);
                dgrad_epilogue<RT>(acc, act, wave, l - 1 >= 1 ? hout : nullptr, p.dZ[l - 1] + t * p.ps_dz + c0, ldb, n_store, H);
                __syncthreads();
                float* tmp = hin;
                hin = hout;
                hout = tmp;
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) dz1[i] = acc[i];
        }
    }
}

int wide_max_slots(const NicEnvDims& d) { return d.store_slots > d.warehouse_slots ? d.store_slots : d.warehouse_slots; }

int wide_shapes_ok(const NicEnvDims& d, int n_out, int H, int n_hidden) {
    const int F = d.n_stores * d.store_slots + d.n_warehouses * d.warehouse_slots;
    return d.n_scenarios > 0 && d.ldb >= d.n_scenarios && d.ldb % 64 == 0 && d.n_stores >= 1 && d.n_stores <= 16 &&
           d.n_warehouses >= 1 && d.n_warehouses <= kMaxWh && d.n_echelons == 0 && d.store_slots >= 2 && d.warehouse_slots >= 2 &&
           wide_max_slots(d) <= 4 && n_out == (d.n_stores + 1) * d.n_warehouses && n_out <= 32 && F + 1 <= kStateRows && H == 512 &&
           n_hidden >= 2 && n_hidden <= kMaxHidden && (int64_t)H * d.ldb < (1ll << 29);
}

size_t wide_fwd_lds_bytes(int H, int n_hidden) {
    return sizeof(float) * ((size_t)2 * H * NB + 2 * kStateRows * LDT + kTabRows * LDT + 32 + (size_t)(n_hidden - 1) * H);
}

}  // namespace

extern "C" {

int nic_wide_rollout_ok(const NicEnvDims* dims, int32_t n_out, int32_t H, int32_t n_hidden) {
    return dims != nullptr && wide_shapes_ok(*dims, n_out, H, n_hidden);
}

int nic_wide_rollout_fwd(const NicWideRollout* w, void* stream) {
    NIC_REQUIRE(w != nullptr, "nic_wide_rollout_fwd: null descriptor");
    const NicEnvDims& d = w->io.dims;
    NIC_REQUIRE(wide_shapes_ok(d, w->n_out, w->H, w->n_hidden), "nic_wide_rollout_fwd: shapes outside the kernel's range (nic_wide_rollout_ok)");
    NIC_REQUIRE(w->T >= 1 && w->adjacency && w->demand && w->states && w->orders && w->logits && w->rewards && w->Wt_in && w->Wq_out,
                "nic_wide_rollout_fwd: null buffer");
    NIC_REQUIRE(w->io.underage.p && w->io.holding.p && w->io.lead_times.p && w->io.wh_holding.p && w->io.wh_lead_times.p,
                "nic_wide_rollout_fwd: null static table");
    for (int l = 1; l < w->n_hidden; ++l) NIC_REQUIRE(w->Wp_hidden[l] && w->b_hidden[l], "nic_wide_rollout_fwd: hidden layer %d missing", l);
    NIC_REQUIRE(w->ld_demand % 4 == 0 && w->ps_demand % 4 == 0 && w->ps_state % 4 == 0 && w->ps_orders % 4 == 0 && w->ldwt_in >= w->H,
                "nic_wide_rollout_fwd: strides must keep rows 16-byte aligned");
    NIC_REQUIRE(((reinterpret_cast<uintptr_t>(w->demand) | reinterpret_cast<uintptr_t>(w->states) | reinterpret_cast<uintptr_t>(w->orders)) & 15) == 0,
                "nic_wide_rollout_fwd: buffers must be 16-byte aligned");
    WideParams p{};
    p.io = w->io;
    p.adj = w->adjacency;
    p.ub = w->upper_bound;
    p.trans = w->transshipment;
    p.F = d.n_stores * d.store_slots + d.n_warehouses * d.warehouse_slots;
    p.T = w->T;
    p.H = w->H;
    p.n_hidden = w->n_hidden;
    p.n_out = w->n_out;
    p.n_blocks = nic::ceil_div(d.n_scenarios, NB);
    p.demand = w->demand;
    p.ps_demand = w->ps_demand;
    p.ld_demand = w->ld_demand;
    p.states = w->states;
    p.orders = w->orders;
    p.logits = w->logits;
    p.rewards = w->rewards;
    p.ps_state = w->ps_state;
    p.ps_orders = w->ps_orders;
    p.ps_logits = w->ps_logits;
    p.ps_hidden = w->ps_hidden;
    for (int l = 0; l < kMaxHidden; ++l) {
        p.hidden[l] = w->hidden[l];
        p.Wp[l] = reinterpret_cast<const float4*>(w->Wp_hidden[l]);
        p.bh[l] = w->b_hidden[l];
    }
    p.Wt_in = w->Wt_in;
    p.ldwt = w->ldwt_in;
    p.Wq = w->Wq_out;
    p.b_out = w->b_out;
    const int cus = nic::cu_count();
    const dim3 grid(p.n_blocks < cus ? p.n_blocks : cus), block(kThreads);
    const size_t lds = wide_fwd_lds_bytes(p.H, p.n_hidden);
    hipStream_t s = nic::as_stream(stream);
    const int ks = p.F + 1 <= 20 ? 10 : 26;
    nic::note_kernelf("wide_fwd_kernel<4,%d,4>", ks);
#define NIC_L(KS_)                                                                                                                   \
    do {                                                                                                                             \
        NIC_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_fwd_kernel<4, KS_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds) == hipSuccess,                                                                     \
                    "nic_wide_rollout_fwd: %zu bytes of LDS refused", lds);                                                          \
        hipLaunchKernelGGL((wide_fwd_kernel<4, KS_, 4>), grid, block, lds, s, p);                                                     \
    } while (0)
    if (ks == 10) NIC_L(10);
    else NIC_L(26);
#undef NIC_L
    return nic::check_launch("nic_wide_rollout_fwd");
}


int nic_wide_rollout_bwd(const NicWideRollout* w, NicTable2 g_reward, float* const* dZ_hidden, int64_t ps_dz, float* dZ_out,
                         int64_t ps_dzout, const float* const* WpT_hidden, const float* Wq_in, const float* Wo_t, void* stream) {
    NIC_REQUIRE(w != nullptr && dZ_hidden != nullptr && WpT_hidden != nullptr, "nic_wide_rollout_bwd: null argument");
    const NicEnvDims& d = w->io.dims;
    NIC_REQUIRE(wide_shapes_ok(d, w->n_out, w->H, w->n_hidden), "nic_wide_rollout_bwd: shapes outside the kernel's range (nic_wide_rollout_ok)");
    NIC_REQUIRE(w->T >= 1 && w->adjacency && w->demand && w->states && w->orders && w->logits && g_reward.p && dZ_out && Wq_in && Wo_t,
                "nic_wide_rollout_bwd: null buffer");
    for (int l = 0; l < w->n_hidden; ++l)
        NIC_REQUIRE(w->hidden[l] && dZ_hidden[l] && (l == 0 || WpT_hidden[l]), "nic_wide_rollout_bwd: history / weights of hidden layer %d missing", l);
    NIC_REQUIRE(ps_dz % 4 == 0 && ps_dzout % 4 == 0, "nic_wide_rollout_bwd: period strides must keep rows 16-byte aligned");
    WideBwdParams p{};
    p.io = w->io;
    p.adj = w->adjacency;
    p.ub = w->upper_bound;
    p.trans = w->transshipment;
    p.F = d.n_stores * d.store_slots + d.n_warehouses * d.warehouse_slots;
    p.T = w->T;
    p.H = w->H;
    p.n_hidden = w->n_hidden;
    p.n_out = w->n_out;
    p.n_blocks = nic::ceil_div(d.n_scenarios, NB);
    p.demand = w->demand;
    p.ps_demand = w->ps_demand;
    p.ld_demand = w->ld_demand;
    p.states = w->states;
    p.orders = w->orders;
    p.logits = w->logits;
    p.ps_state = w->ps_state;
    p.ps_orders = w->ps_orders;
    p.ps_logits = w->ps_logits;
    p.ps_hidden = w->ps_hidden;
    p.g_reward = g_reward;
    p.dZ_out = dZ_out;
    p.ps_dz = ps_dz;
    p.ps_dzout = ps_dzout;
    for (int l = 0; l < kMaxHidden; ++l) {
        p.hidden[l] = l < w->n_hidden ? w->hidden[l] : nullptr;
        p.dZ[l] = l < w->n_hidden ? dZ_hidden[l] : nullptr;
        p.WpT[l] = (l >= 1 && l < w->n_hidden) ? reinterpret_cast<const float4*>(WpT_hidden[l]) : nullptr;
    }
    p.Wq_in = Wq_in;
    p.Wo_t = Wo_t;
    const int cus = nic::cu_count();
    const dim3 grid(p.n_blocks < cus ? p.n_blocks : cus), block(kThreads);
    const size_t lds = sizeof(float) * ((size_t)2 * p.H * NB + kStateRows * LDT + kTabRows * LDT + 32);
    hipStream_t s = nic::as_stream(stream);
    const int ns = (p.n_out + 1) / 2;
    nic::note_kernelf("wide_bwd_kernel<4,%d,4>", ns <= 4 ? 4 : (ns <= 9 ? 9 : 16));
#define NIC_L(NS_)                                                                                                                   \
    do {                                                                                                                             \
        NIC_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_bwd_kernel<4, NS_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds) == hipSuccess,                                                                     \
                    "nic_wide_rollout_bwd: %zu bytes of LDS refused", lds);                                                          \
        hipLaunchKernelGGL((wide_bwd_kernel<4, NS_, 4>), grid, block, lds, s, p);                                                     \
    } while (0)
    if (ns <= 4) NIC_L(4);
    else if (ns <= 9) NIC_L(9);
    else NIC_L(16);
#undef NIC_L
    return nic::check_launch("nic_wide_rollout_bwd");
}

}  // extern "C"
