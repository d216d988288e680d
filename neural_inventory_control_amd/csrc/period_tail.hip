// The per-period "tail" of the vanilla_warehouse rollout in ONE launch per direction (round 5).
//
// Between two periods' hidden-layer GEMMs the per-period route used to run three dependent launches forward
//     logits GEMM (n_out x K)  ->  softmax head + env step  ->  next period's first layer (N1 x (F + 1), ELU)
// and three backward
//     first layer's input gradient of period t+1 (F x N1, accumulated into the state gradient)  ->  env adjoint + head adjoint of
//     period t  ->  logits layer's backward (input gradient with ELU' + weight gradient)
// every stage column-local (a scenario only ever reads its own column) and none of them large enough to fill the chip at the
// batch an 8-GPU shard or the reference's shipped YAML runs (8,192 / 1,024 scenarios: 9-16 us per launch whatever the batch).
// Here a workgroup of four wavefronts owns a BLOCK of 32 scenarios (the N of v_mfma_f32_32x32x2_f32) through all three stages:
//   forward   A  logits tile [n_out <= 32][32] = W_out H_last: the four wavefronts split the contraction and add their
//                accumulators through LDS once - gemm_wx_stream_kernel<1, 4>'s order, bit for bit;
//             B  head + env step on LDS tiles (state, demand, logits, orders [rows][32 scenarios]) with the NIC_HD bodies of
//                policy_heads_body.h / env_step_body.h, unchanged: they are handed a NicEnvStepIO whose pointers are the LDS
//                tiles (tile stride LDT) and whose static tables are offset to the block's first scenario - same arithmetic and
//                same Sum4 orders as head_env_fwd_kernel, bit for bit;
//             C  next period's first layer from the state tile in LDS (+ the row of ones that carries the bias): every wavefront
//                a quarter of the 32-row output blocks, thin_in_fwd_kernel's contraction order and ELU, bit for bit.
//   backward  A' G = Wt_in[F][N1] dZ1(t+1) for the block (two 32-row tiles, contraction split over the wavefronts as in
//                gemm_wx_stream_kernel<1, 4, EPI_DGRAD>) + the env part of the state gradient the previous launch left in HBM;
//             B' env adjoint + head adjoint on LDS tiles (head_env_bwd_kernel's bodies);
//             C' logits layer: dH = ELU'(H) * W_out^T dZ (thin_bwd_kernel's chain, bit for bit) and the weight-gradient partial
//                sum of the block, kept in registers over the workgroup's blocks and added to the workgroup's slab slot once.
// HBM traffic per scenario-period: forward 4 (K + N1) + the env step's bytes (H_last read, next H_first written); backward
// 4 (N1 + 2 K) + the env adjoint's bytes.  What the fusion removes is launches and their dependent start-up, not bytes:
// MFMA-time floors per block are 4.5 us forward / 6.1 us backward (17 -> 32 and 51 -> 64 row padding included).
// Built with -ffp-contract=off (head / env arithmetic rounds like the reference's separate aten ops); MFMA chains are fma by
// construction.  Reference: neural_networks.py:393-426 (+ :140-166), environment.py:110-299, trainer.py:190-213.
#include "tail_pieces.h"

namespace {

struct TailParams {
    NicEnvStepIO io;        // period t, global buffers (dims.ldb = the scenario stride of every [rows][ldb] block below)
    const int32_t* adj;     // [Wn][S]
    float ub;
    int trans;
    int F;                  // state rows S Ws + Wn Ww (store rows then warehouse rows, one contiguous block)
    // logits layer
    const float* W;         // [n_out][ldw]
    int64_t ldw;
    const float* bias;      // [n_out] or null
    int n_out, K;           // K = width of the last hidden layer
    const float* H;         // [K][ldb]   last hidden activation of period t
    float* Z;               // [n_out][ldb] logits of period t      (forward: written; backward: read)
    // first layer (transposed, bias as row F)
    const float* Wt;        // [F + 1][ldwt]
    int64_t ldwt;
    int N1;
    // forward outputs
    float* state_out;       // [F][ldb]   state of period t+1
    float* reward;          // [ldb]
    float* Y;               // [N1][ldb]  first hidden activation of period t+1 (null: last period)
    // backward
    const float* dZ1;       // [N1][ldb]  first layer's pre-activation gradient of period t+1 (null: last period)
    const float* g_next;    // [F][ldb]   env part of d loss / d state(t+1) (read when dZ1 != null)
    NicTable2 g_reward;
    float* g_out;           // [F][ldb]   env part of d loss / d state(t)
    float* dH;              // [K][ldb]   gradient of the last hidden layer's pre-activation
    float* slab;            // [n_slots][n_out][lds]
    int64_t lds_;
    int first;              // first launch of a backward sweep: the slab slot is written, not added to
    int n_blocks;
};

// LDS carve-up shared by both directions (float offsets)
struct Lds {
    static constexpr int zt = 0;                          // logits          [32][LDT]
    static constexpr int st = zt + 32 * LDT;              // state(t)        [52][LDT]
    static constexpr int dm = st + kStateRows * LDT;      // demand          [SMAX][LDT] (SMAX <= 64)
};

// ================================================================================================================================
// forward
// ================================================================================================================================
template <int MAXW, int MAXSQ, int KS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 4))) void tail_fwd_kernel(TailParams p) {
    constexpr int SMAX = 4 * MAXSQ;
    constexpr int o_tb = Lds::dm + SMAX * LDT;            // static tables                       [68][LDT]
    constexpr int o_adj = o_tb + kTabRows * LDT;          // adjacency (int)                     [Wn][S] <= 32
    constexpr int o_sn = o_adj + 32;                      // state(t+1) + ones row + zero rows  [52][LDT]
    constexpr int o_od = o_sn + kStateRows * LDT;         // orders                              [32][LDT]
    constexpr int o_ex = o_od + 32 * LDT;                 // exchange arrays
    constexpr int o_xm = o_ex, o_xd = o_xm + 4 * NB, o_xn = o_xd + 4 * NB, o_rq = o_xn + 4 * NB, o_cw = o_rq + 4 * NB,
                  o_part = o_cw + kMaxWh * NB, o_end = o_part + kChunk * 4 * NB;
    constexpr int RED = 3 * 16 * 64;                      // K-split reduction buffer, aliases [o_sn, ...) (dead before phase B)
    constexpr int TOTAL = (o_end - o_sn >= RED) ? o_end : o_sn + RED;
    __shared__ __attribute__((aligned(16))) float lds[TOTAL];
    float* zt = lds + Lds::zt;
    float* st = lds + Lds::st;
    float* dm = lds + Lds::dm;
    float* tb = lds + o_tb;
    int* adj_l = reinterpret_cast<int*>(lds + o_adj);
    float* sn = lds + o_sn;
    float* od = lds + o_od;
    float* red = lds + o_sn;
    float (*xm)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xm);
    float (*xd)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xd);
    int (*xn)[NB] = reinterpret_cast<int (*)[NB]>(lds + o_xn);
    float (*rq)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_rq);
    float (*cw)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_cw);
    float (*part)[4][NB] = reinterpret_cast<float (*)[4][NB]>(lds + o_part);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    const NicEnvDims& d = p.io.dims;
    const int B = d.n_scenarios, S = d.n_stores, Wn = d.n_warehouses, Ww = d.warehouse_slots;
    const int64_t ldb = d.ldb;
    const int c0 = blockIdx.x * NB;
    const int nlive = min(NB, B - c0);
    const int ncols = (B + 3) / 4 * 4;   // columns the GEMM stages write (as nic_linear_fwd)

    TAIL_STAMP(0);
    // ---- state and demand of the period: requested now, parked in registers until the logits are done
    float4 pst[2], pdm[(SMAX + 31) / 32];
    TabRegs ptb;
    tile_fetch<2>(p.io.store_inv, ldb, c0, p.F, pst);
    tile_fetch<(SMAX + 31) / 32>(p.io.demand.p, p.io.demand.loc_stride, c0, S, pdm);
    tables_fetch(p.io, c0, nlive, ptb);
    const int padj = p.adj[tid < S * Wn ? tid : 0];

    // ---- A: logits
    {
        f32x16 acc[1];
        ksplit_contract<1>(p.W, p.ldw, p.n_out, p.H, ldb, p.K, c0, acc);
        TAIL_STAMP(1);
        float aux[16];
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = crow(r, h);
                aux[r] = (p.bias != nullptr && row < p.n_out) ? p.bias[row] : 0.f;
            }
        }
        ksplit_publish<1>(red, acc);
        __syncthreads();
        if (wave == 0) {
            ksplit_collect<1>(red, acc);
            const int col = c0 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = crow(r, h);
                const float y = acc[0][r] + aux[r];
                zt[row * LDT + li] = y;
                if (row < p.n_out && col < ncols) p.Z[(int64_t)row * ldb + col] = y;
            }
        }
        tile_put<2>(st, p.F, pst);
        tile_put<(SMAX + 31) / 32>(dm, S, pdm);
        tables_put(tb, p.io, ptb);
        if (tid < 32) adj_l[tid] = padj;
    }
    // first fragments of phase C: in flight while the head and the env step run
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.Wt, (int64_t)(p.F + 1) * p.ldwt);
    const int lw4 = (int)p.ldwt * 4;
    const int all_blocks = (p.N1 + 31) / 32, per_w = (all_blocks + 3) / 4;   // (a ragged last block: its rows >= N1 are never stored)
    const int nb_lo = wave * per_w, nb_hi = min(nb_lo + per_w, all_blocks);
    float a[KS];
    auto load_block = [&](int nb) {
        const int vw = h * lw4 + (nb * 32 + li) * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = ldf(rW, vw, 2 * s * lw4);
    };
    if (p.Y != nullptr && nb_lo < nb_hi) load_block(nb_lo);
    TAIL_STAMP(2);
    __syncthreads();   // logits, state and demand tiles published; the reduction buffer is dead
    TAIL_STAMP(3);

    // next-state tile: zeros (padding columns, rows past F + 1), ones in row F (the bias row of the first layer)
    for (int i = tid; i < kStateRows * LDT; i += kThreads) sn[i] = (i / LDT == p.F) ? 1.f : 0.f;

    // ---- B: head + env step (head_env_fwd_kernel's body on LDS tiles; wavefronts 0 and 1 hold the 32 x 4 quad lanes)
    const bool active = tid < 4 * NB;
    const int x = tid & (NB - 1), q = (tid >> 5) & 3;
    const bool live = active && x < nlive;
    const int bb = x < nlive ? x : nlive - 1;   // dead lanes shadow the last live scenario (loads only)
    const NicEnvStepIO io = block_io(p.io, nlive, st, dm, od, tb);
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
    TAIL_STAMP(4);
    float* sn_wh = sn + S * d.store_slots * LDT;
    {
        // this lane's stores s = q, q + 4, ... one at a time (env_fwd_store_t: the arithmetic of one store of env_fwd_stores, bit
        // for bit - tests/hostsim; the batching of env_fwd_stores hides HBM latency, which LDS tiles do not have, at 60 registers)
        const nic::IoAccess ac{io, x, sn, sn_wh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        float r = 0.f;
        if (live)
            for (int s_ = q; s_ < S; s_ += nic::kQuad) r += nic::env_fwd_store_t<MAXW>(ac, s_);
        if (active) rq[q][x] = r;
    }
    TAIL_STAMP(5);
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
        p.reward[c0 + x] = total + r_wh;
    }
    TAIL_STAMP(6);
    // orders and next state of the live scenarios -> HBM (the backward sweep and the next period's GEMMs read them)
    tile_store(od, const_cast<float*>(p.io.store_orders.p), ldb, c0, S * Wn + Wn, nlive);
    tile_store(sn, p.state_out, ldb, c0, p.F, nlive);
    TAIL_STAMP(7);
    if (p.Y == nullptr || nb_lo >= nb_hi) return;

    // ---- C: first layer of period t+1 (thin_in_fwd_kernel<KS, false>: K + 1 rows, the bias inside the contraction)
    float xs[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xs[s] = sn[(2 * s + h) * LDT + li];
    const int ld4 = (int)ldb * 4;
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(p.Y + c0, (int64_t)p.N1 * ldb - c0);   // stores to rows >= N1 are dropped
    const int vo = 4 * h * ld4 + li * 4;
    const bool col_live = c0 + li < ncols;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first fragments (and this wavefront's tile stores)
    // One wavefront per SIMD at the batches this kernel is for: the MFMA chain of block b + 1 (26 x 64 cycles on the matrix pipe) and
    // the ELUs of block b (16 x ~15 VALU instructions) are written interleaved, one ELU behind every MFMA, so that the two pipes
    // run side by side; thin_in_fwd_kernel gets the same overlap from four wavefronts per SIMD.  Same chains, same bits.
    f32x16 y, yn;
    auto chain = [&](int nb_reload, f32x16& acc, auto with_elu) {
        const int vw = h * lw4 + (nb_reload * 32 + li) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < (KS > 16 ? KS : 16); ++s) {
            if (s < KS) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xs[s], acc, 0, 0, 0);
                a[s] = ldf(rW, vw, 2 * s * lw4);
            }
            if (decltype(with_elu)::value && s < 16) y[s] = elu_f(y[s]);
        }
    };
    chain(nb_lo + 1 < nb_hi ? nb_lo + 1 : nb_lo, y, std::false_type{});
    for (int nb = nb_lo; nb < nb_hi; ++nb) {
        if (nb + 1 < nb_hi) {
            chain(nb + 2 < nb_hi ? nb + 2 : nb + 1, yn, std::true_type{});
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y[r] = elu_f(y[r]);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (col_live) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[r]), rY, vo, (nb * 32 + (r & 3) + 8 * (r >> 2)) * ld4, 0);
        }
        y = yn;
    }
    TAIL_STAMP(8);
}

// ================================================================================================================================
// backward
// ================================================================================================================================
constexpr int kXLD = NB + 4;   // wave-private H chunk [32 rows][32 scenarios] row stride

constexpr int kGroup = 4;      // blocks of a workgroup whose adjoints run before their logits-layer backward (see below)

// A workgroup walks its blocks (blockIdx.x, + gridDim.x, ...) in GROUPS of kGroup: first the adjoint stages A' and B' of every
// block of the group (the logits gradient tile of each block stays in LDS), then stage C' for the group with the weight-gradient
// accumulators (64 registers per lane) alive only there, then one read-modify-write of the workgroup's slab slot.  (With one
// loop over blocks the accumulators are alive across the env / head adjoints and the kernel needs ~340 registers.)
template <int MAXW, int MAXSQ, int NS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void tail_bwd_kernel(TailParams p) {
    constexpr int SMAX = 4 * MAXSQ;
    constexpr int o_dz = 0;                               // logits gradients of the group  [kGroup][32][LDT]
    constexpr int o_w = o_dz + kGroup * 32 * LDT;         // ---- working tiles of stages A' / B' (dead in stage C') ----
    constexpr int o_zt = o_w;                             // logits(t)                 [32][LDT]
    constexpr int o_st = o_zt + 32 * LDT;                 // state(t)                  [52][LDT]
    constexpr int o_dm = o_st + kStateRows * LDT;         // demand(t)                 [SMAX][LDT]
    constexpr int o_od = o_dm + SMAX * LDT;               // orders(t)                 [32][LDT]
    constexpr int o_gs = o_od + 32 * LDT;                 // d loss / d state(t+1)     [64][LDT] (two 32-row tiles of G)
    constexpr int o_gi = o_gs + 64 * LDT;                 // env part of d loss / d state(t)  [52][LDT]
    constexpr int o_go = o_gi + kStateRows * LDT;         // order gradients           [32][LDT]
    constexpr int o_ex = o_go + 32 * LDT;
    constexpr int o_xm = o_ex, o_xd = o_xm + 4 * NB, o_xt = o_xd + 4 * NB, o_xs = o_xt + 4 * NB, o_gwa = o_xs + 4 * NB,
                  o_part = o_gwa + kMaxWh * NB, o_tb = o_part + kChunk * 4 * NB, o_adj = o_tb + kTabRows * LDT, o_wend = o_adj + 32;
    constexpr int RED = 3 * 2 * 16 * 64;                  // K-split reduction buffer (two row tiles): aliases [o_gi, ...)
    constexpr int XT = 4 * 32 * kXLD;                     // four wave-private H chunks (stage C'): alias the working tiles
    constexpr int o_end0 = (o_gi + RED > o_wend) ? o_gi + RED : o_wend;
    constexpr int o_end = (o_w + XT > o_end0) ? o_w + XT : o_end0;
    __shared__ __attribute__((aligned(16))) float lds[o_end];
    float* zt = lds + o_zt;
    float* st = lds + o_st;
    float* dm = lds + o_dm;
    float* od = lds + o_od;
    float* gs = lds + o_gs;
    float* gi = lds + o_gi;
    float* go = lds + o_go;
    float* tb = lds + o_tb;
    int* adj_l = reinterpret_cast<int*>(lds + o_adj);
    float* red = lds + o_gi;
    float (*xm)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xm);
    float (*xd)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xd);
    float (*xt)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xt);
    float (*xs)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_xs);
    float (*gwa)[NB] = reinterpret_cast<float (*)[NB]>(lds + o_gwa);
    float (*part)[4][NB] = reinterpret_cast<float (*)[4][NB]>(lds + o_part);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h = lane >> 5;
    float* xw = lds + o_w + wave * 32 * kXLD;
    const NicEnvDims& d = p.io.dims;
    const int B = d.n_scenarios, S = d.n_stores, Wn = d.n_warehouses, Ww = d.warehouse_slots;
    const int64_t ldb = d.ldb;
    const int ldb32 = (int)ldb;
    const int n_ord = S * Wn + Wn;
    const bool active = tid < 4 * NB;
    const int x = tid & (NB - 1), q = (tid >> 5) & 3;
    constexpr int KC = 4;   // this wavefront's 32-row chunks of the last hidden layer: wave, wave + 4, ... (K / 32 <= 4 KC)
    const int n_chunks_all = (p.K + 31) / 32;   // (a ragged last chunk: rows >= K read as zeros, their stores are dropped)
    const __amdgpu_buffer_rsrc_t rWo = make_rsrc(p.W, (int64_t)p.n_out * p.ldw);   // rows >= n_out read as zeros
    const __amdgpu_buffer_rsrc_t rH = make_rsrc(p.H, (int64_t)p.K * p.io.dims.ldb);
    const __amdgpu_buffer_rsrc_t rdH = make_rsrc(p.dH, (int64_t)p.K * p.io.dims.ldb);
    const int stride = (int)gridDim.x;
    bool first_group = true;

    for (int g0 = blockIdx.x; g0 < p.n_blocks; g0 += kGroup * stride, first_group = false) {
        if (!first_group) __syncthreads();   // stage C' of the previous group is done with the H chunks that alias the working tiles
        int n_in_group = 0;
#pragma clang loop unroll(disable)
        for (int j = 0; j < kGroup; ++j) {
            const int blk = g0 + j * stride;
            if (blk >= p.n_blocks) break;
            ++n_in_group;
            float* dz = lds + o_dz + j * 32 * LDT;
            const int c0 = blk * NB;
            const int nlive = min(NB, B - c0);
            const bool live = active && x < nlive;
            const int bb = x < nlive ? x : nlive - 1;
            if (j > 0) __syncthreads();   // the previous block's working tiles are dead
            if (j == 0 && first_group) TAIL_STAMP(0);

            // ---- tiles of period t: state, demand, orders, logits; the env part of the next state's gradient
            float4 pst[2], pdm[(SMAX + 31) / 32], pod[1], pzt[1], pgs[2];
            TabRegs ptb;
            tables_fetch(p.io, c0, nlive, ptb);
            const int padj = p.adj[tid < S * Wn ? tid : 0];
            tile_fetch<2>(p.io.store_inv, ldb, c0, p.F, pst);
            tile_fetch<(SMAX + 31) / 32>(p.io.demand.p, p.io.demand.loc_stride, c0, S, pdm);
            tile_fetch<1>(p.io.store_orders.p, ldb, c0, n_ord, pod);
            tile_fetch<1>(p.Z, ldb, c0, p.n_out, pzt);
            if (p.dZ1 != nullptr) tile_fetch<2>(p.g_next, ldb, c0, p.F, pgs);
            else pgs[0] = pgs[1] = make_float4(0.f, 0.f, 0.f, 0.f);

            // ---- A': G = Wt_in[:F] dZ1(t+1), two 32-row tiles, + the env part (gemm_wx_stream_kernel<1, 4, EPI_DGRAD>, accumulate)
            if (p.dZ1 != nullptr) {
                f32x16 acc[2];
                ksplit_contract<2, 4>(p.Wt, p.ldwt, p.F, p.dZ1, ldb, p.N1, c0, acc);
                if (j == 0 && first_group) TAIL_STAMP(1);
                ksplit_publish<2>(red, acc);
                __syncthreads();
                if (wave == 0) {
                    ksplit_collect<2>(red, acc);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gs[(32 * i + crow(r, h)) * LDT + li] = acc[i][r];
                }
                __syncthreads();
                // gs <- G + env part (the stream kernel's y = acc; y += C), every thread the float4s it fetched
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (tid >> 3) + 32 * i;
                    if (r < p.F) {
                        float4* cell = reinterpret_cast<float4*>(gs + r * LDT + (tid & 7) * 4);
                        float4 g = *cell;
                        g.x += pgs[i].x; g.y += pgs[i].y; g.z += pgs[i].z; g.w += pgs[i].w;
                        *cell = g;
                    }
                }
            } else {
                tile_put<2>(gs, p.F, pgs);   // last period: no later state, zeros
            }
            tile_put<2>(st, p.F, pst);
            tile_put<(SMAX + 31) / 32>(dm, S, pdm);
            tile_put<1>(od, n_ord, pod);
            tile_put<1>(zt, p.n_out, pzt);
            tables_put(tb, p.io, ptb);
            if (tid < 32) adj_l[tid] = padj;
            for (int i = tid; i < 32 * LDT; i += kThreads) dz[i] = 0.f;
            if (j == 0 && first_group) TAIL_STAMP(2);
            __syncthreads();
            if (j == 0 && first_group) TAIL_STAMP(3);

            // ---- B': env adjoint + head adjoint (head_env_bwd_kernel's bodies on LDS tiles)
            const NicEnvStepIO io = block_io(p.io, nlive, st, dm, od, tb);
            const float gr = live ? p.g_reward.p[(int64_t)(c0 + x) * p.g_reward.scn_stride] : 0.f;
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
            if (live) {   // (one store at a time, as in the forward kernel: env_bwd_store_t = one store of env_bwd_stores)
                const nic::IoAccess ac{io, x, nullptr, nullptr, gs, gs_wh, gi, gi_wh, go, go_wh};
                for (int s_ = q; s_ < S; s_ += nic::kQuad) nic::env_bwd_store_t<MAXW>(ac, gr, [&](int w) { return gwa[w][x]; }, s_);
            }
            nic::lds_barrier();
            if (j == 0 && first_group) TAIL_STAMP(4);
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
            nic::lds_barrier();
            if (j == 0 && first_group) TAIL_STAMP(5);
            tile_store(gi, p.g_out, ldb, c0, p.F, nlive);
            if (j == 0 && first_group) TAIL_STAMP(6);
        }
        __syncthreads();   // every block's logits gradient is in LDS; the working tiles are dead (the H chunks take their place)
        if (first_group) TAIL_STAMP(7);

        // ---- C': logits layer backward of the group, this wavefront's row chunks of H (thin_bwd_kernel's arithmetic on
        // 32-scenario blocks); wave-private from here to the slab update: no workgroup barrier
        f32x16 wacc[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[c][r] = 0.f;
        float bias_acc = 0.f;
        const bool arow = li < 2 * NS;
        const int srow = lane >> 3, scol = (lane & 7) * 4;   // staging map of a [32][32] chunk: row (lane >> 3) + 8 u, columns 4 (lane & 7) ..
        // The row stride goes through an empty asm once per group: everything derived from it (64 row offsets of the dH stores, 36
        // fragment offsets) is then recomputed where it is used - a few SALU / VALU instructions - instead of being hoisted out of
        // the loops as ~100 live registers.
        int ldbo = ldb32;
        asm volatile("" : "+s"(ldbo));
        float4 xv[4];
        auto load_x = [&](int c0_, int ch) {   // (buffer addressing: one per-lane offset, the chunk's rows as scalar offsets)
            const int vx = (srow * ldbo + c0_ + scol) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                xv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rH, vx, (ch * 32 + 8 * u) * ldbo * 4, 0));
        };
        // W_out^T fragments of a chunk: A[i = k][kk = n], n = 2 s + h (L2-resident), fetched one chunk ahead like the H chunk
        float aW[NS];
        auto load_w = [&](int ch) {
            int ldwo = (int)p.ldw;
            asm volatile("" : "+s"(ldwo));
            const int vw_ = (h * ldwo + li) * 4;
#pragma unroll
            for (int s = 0; s < NS; ++s) aW[s] = ldf(rWo, vw_, (2 * s * ldwo + ch * 32) * 4);
        };
        if (wave < n_chunks_all) {
            load_x(g0 * NB, wave);
            load_w(wave);
        }
#pragma clang loop unroll(disable)
        for (int j = 0; j < n_in_group; ++j) {
            const int blk = g0 + j * stride;
            const float* dz = lds + o_dz + j * 32 * LDT;
            const int c0 = blk * NB;
            const int nlive = min(NB, B - c0);
            const int vdh = (4 * h * ldbo + c0 + li) * 4;
            if (wave == 0 && arow) {   // bias gradient: lane (i = n, h) sums its 16 scenarios of the block
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 av = *reinterpret_cast<const float4*>(dz + li * LDT + 8 * qq + 4 * h);
                    bias_acc += (av.x + av.y) + (av.z + av.w);
                }
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int ch = wave + 4 * c;
                if (ch >= n_chunks_all) break;
                __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise hoists every chunk's fragment reads to the top)
                // chunk: registers (fetched one chunk ahead) -> wave-private LDS tile, scenarios past the last live one zeroed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = xv[u];
                    v.x = scol + 0 < nlive ? v.x : 0.f;
                    v.y = scol + 1 < nlive ? v.y : 0.f;
                    v.z = scol + 2 < nlive ? v.z : 0.f;
                    v.w = scol + 3 < nlive ? v.w : 0.f;
                    *reinterpret_cast<float4*>(xw + (srow + 8 * u) * kXLD + scol) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // next chunk (of this block, or the first one of the group's next block): in flight under this chunk's MFMAs
                if (ch + 4 < n_chunks_all) load_x(c0, ch + 4);
                else if (j + 1 < n_in_group) load_x((blk + stride) * NB, wave);
                float aWc[NS];   // this chunk's fragments (the next chunk's are requested below)
#pragma unroll
                for (int s = 0; s < NS; ++s) aWc[s] = aW[s];
                // weight gradient: D[n][k] += sum_b dZ[n][b] H[k][b], 16 MFMA steps over the 32 scenarios
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 ay = arow ? *reinterpret_cast<const float4*>(dz + li * LDT + 8 * qq + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 bx = *reinterpret_cast<const float4*>(xw + li * kXLD + 8 * qq + 4 * h);
                    wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.x, bx.x, wacc[c], 0, 0, 0);
                    wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.y, bx.y, wacc[c], 0, 0, 0);
                    wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.z, bx.z, wacc[c], 0, 0, 0);
                    wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.w, bx.w, wacc[c], 0, 0, 0);
                }
                // input gradient of the chunk
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aWc[s], dz[(2 * s + h) * LDT + li], acc, 0, 0, 0);
                if (ch + 4 < n_chunks_all) load_w(ch + 4);
                else if (j + 1 < n_in_group) load_w(wave);
                if (li < nlive) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ku = (r & 3) + 8 * (r >> 2);
                        const float g = elu_grad_from_out(xw[(ku + 4 * h) * kXLD + li]);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r] * g), rdH, vdh, (ch * 32 + ku) * ldbo * 4, 0);
                    }
                }
            }
        }

        if (first_group) TAIL_STAMP(8);
        // ---- slab slot of this workgroup: D layout lane (j = k, h), register r <-> output row n = crow(r, h).  Buffer addressing
        // (rows >= n_out lie beyond the descriptor: loads return 0, stores are dropped) with the row stride laundered like ldbo.
        const bool overwrite = p.first && first_group;
        int ldso = (int)p.lds_;
        asm volatile("" : "+s"(ldso));
        const __amdgpu_buffer_rsrc_t rS = make_rsrc(p.slab + (int64_t)blockIdx.x * p.n_out * p.lds_, (int64_t)p.n_out * p.lds_);
        const int vs_ = (4 * h * ldso + li) * 4;
        // (all loads of the slot first, then the sums and stores: one L2 round trip per workgroup instead of one per chunk; columns
        // of a ragged last chunk - the slab row ends at K + 1 - are sent beyond the descriptor)
        auto vs_k = [&](int ch) { return ch * 32 + li < p.K ? vs_ : 0x7fffff00; };
        f32x16 old[KC];
        if (!overwrite) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int ch = wave + 4 * c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ku = (r & 3) + 8 * (r >> 2);
                    old[c][r] = ch < n_chunks_all ? ldf(rS, vs_k(ch), (ku * ldso + ch * 32) * 4) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) old[c][r] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int ch = wave + 4 * c;
            if (ch >= n_chunks_all) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ku = (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old[c][r] + wacc[c][r]), rS, vs_k(ch), (ku * ldso + ch * 32) * 4, 0);
            }
        }
        if (wave == 0) {
            const float total = bias_acc + __shfl_xor(bias_acc, 32);
            if (h == 0 && li < p.n_out) {
                float* cell = p.slab + ((int64_t)blockIdx.x * p.n_out + li) * p.lds_ + p.K;
                *cell = (overwrite ? 0.f : *cell) + total;
            }
        }
        if (first_group) TAIL_STAMP(9);
    }
}

int max_slots(const NicEnvDims& d) { return d.store_slots > d.warehouse_slots ? d.store_slots : d.warehouse_slots; }

int tail_shapes_ok(const NicEnvDims& d, int n_out, int K, int N1) {
    const int F = d.n_stores * d.store_slots + d.n_warehouses * d.warehouse_slots;
    return d.n_scenarios > 0 && d.ldb >= d.n_scenarios && d.ldb % 64 == 0 && d.n_stores >= 1 && d.n_stores <= 16 &&
           d.n_warehouses >= 1 && d.n_warehouses <= kMaxWh && d.n_echelons == 0 && d.store_slots >= 2 && d.warehouse_slots >= 2 &&
           max_slots(d) <= 8 && n_out == (d.n_stores + 1) * d.n_warehouses && n_out <= 32 && F + 1 <= kStateRows && K >= 1 &&
           K <= 512 && N1 >= 1 && (int64_t)N1 * d.ldb < (1ll << 29) && (int64_t)K * d.ldb < (1ll << 29);
}

int validate(const NicPeriodTail* t, const char* who) {
    NIC_REQUIRE(t != nullptr, "%s: null descriptor", who);
    const NicEnvStepIO& io = t->io;
    const NicEnvDims& d = io.dims;
    NIC_REQUIRE(tail_shapes_ok(d, t->n_out, t->K, t->N1), "%s: shapes outside the fused tail's range (nic_period_tail_ok)", who);
    NIC_REQUIRE(t->adjacency && t->W_out && t->Wt_in && t->ldw_out >= t->K && t->ldwt_in >= t->N1 && t->ldw_out % 4 == 0 &&
                    t->ldwt_in % 4 == 0,
                "%s: weights missing or rows not padded to 16 bytes", who);
    NIC_REQUIRE(io.store_inv && io.wh_inv && io.demand.p && io.store_orders.p && io.wh_orders.p && io.underage.p && io.holding.p &&
                    io.lead_times.p && io.wh_holding.p && io.wh_lead_times.p,
                "%s: null buffer in io", who);
    NIC_REQUIRE(io.wh_inv == io.store_inv + (int64_t)d.n_stores * d.store_slots * d.ldb,
                "%s: the state must be one [S Ws + Wn Ww][ldb] block", who);
    NIC_REQUIRE(io.store_orders.scn_stride == 1 && io.store_orders.sup_stride == d.ldb &&
                    io.store_orders.loc_stride == (int64_t)d.n_warehouses * d.ldb && io.wh_orders.scn_stride == 1 &&
                    io.wh_orders.loc_stride == d.ldb &&
                    io.wh_orders.p == io.store_orders.p + (int64_t)d.n_stores * d.n_warehouses * d.ldb,
                "%s: the orders must be one dense [S Wn + Wn][ldb] block", who);
    NIC_REQUIRE(io.demand.scn_stride == 1 && io.demand.loc_stride % 4 == 0, "%s: demand must be a [S][ld] block", who);
    NIC_REQUIRE(((reinterpret_cast<uintptr_t>(io.store_inv) | reinterpret_cast<uintptr_t>(io.demand.p) |
                  reinterpret_cast<uintptr_t>(io.store_orders.p) | reinterpret_cast<uintptr_t>(t->W_out) |
                  reinterpret_cast<uintptr_t>(t->Wt_in)) & 15) == 0,
                "%s: buffers must be 16-byte aligned", who);
    return 0;
}

TailParams base_params(const NicPeriodTail* t) {
    TailParams p{};
    p.io = t->io;
    p.adj = t->adjacency;
    p.ub = t->upper_bound;
    p.trans = t->transshipment;
    const NicEnvDims& d = t->io.dims;
    p.F = d.n_stores * d.store_slots + d.n_warehouses * d.warehouse_slots;
    p.W = t->W_out;
    p.ldw = t->ldw_out;
    p.bias = t->b_out;
    p.n_out = t->n_out;
    p.K = t->K;
    p.Wt = t->Wt_in;
    p.ldwt = t->ldwt_in;
    p.N1 = t->N1;
    p.n_blocks = nic::ceil_div(d.n_scenarios, NB);
    return p;
}

}  // namespace

#ifdef NIC_TUNING_BUILD
extern "C" int nic_tuning_set_tail_stamps(void* buf) {   // buf: device memory, [4][16] u64 (or null)
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

int nic_period_tail_ok(const NicEnvDims* dims, int32_t n_out, int32_t K, int32_t N1) {
    return dims != nullptr && tail_shapes_ok(*dims, n_out, K, N1);
}

int nic_period_tail_bwd_slots(int32_t n_scenarios) {
    const int blocks = nic::ceil_div(n_scenarios > 0 ? n_scenarios : 1, NB);
    const int cap = nic::cu_count();   // (the backward kernel holds ~370 registers: one workgroup per CU; more blocks: groups)
    return blocks < cap ? blocks : cap;
}

#define NIC_TAIL_DISPATCH(LAUNCH, X)                       \
    do {                                                   \
        if (max_slots(d) <= 4) LAUNCH(4, 4, X);            \
        else LAUNCH(8, 4, X);                              \
    } while (0)

int nic_period_tail_fwd(const NicPeriodTail* t, const float* H_last, float* Z, float* state_out, float* reward, float* H_first_next,
                        void* stream) {
    if (int e = validate(t, "nic_period_tail_fwd")) return e;
    NIC_REQUIRE(H_last && Z && state_out && reward, "nic_period_tail_fwd: null buffer");
    NIC_REQUIRE(((reinterpret_cast<uintptr_t>(H_last) | reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(state_out)) & 15) == 0,
                "nic_period_tail_fwd: buffers must be 16-byte aligned");
    TailParams p = base_params(t);
    const NicEnvDims& d = t->io.dims;
    p.H = H_last;
    p.Z = Z;
    p.state_out = state_out;
    p.reward = reward;
    p.Y = H_first_next;
    const int ks = p.F + 1 <= 20 ? 10 : (p.F + 1 <= 36 ? 18 : 26);
    const dim3 grid(p.n_blocks), block(kThreads);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernelf("tail_fwd_kernel<%d,4,%d>", max_slots(d) <= 4 ? 4 : 8, ks);
#define NIC_L(MW, SQ, KS_) hipLaunchKernelGGL((tail_fwd_kernel<MW, SQ, KS_>), grid, block, 0, s, p)
    if (ks == 10) NIC_TAIL_DISPATCH(NIC_L, 10);
    else if (ks == 18) NIC_TAIL_DISPATCH(NIC_L, 18);
    else NIC_TAIL_DISPATCH(NIC_L, 26);
#undef NIC_L
    return nic::check_launch("nic_period_tail_fwd");
}

int nic_period_tail_bwd(const NicPeriodTail* t, const float* Z, const float* H_last, const float* dZ_first_next,
                        const float* g_state_next, NicTable2 g_reward, float* g_state_out, float* dH_last, float* slab, int64_t lds,
                        int32_t n_slots, int32_t first, void* stream) {
    if (int e = validate(t, "nic_period_tail_bwd")) return e;
    NIC_REQUIRE(Z && H_last && g_reward.p && g_state_out && dH_last && slab, "nic_period_tail_bwd: null buffer");
    NIC_REQUIRE((dZ_first_next == nullptr) == (g_state_next == nullptr),
                "nic_period_tail_bwd: dZ_first_next and g_state_next go together (both null for the last period)");
    NIC_REQUIRE(lds >= t->K + 1 && n_slots >= 1, "nic_period_tail_bwd: bad slab geometry");
    NIC_REQUIRE(((reinterpret_cast<uintptr_t>(H_last) | reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(g_state_out) |
                  reinterpret_cast<uintptr_t>(g_state_next) | reinterpret_cast<uintptr_t>(dZ_first_next)) & 15) == 0,
                "nic_period_tail_bwd: buffers must be 16-byte aligned");
    TailParams p = base_params(t);
    const NicEnvDims& d = t->io.dims;
    p.Z = const_cast<float*>(Z);
    p.H = H_last;
    p.dZ1 = dZ_first_next;
    p.g_next = g_state_next;
    p.g_reward = g_reward;
    p.g_out = g_state_out;
    p.dH = dH_last;
    p.slab = slab;
    p.lds_ = lds;
    p.first = first;
    const int want = nic_period_tail_bwd_slots(d.n_scenarios);
    const int wgs = n_slots < want ? n_slots : want;
    const int ns = (p.n_out + 1) / 2;
    const dim3 grid(wgs), block(kThreads);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernelf("tail_bwd_kernel<%d,4,%d>", max_slots(d) <= 4 ? 4 : 8, ns <= 4 ? 4 : (ns <= 9 ? 9 : 16));
#define NIC_L(MW, SQ, NS_) hipLaunchKernelGGL((tail_bwd_kernel<MW, SQ, NS_>), grid, block, 0, s, p)
    if (ns <= 4) NIC_TAIL_DISPATCH(NIC_L, 4);
    else if (ns <= 9) NIC_TAIL_DISPATCH(NIC_L, 9);
    else NIC_TAIL_DISPATCH(NIC_L, 16);
#undef NIC_L
    return nic::check_launch("nic_period_tail_bwd");
}

}  // extern "C"
