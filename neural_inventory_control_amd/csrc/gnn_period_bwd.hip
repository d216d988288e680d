// GNN policy: ONE launch per period for the BACKWARD of the whole policy (include/nic_rollout.h: nic_gnn_period_bwd; the adjoint of
// csrc/gnn_period.hip, i.e. of the reference's neural_networks.py:1105-1192 initial embeddings, :1229-1340 message passing,
// :1367-1392 output MLP).  Round 5's backward launches one `mlp3_bwd_hist` kernel per MLP and a segment sum behind each of them:
// every input gradient ([96][entities] rows) makes a round trip through HBM before the transposed gather adds it up, nine dependent
// launches per period.  Here a workgroup owns kSub blocks of 16 scenarios for all five MLPs.
//
// Layout.  The forward kernel's: an (entity, 16 scenarios) tile of a 32-row quantity is two accumulators of v_mfma_f32_16x16x4_f32,
// lane (n = lane & 15, g = lane >> 4) of accumulator rb holds rows 16 rb + 4 g + i of scenario n ("D layout") - which is the B operand
// of the next contraction, so the chain dz3 -> dh2 -> dz2 -> dh1 -> dz1 -> dx stays in registers with the TRANSPOSED weights as
// pre-packed A fragments in LDS (`ops.GnnPeriodBwdPack`).  The weight gradients dW += dz x^T contract over the block's 16 scenarios:
// A = dz^T (through a wave-private padded LDS tile: lane (m, q) takes row m, scenarios 4 q .. 4 q + 3 as one 16-byte read),
// B = x^T (the same 16-byte pieces straight from the forward's history rows), four MFMAs per 16 x 16 block of dW; the blocks stay in
// registers over all tiles a wavefront walks and the eight wavefronts add them through LDS in wave order into the workgroup's slab
// slot once per MLP (deterministic: no atomics).
//
// Stages (entities of a stage are dealt to the wavefronts round robin; tiles that cross a stage go through a per-workgroup scratch
// area in global memory in the lanes' own order - 2 KB contiguous per tile, L2-resident - so ANY graph size runs, unlike the forward's
// LDS-resident embeddings):
//   A  live edges   output MLP backward                       d_out -> DE[e] = d edges1[e]
//   B  live edges   edge update backward                      SE[e] = dz1;  DE[e] += W1_self^T dz1
//   C1 nodes        DN[v] = W1_src^T sum SE[e: src = v] + W1_tgt^T sum SE[e: tgt = v]   (aggregate FIRST, multiply once per node);
//                   the edge update's dW1 columns of the two endpoint segments = (sum SE)^T-contractions with nodes1[v]
//   C2 nodes        node update backward                      DN[v] += W1_self^T dz1;  XIN[v], XOUT[v] = scale W1_in/out^T dz1
//   D  all edges    initial edge backward on DE[e] + XIN[tgt] + XOUT[src]          SIE[e] = dz1
//   E1 nodes        DN[v] += W1_src^T sum SIE[e: src = v] + W1_tgt^T sum SIE[e: tgt = v];  initial edge's dW1 likewise
//   E2 nodes        initial node backward; the pipeline rows of its input gradient are added to the state gradient
// FP32 throughout.  Summation orders differ from the per-MLP launches' (which differ from autograd's): agreement to rounding.
#include "env_step_body.h"
#include "gnn_alloc_body.h"
#include "nic_common.h"
#include "small_rollout_body.h"

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NB = 16;           // scenarios per block (the N of the MFMA)
constexpr int kWaves = 8;        // two per SIMD: up to 80 accumulator registers per MLP + the chain's operands need the 256-register budget
constexpr int kTile = 512;       // floats of one tile
constexpr int kScrLd = 20;       // padded row of the wave-private transposition tile ([32][20] floats: both access patterns conflict-free)
constexpr int kScr = 32 * kScrLd;
constexpr int kRed = kWaves * 4 * 256;   // floats of the cross-wave reduction area (four 16 x 16 blocks per wavefront)

#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_gnnb_stamps = nullptr;
#define GNNB_STAMP(point)                                                                          \
    do {                                                                                           \
        if (g_gnnb_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0)                \
            g_gnnb_stamps[(threadIdx.x >> 6) * 16 + (point)] = wall_clock64();                     \
    } while (0)
#else
#define GNNB_STAMP(point) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ int uni(const int* tab, int i) { return __builtin_amdgcn_readfirstlane(tab[i]); }
__device__ __forceinline__ f32x4 elu_grad4(f32x4 d, f32x4 y) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = d[i] * nic::elu1_grad_from_out(y[i]);
    return o;
}

// Rows of a [rows][.] buffer seen from one (entity, block): a buffer descriptor based at (row 0, the entity's column, the block's
// first scenario), rows as scalar offsets (csrc/gnn_period.hip: RowOut).
struct Rows {
    __amdgpu_buffer_rsrc_t r;
    int row_bytes;
};
__device__ __forceinline__ Rows rows_at(const float* p, int64_t row_stride) {
    Rows o;
    o.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
    o.row_bytes = (int)row_stride * 4;
    return o;
}
constexpr int kDeadLane = 0x7fffffff;
// D layout: rows 16 rb + 4 g + i of scenario n
__device__ __forceinline__ f32x4 get_d(const Rows& o, int rb, int g, int n, bool live) {
    const int voff = live ? 4 * g * o.row_bytes + 4 * n : kDeadLane;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o.r, voff, (16 * rb + i) * o.row_bytes, 0));
    return v;
}
// T layout (the B operand of a weight-gradient contraction): row 16 cb + m, scenarios 4 q .. 4 q + 3
__device__ __forceinline__ f32x4 get_t(const Rows& o, int cb, int m, int q) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o.r, m * o.row_bytes + 16 * q, 16 * cb * o.row_bytes, 0));
}
__device__ __forceinline__ void add_rows(float* p, int64_t row_stride, int row, float v) { p[(int64_t)row * row_stride] += v; }

// tiles in the workgroup's scratch area (global memory), in the lanes' own order
struct Tile2 {
    f32x4 a, b;
};
__device__ __forceinline__ Tile2 tile_load(const float* t, int lane) { return Tile2{lds4(t + lane * 4), lds4(t + 256 + lane * 4)}; }
__device__ __forceinline__ void tile_store(float* t, int lane, f32x4 a, f32x4 b) {
    *reinterpret_cast<f32x4*>(t + lane * 4) = a;
    *reinterpret_cast<f32x4*>(t + 256 + lane * 4) = b;
}

// o += M x for a packed 32 x 32 matrix ([rb][q][lane][4] A fragments in LDS) and x in D layout
__device__ __forceinline__ void tmul32(const float* pk, f32x4 x0, f32x4 x1, int lane, f32x4& o0, f32x4& o1) {
    const f32x4 w00 = lds4(pk + (0 * 64 + lane) * 4), w01 = lds4(pk + (1 * 64 + lane) * 4);
    const f32x4 w10 = lds4(pk + (2 * 64 + lane) * 4), w11 = lds4(pk + (3 * 64 + lane) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o0 = mfma4(w00[j], x0[j], o0);
        o1 = mfma4(w10[j], x0[j], o1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o0 = mfma4(w01[j], x1[j], o0);
        o1 = mfma4(w11[j], x1[j], o1);
    }
}

// D layout -> the wave's transposition tile -> A operand of a weight-gradient contraction (row 16 rb + m, scenarios 4 q .. 4 q + 3)
__device__ __forceinline__ void transpose(float* scr, f32x4 d0, f32x4 d1, int lane, f32x4 (&a)[2]) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        scr[(4 * g + i) * kScrLd + n] = d0[i];
        scr[(16 + 4 * g + i) * kScrLd + n] = d1[i];
    }
    __builtin_amdgcn_wave_barrier();
    a[0] = lds4(scr + n * kScrLd + 4 * g);
    a[1] = lds4(scr + (16 + n) * kScrLd + 4 * g);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// dW block (rbo, cb) += dz^T-fragment x^T-fragment over the block's 16 scenarios
__device__ __forceinline__ void wgrad32(const f32x4 (&a)[2], const f32x4 (&xt)[2], f32x4 (&w)[2][2], bool one_row_block = false) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w[0][cb] = mfma4(a[0][i], xt[cb][i], w[0][cb]);
            if (!one_row_block) w[1][cb] = mfma4(a[1][i], xt[cb][i], w[1][cb]);
        }
}

// Transposed packs of one MLP in LDS: [L3^T][L2^T][L1^T segment 0][segment 1]...; a 32 x 32 layer is 1024 floats, the output MLP's
// 32 x 1 last layer 512 (one contraction step)
struct PackT {
    const float *l3, *l2, *l1;
};
__host__ __device__ inline int packt_floats(int n_out, int segs) { return (n_out == 1 ? 512 : 1024) + 1024 + 1024 * segs; }
__device__ __forceinline__ PackT packt_at(const float* base, int n_out) {
    PackT W;
    W.l3 = base;
    W.l2 = base + (n_out == 1 ? 512 : 1024);
    W.l1 = W.l2 + 1024;
    return W;
}

// weight-gradient accumulators of one MLP with KS 32-row input segments
template <int KS>
struct Acc {
    f32x4 w1[KS][2][2], w2[2][2], w3[2][2];
    float b1[2], b2[2], b3[2], lead[2];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                w2[rb][cb] = zero4();
                w3[rb][cb] = zero4();
#pragma unroll
                for (int s = 0; s < KS; ++s) w1[s][rb][cb] = zero4();
            }
            b1[rb] = b2[rb] = b3[rb] = lead[rb] = 0.f;
        }
    }
};

// Layers 3 and 2 of an MLP's backward for one tile, and dz1 (D layout + its transposed fragments).  dz3 = dY * act'(Y), already masked.
template <bool NOUT1, int KS>
__device__ __forceinline__ void mlp_bwd_tile(const PackT& W, float* scr, int lane, bool live, f32x4 dz30, f32x4 dz31, const Rows& H2,
                                             const Rows& H1, Acc<KS>& A, f32x4 (&dz1)[2], f32x4 (&a1)[2]) {
    const int n = lane & 15, g = lane >> 4;
    const f32x4 h2d0 = get_d(H2, 0, g, n, live), h2d1 = get_d(H2, 1, g, n, live);
    f32x4 h2t[2] = {get_t(H2, 0, n, g), get_t(H2, 1, n, g)};
    const f32x4 h1d0 = get_d(H1, 0, g, n, live), h1d1 = get_d(H1, 1, g, n, live);
    f32x4 h1t[2] = {get_t(H1, 0, n, g), get_t(H1, 1, n, g)};
    f32x4 a[2];
    // layer 3: dW3 += dz3 h2^T, dh2 = W3^T dz3
    transpose(scr, dz30, dz31, lane, a);
    A.b3[0] += sum4(a[0]);
    if (!NOUT1) A.b3[1] += sum4(a[1]);
    wgrad32(a, h2t, A.w3, NOUT1);
    f32x4 d0 = zero4(), d1 = zero4();
    if (NOUT1) {   // one output row: one contraction step (row 0 lives in lane group 0, register 0; the other groups' weights are zero)
        const f32x4 w0 = lds4(W.l3 + (0 * 64 + lane) * 4), w1 = lds4(W.l3 + (1 * 64 + lane) * 4);
        d0 = mfma4(w0[0], dz30[0], d0);
        d1 = mfma4(w1[0], dz30[0], d1);
    } else {
        tmul32(W.l3, dz30, dz31, lane, d0, d1);
    }
    const f32x4 dz20 = elu_grad4(d0, h2d0), dz21 = elu_grad4(d1, h2d1);
    // layer 2
    transpose(scr, dz20, dz21, lane, a);
    A.b2[0] += sum4(a[0]);
    A.b2[1] += sum4(a[1]);
    wgrad32(a, h1t, A.w2);
    d0 = zero4();
    d1 = zero4();
    tmul32(W.l2, dz20, dz21, lane, d0, d1);
    dz1[0] = elu_grad4(d0, h1d0);
    dz1[1] = elu_grad4(d1, h1d1);
    transpose(scr, dz1[0], dz1[1], lane, a1);
    A.b1[0] += sum4(a1[0]);
    A.b1[1] += sum4(a1[1]);
}

// ---- the workgroup's slab slot += sum over its wavefronts (in wave order) of their accumulators ---------------------------------
// One 32 x 32 group (four 16 x 16 blocks [rbo][cb]) per round: every wavefront writes its blocks to LDS, then each thread adds the
// eight copies of two elements and adds the sum to the slab.  The slab's old values are fetched by the caller-independent prologue
// below BEFORE the rounds, so that no round waits for global memory.
struct SlabRef {
    float* p;        // the workgroup's slot: [rows][ld]
    int64_t ld;
    int rows, cols;  // valid rows / columns (of the whole matrix; a group covers columns col0 .. col0 + 31)
};
__device__ __forceinline__ int64_t group_addr(const SlabRef& S, int col0, bool& ok) {
    const int t = threadIdx.x, c = t >> 7, lane = (t >> 1) & 63, half = t & 1;
    const int row = 16 * (c >> 1) + 4 * (lane >> 4) + 2 * half, col = col0 + 16 * (c & 1) + (lane & 15);
    ok = row < S.rows && col < S.cols;   // (rows come in pairs: row + 1 is checked by the caller)
    return (int64_t)row * S.ld + col;
}
struct Old2 {
    float o0, o1;
};
__device__ __forceinline__ Old2 group_fetch(const SlabRef& S, int col0) {
    bool ok;
    const int64_t at = group_addr(S, col0, ok);
    const int t = threadIdx.x, c = t >> 7, ln = (t >> 1) & 63, half = t & 1;
    const int row = 16 * (c >> 1) + 4 * (ln >> 4) + 2 * half;
    Old2 o{0.f, 0.f};
    if (ok) {
        o.o0 = S.p[at];
        if (row + 1 < S.rows) o.o1 = S.p[at + S.ld];
    }
    return o;
}
__device__ __forceinline__ void group_reduce(float* red, const f32x4 (&w)[2][2], const SlabRef& S, int col0, int wave, int lane,
                                             const Old2& old) {
    float* mine = red + wave * 1024;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4*>(mine + (2 * rb + cb) * 256 + lane * 4) = w[rb][cb];
    bool ok;
    const int64_t at = group_addr(S, col0, ok);
    const int t = threadIdx.x, c = t >> 7, ln = (t >> 1) & 63, half = t & 1;
    const int row = 16 * (c >> 1) + 4 * (ln >> 4) + 2 * half;
    const float o0 = old.o0, o1 = old.o1;
    nic::lds_barrier();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < kWaves; ++w_) {
        const float2 v = *reinterpret_cast<const float2*>(red + w_ * 1024 + c * 256 + ln * 4 + 2 * half);
        s0 += v.x;
        s1 += v.y;
    }
    if (ok) {
        S.p[at] = o0 + s0;
        if (row + 1 < S.rows) S.p[at + S.ld] = o1 + s1;
    }
    nic::lds_barrier();
}
// bias-like columns: per-lane partial sums p[kind][rb] of row 16 rb + (lane & 15) (the four lane groups hold four scenario quarters)
template <int NK>
__device__ __forceinline__ void column_reduce(float* red, const float (&p)[NK][2], float* const (&dst)[NK], const int64_t (&ld)[NK],
                                              const int (&rows)[NK], int wave, int lane) {
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) red[(wave * NK * 2 + 2 * k + rb) * 64 + lane] = p[k][rb];
    nic::lds_barrier();
    const int t = threadIdx.x, k_ = t >> 5, row = t & 31;
    if (k_ < NK) {
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < kWaves; ++w_)
#pragma unroll
            for (int q = 0; q < 4; ++q) s += red[(w_ * NK * 2 + 2 * k_ + (row >> 4)) * 64 + 16 * q + (row & 15)];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (k == k_ && row < rows[k] && dst[k] != nullptr) dst[k][(int64_t)row * ld[k]] += s;
    }
    nic::lds_barrier();
}
template <int KS>
__device__ __forceinline__ void flush(float* red, Acc<KS>& A, const NicGnnPeriodBwdMlp& M, int K, int n_out, int lead_col, int wave,
                                      int lane) {
    const int slot = blockIdx.x;
    const SlabRef S1{M.slab1 + (int64_t)slot * 32 * M.lds1, M.lds1, 32, K};
    const SlabRef S2{M.slab2 + (int64_t)slot * 32 * M.lds2, M.lds2, 32, 32};
    const SlabRef S3{M.slab3 + (int64_t)slot * n_out * M.lds3, M.lds3, n_out, 32};
    // (the slot's old values of every group first: no round below waits for global memory)
    const Old2 o3 = group_fetch(S3, 0), o2 = group_fetch(S2, 0);
    Old2 o1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) o1[s] = group_fetch(S1, 32 * s);
    group_reduce(red, A.w3, S3, 0, wave, lane, o3);
    group_reduce(red, A.w2, S2, 0, wave, lane, o2);
#pragma unroll
    for (int s = 0; s < KS; ++s) group_reduce(red, A.w1[s], S1, 32 * s, wave, lane, o1[s]);
    const float p[4][2] = {{A.b1[0], A.b1[1]}, {A.b2[0], A.b2[1]}, {A.b3[0], A.b3[1]}, {A.lead[0], A.lead[1]}};
    float* const dst[4] = {S1.p + K, S2.p + 32, S3.p + 32, lead_col >= 0 ? S1.p + lead_col : nullptr};
    const int64_t ld[4] = {S1.ld, S2.ld, S3.ld, S1.ld};
    const int rows[4] = {32, 32, n_out, 32};
    column_reduce<4>(red, p, dst, ld, rows, wave, lane);
}

// Packs go to LDS through registers, four 16-byte pieces in flight per thread (one piece per trip was 12 us for the 80 KB).
__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n_floats, int tid, int n_threads) {
    const int step = n_threads * 4;
    for (int i0 = tid * 4; i0 < n_floats; i0 += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (i0 + u * step < n_floats ? i0 + u * step : i0));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * step < n_floats) *reinterpret_cast<f32x4*>(dst + i0 + u * step) = v[u];
    }
}
__device__ __forceinline__ void copy_words(int* dst, const void* __restrict__ src, int n, int tid, int n_threads) {
    for (int i = tid; i < n; i += n_threads) dst[i] = reinterpret_cast<const int*>(src)[i];
}

// Env-step adjoint + allocation adjoint of ONE 16-scenario block on ONE wavefront (16 scenarios x 4 quad lanes): what
// csrc/gnn_alloc_env.hip's backward kernel does - the same bodies in the same order - with wavefront-scope synchronisation only, so
// that the other wavefronts of the workgroup stage the weight packs meanwhile.  The quad's partial sums go through LDS, the order
// gradients (written by the four lanes of a quad, read by its first) through global memory: both behind a fence that drains the
// wavefront's stores before its next loads.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
template <int MAXW>
__device__ __forceinline__ void env_alloc_adjoint(const NicGnnPeriodBwd& P, int blk, float* part, int lane) {
    const NicEnvStepIO& io = P.io;
    const int x = lane & 15, q = lane >> 4, S = io.dims.n_stores;
    const int64_t ldb = P.ldb, b = (int64_t)blk * NB + x;
    const bool act = b < P.n_scenarios;
    const float gr = act ? P.g_reward.p[b * P.g_reward.scn_stride] : 0.f;
    float* g_store_orders = P.g_orders;
    float* g_wh_orders = P.g_orders + (int64_t)S * ldb;
    part[q * NB + x] = act ? nic::env_ship_partial(io, 0, b, q) : 0.f;
    wave_sync();
    if (q == 0) {
        const float shipped = nic::combine4(part[0 * NB + x], part[1 * NB + x], part[2 * NB + x], part[3 * NB + x]);
        part[4 * NB + x] = act ? nic::env_bwd_warehouse<MAXW>(io, P.g_wh_out, gr, 0.f, 0, shipped, P.g_wh_in, g_wh_orders, b) : 0.f;
    }
    wave_sync();
    if (act) nic::env_bwd_stores<MAXW>(io, P.g_store_out, gr, [&](int) { return part[4 * NB + x]; }, P.g_store_in, g_store_orders, b, q);
    wave_sync();   // the order gradients (and the warehouse's on-hand gradient) are read by the q = 0 lane of the quad
    if (act && q == 0)
        nic::gnn_alloc_bwd_one(P.mlp[4].Y, io.wh_inv, P.g_orders, P.sums, P.ratio, P.scale, P.d_out, P.g_wh_in, S, P.n_edges, P.e_self,
                               P.e_supplier, P.cap_at_one, b, ldb);
}

// sum of the tiles a CSR list names, in list order, four loads in flight
__device__ __forceinline__ Tile2 list_sum(const float* tiles, const int* off, const int* items, int v, int lane, bool& any) {
    const int lo = uni(off, v), hi = uni(off, v + 1);
    f32x4 s0 = zero4(), s1 = zero4();
    for (int p0 = lo; p0 < hi; p0 += 4) {
        f32x4 t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* t = tiles + (int64_t)uni(items, p0 + u < hi ? p0 + u : p0) * kTile;
            t0[u] = lds4(t + lane * 4);
            t1[u] = lds4(t + 256 + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u < hi) {
                s0 += t0[u];
                s1 += t1[u];
            }
    }
    any = hi > lo;
    return Tile2{s0, s1};
}

// static graph in LDS (ints; floats bit-cast)
struct Tabs {
    const int *src, *tgt, *lead, *row0, *slots, *in_scale, *out_scale;
    const int *off[4], *items[4];   // node -> edges lists: 0 as source (live), 1 as target (live), 2 as source (all), 3 as target (all)
};
__host__ __device__ inline int tab_words(int N, int E, const int32_t* n_items) {
    return 3 * E + 4 * N + 4 * (N + 1) + n_items[0] + n_items[1] + n_items[2] + n_items[3];
}

// MAXW: register slots of the env adjoint's pipelines (only used with fuse_env)
template <int MAXW>
__global__ __launch_bounds__(kWaves * 64) void gnn_period_bwd_kernel(const NicGnnPeriodBwd P) {
    extern __shared__ __align__(16) float lds[];
    const int N = P.n_nodes, E = P.n_edges, L = P.n_live, nsub = P.n_sub;
    // LDS: transposed packs of the five MLPs | wave-private transposition tiles | reduction area | tables
    const int sz_in = packt_floats(32, 1), sz_ie = packt_floats(32, 2), sz_nu = packt_floats(32, 3), sz_eu = packt_floats(32, 3),
              sz_out = packt_floats(1, 1);
    float* w_in = lds;
    float* w_ie = w_in + sz_in;
    float* w_nu = w_ie + sz_ie;
    float* w_eu = w_nu + sz_nu;
    float* w_out = w_eu + sz_eu;
    float* scr_all = w_out + sz_out;
    float* red = scr_all + kWaves * kScr;
    int* tabs = reinterpret_cast<int*>(red + kRed);
    Tabs G;
    G.src = tabs;
    G.tgt = G.src + E;
    G.lead = G.tgt + E;
    G.row0 = G.lead + E;
    G.slots = G.row0 + N;
    G.in_scale = G.slots + N;
    G.out_scale = G.in_scale + N;
    {
        const int* p = G.out_scale + N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            G.off[k] = p;
            G.items[k] = p + N + 1;
            p += N + 1 + P.n_items[k];
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    float* scr = scr_all + wave * kScr;
    GNNB_STAMP(0);
    const int n_blocks = (P.n_scenarios + NB - 1) / NB;
    const int first0 = blockIdx.x * nsub;
    // fuse_env: the first round's env / allocation adjoint on one wavefront per sub-block WHILE the others stage packs and tables
    const int env_waves = P.fuse_env ? ((n_blocks - first0) < nsub ? (n_blocks - first0) : nsub) : 0;
    if (wave < env_waves) {
        env_alloc_adjoint<MAXW>(P, first0 + wave, red + wave * (5 * NB), lane);
    } else {
        const int tid = threadIdx.x - env_waves * 64, nt = (kWaves - env_waves) * 64;
        copy_to_lds(w_in, P.mlp[0].wpk_t, sz_in, tid, nt);
        copy_to_lds(w_ie, P.mlp[1].wpk_t, sz_ie, tid, nt);
        copy_to_lds(w_nu, P.mlp[2].wpk_t, sz_nu, tid, nt);
        copy_to_lds(w_eu, P.mlp[3].wpk_t, sz_eu, tid, nt);
        copy_to_lds(w_out, P.mlp[4].wpk_t, sz_out, tid, nt);
        copy_words(const_cast<int*>(G.src), P.src, E, tid, nt);
        copy_words(const_cast<int*>(G.tgt), P.tgt, E, tid, nt);
        copy_words(const_cast<int*>(G.lead), P.lead, E, tid, nt);
        copy_words(const_cast<int*>(G.row0), P.node_row0, N, tid, nt);
        copy_words(const_cast<int*>(G.slots), P.node_slots, N, tid, nt);
        copy_words(const_cast<int*>(G.in_scale), P.agg_scale, N, tid, nt);
        copy_words(const_cast<int*>(G.out_scale), P.agg_scale + N, N, tid, nt);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            copy_words(const_cast<int*>(G.off[k]), P.list_off[k], N + 1, tid, nt);
            copy_words(const_cast<int*>(G.items[k]), P.list_items[k], P.n_items[k], tid, nt);
        }
    }
    __syncthreads();
    GNNB_STAMP(1);
    const PackT WIN = packt_at(w_in, 32), WIE = packt_at(w_ie, 32), WNU = packt_at(w_nu, 32), WEU = packt_at(w_eu, 32),
                WOUT = packt_at(w_out, 1);
    const int64_t ldb = P.ldb;
    const int64_t hent = (ldb / 32) * 1024;   // floats between entities in a native history
    // scratch tiles of the workgroup: [sub][DE E | SE L | DN N | XIN N | XOUT N | SIE E]
    const int per_sub = 2 * E + L + 3 * N;
    float* const T0 = P.scratch + (int64_t)blockIdx.x * nsub * per_sub * kTile;
    const int oDE = 0, oSE = E, oDN = E + L, oXI = E + L + N, oXO = E + L + 2 * N, oSI = E + L + 3 * N;

    for (int first = first0; first < n_blocks; first += gridDim.x * nsub) {
        // per-item geometry: sub-block `sub` of this round
        auto block_of = [&](int sub, int64_t& b0, bool& live, int64_t& hblk, float*& tiles) {
            const int bk = first + sub;
            b0 = (int64_t)bk * NB;
            live = bk < n_blocks && b0 + n < P.n_scenarios;
            hblk = (b0 >> 5) * 1024 + (b0 & 31);
            tiles = T0 + (int64_t)sub * per_sub * kTile;
        };
        const int subs = (n_blocks - first) < nsub ? (n_blocks - first) : nsub;   // sub-blocks that exist in this round
        if (P.fuse_env && first != first0) {   // (later rounds of a grid smaller than the batch: the first round's ran beside the staging)
            if (wave < subs) env_alloc_adjoint<MAXW>(P, first + wave, red + wave * (5 * NB), lane);
            __syncthreads();
        }

        // ---- A: output MLP (live edges): dz3 = d_out * softplus'(z) = d_out * (1 - exp(-y)) ------------------------------------------
        {
            Acc<1> A;
            A.clear();
            const NicGnnPeriodBwdMlp& M = P.mlp[4];
            for (int item = wave; item < subs * L; item += kWaves) {
                const int e = item / subs, sub = item - e * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const bool lane0 = live && g == 0;
                const float gy = lane0 ? P.d_out[(int64_t)e * ldb + b0 + n] : 0.f;
                const float yo = lane0 ? M.Y[(int64_t)e * ldb + b0 + n] : 0.f;
                const float dz = gy * (1.f - expf(-yo));
                const Rows H2 = rows_at(M.H2 + e * hent + hblk, 32), H1 = rows_at(M.H1 + e * hent + hblk, 32);
                const Rows X = rows_at(P.edges1 + (int64_t)e * ldb + b0, P.edge_row_stride);
                f32x4 xt[2] = {get_t(X, 0, n, g), get_t(X, 1, n, g)};
                f32x4 dz1[2], a1[2];
                mlp_bwd_tile<true, 1>(WOUT, scr, lane, live, f32x4{dz, 0.f, 0.f, 0.f}, zero4(), H2, H1, A, dz1, a1);
                wgrad32(a1, xt, A.w1[0]);
                f32x4 d0 = zero4(), d1 = zero4();
                tmul32(WOUT.l1, dz1[0], dz1[1], lane, d0, d1);
                tile_store(tiles + (int64_t)(oDE + e) * kTile, lane, d0, d1);
            }
            flush<1>(red, A, M, 32, 1, -1, wave, lane);
        }
        __syncthreads();
        GNNB_STAMP(2);
        // ---- B: edge update (live edges): edges1 = edges0 + MLP([edges0 | nodes1[src] | nodes1[tgt]]) -------------------------------
        Acc<3> AEU;
        AEU.clear();
        {
            const NicGnnPeriodBwdMlp& M = P.mlp[3];
            for (int item = wave; item < subs * L; item += kWaves) {
                const int e = item / subs, sub = item - e * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const Tile2 dy = tile_load(tiles + (int64_t)(oDE + e) * kTile, lane);
                const Rows Y = rows_at(M.Y + (int64_t)e * ldb + b0, M.row_stride);
                const f32x4 y0 = get_d(Y, 0, g, n, live), y1 = get_d(Y, 1, g, n, live);
                const Rows H2 = rows_at(M.H2 + e * hent + hblk, 32), H1 = rows_at(M.H1 + e * hent + hblk, 32);
                const Rows X = rows_at(P.edges0 + (int64_t)e * ldb + b0, P.edge_row_stride);
                f32x4 xt[2] = {get_t(X, 0, n, g), get_t(X, 1, n, g)};
                f32x4 dz1[2], a1[2];
                mlp_bwd_tile<false, 3>(WEU, scr, lane, live, live ? elu_grad4(dy.a, y0) : zero4(), live ? elu_grad4(dy.b, y1) : zero4(), H2,
                                       H1, AEU, dz1, a1);
                wgrad32(a1, xt, AEU.w1[0]);
                tile_store(tiles + (int64_t)(oSE + e) * kTile, lane, dz1[0], dz1[1]);
                f32x4 d0 = dy.a, d1 = dy.b;
                tmul32(WEU.l1, dz1[0], dz1[1], lane, d0, d1);
                tile_store(tiles + (int64_t)(oDE + e) * kTile, lane, d0, d1);
            }
        }
        __syncthreads();
        GNNB_STAMP(3);
        // ---- C1: nodes: d nodes1[v] = W1_src^T sum dz1[e: src = v] + W1_tgt^T sum dz1[e: tgt = v]; the same sums give the edge
        // update's dW1 columns of the endpoint segments (sum_e dz1[e] nodes1[src e]^T = sum_v (sum_{e: src = v} dz1[e]) nodes1[v]^T)
        {
            for (int item = wave; item < subs * N; item += kWaves) {
                const int v = item / subs, sub = item - v * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const Rows X = rows_at(P.nodes1 + (int64_t)v * ldb + b0, P.node_row_stride);
                f32x4 xt[2] = {get_t(X, 0, n, g), get_t(X, 1, n, g)};
                f32x4 d0 = zero4(), d1 = zero4();
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    bool any;
                    const Tile2 s = list_sum(tiles + (int64_t)oSE * kTile, G.off[side], G.items[side], v, lane, any);
                    if (any) {
                        tmul32(WEU.l1 + 1024 * (1 + side), s.a, s.b, lane, d0, d1);
                        f32x4 a[2];
                        transpose(scr, s.a, s.b, lane, a);
                        wgrad32(a, xt, AEU.w1[1 + side]);
                    }
                }
                tile_store(tiles + (int64_t)(oDN + v) * kTile, lane, d0, d1);
            }
            flush<3>(red, AEU, P.mlp[3], 96, 32, -1, wave, lane);
        }
        __syncthreads();
        GNNB_STAMP(4);
        // ---- C2: node update: nodes1 = nodes0 + MLP([nodes0 | agg_in | agg_out]) ---------------------------------------------------
        {
            Acc<3> A;
            A.clear();
            const NicGnnPeriodBwdMlp& M = P.mlp[2];
            for (int item = wave; item < subs * N; item += kWaves) {
                const int v = item / subs, sub = item - v * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const Tile2 dy = tile_load(tiles + (int64_t)(oDN + v) * kTile, lane);
                const Rows Y = rows_at(M.Y + (int64_t)v * ldb + b0, M.row_stride);
                const f32x4 y0 = get_d(Y, 0, g, n, live), y1 = get_d(Y, 1, g, n, live);
                const Rows H2 = rows_at(M.H2 + v * hent + hblk, 32), H1 = rows_at(M.H1 + v * hent + hblk, 32);
                const Rows X0 = rows_at(P.nodes0 + (int64_t)v * ldb + b0, P.node_row_stride);
                const Rows X1 = rows_at(P.agg + (int64_t)v * ldb + b0, 2 * P.node_row_stride);
                const Rows X2 = rows_at(P.agg + (int64_t)(N + v) * ldb + b0, 2 * P.node_row_stride);
                f32x4 xt0[2] = {get_t(X0, 0, n, g), get_t(X0, 1, n, g)};
                f32x4 xt1[2] = {get_t(X1, 0, n, g), get_t(X1, 1, n, g)};
                f32x4 xt2[2] = {get_t(X2, 0, n, g), get_t(X2, 1, n, g)};
                f32x4 dz1[2], a1[2];
                mlp_bwd_tile<false, 3>(WNU, scr, lane, live, live ? elu_grad4(dy.a, y0) : zero4(), live ? elu_grad4(dy.b, y1) : zero4(), H2,
                                       H1, A, dz1, a1);
                wgrad32(a1, xt0, A.w1[0]);
                wgrad32(a1, xt1, A.w1[1]);
                wgrad32(a1, xt2, A.w1[2]);
                f32x4 d0 = dy.a, d1 = dy.b;
                tmul32(WNU.l1, dz1[0], dz1[1], lane, d0, d1);
                tile_store(tiles + (int64_t)(oDN + v) * kTile, lane, d0, d1);
                const float si = __int_as_float(uni(G.in_scale, v)), so = __int_as_float(uni(G.out_scale, v));
                d0 = zero4();
                d1 = zero4();
                tmul32(WNU.l1 + 1024, dz1[0], dz1[1], lane, d0, d1);
                tile_store(tiles + (int64_t)(oXI + v) * kTile, lane, d0 * si, d1 * si);
                d0 = zero4();
                d1 = zero4();
                tmul32(WNU.l1 + 2048, dz1[0], dz1[1], lane, d0, d1);
                tile_store(tiles + (int64_t)(oXO + v) * kTile, lane, d0 * so, d1 * so);
            }
            flush<3>(red, A, M, 96, 32, -1, wave, lane);
        }
        __syncthreads();
        GNNB_STAMP(5);
        // ---- D: initial edge: edges0 = MLP([nodes0[src] | nodes0[tgt] | lead]) on d edges0 = DE + XIN[tgt] + XOUT[src] ----------------
        Acc<2> AIE;
        AIE.clear();
        {
            const NicGnnPeriodBwdMlp& M = P.mlp[1];
            for (int item = wave; item < subs * E; item += kWaves) {
                const int e = item / subs, sub = item - e * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const int s_ = uni(G.src, e), t_ = uni(G.tgt, e);
                const float lead = __int_as_float(uni(G.lead, e));
                // (every tile is loaded - a missing one from a valid dummy address - and dropped by a select)
                const Tile2 de = tile_load(tiles + (int64_t)(oDE + (e < L ? e : 0)) * kTile, lane);
                const Tile2 xi = tile_load(tiles + (int64_t)(oXI + (t_ >= 0 ? t_ : 0)) * kTile, lane);
                const Tile2 xo = tile_load(tiles + (int64_t)(oXO + (s_ >= 0 ? s_ : 0)) * kTile, lane);
                f32x4 dy0 = e < L ? de.a : zero4(), dy1 = e < L ? de.b : zero4();
                if (t_ >= 0) {
                    dy0 += xi.a;
                    dy1 += xi.b;
                }
                if (s_ >= 0) {
                    dy0 += xo.a;
                    dy1 += xo.b;
                }
                const Rows Y = rows_at(M.Y + (int64_t)e * ldb + b0, M.row_stride);
                const f32x4 y0 = get_d(Y, 0, g, n, live), y1 = get_d(Y, 1, g, n, live);
                const Rows H2 = rows_at(M.H2 + e * hent + hblk, 32), H1 = rows_at(M.H1 + e * hent + hblk, 32);
                f32x4 dz1[2], a1[2];
                mlp_bwd_tile<false, 2>(WIE, scr, lane, live, live ? elu_grad4(dy0, y0) : zero4(), live ? elu_grad4(dy1, y1) : zero4(), H2, H1,
                                       AIE, dz1, a1);
                AIE.lead[0] += lead * sum4(a1[0]);
                AIE.lead[1] += lead * sum4(a1[1]);
                tile_store(tiles + (int64_t)(oSI + e) * kTile, lane, dz1[0], dz1[1]);
            }
        }
        __syncthreads();
        GNNB_STAMP(6);
        // ---- E1: nodes: d nodes0[v] = DN[v] + W1_src^T sum dz1[e: src = v] + W1_tgt^T sum dz1[e: tgt = v] (all edges) -----------------
        {
            for (int item = wave; item < subs * N; item += kWaves) {
                const int v = item / subs, sub = item - v * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const Rows X = rows_at(P.nodes0 + (int64_t)v * ldb + b0, P.node_row_stride);
                f32x4 xt[2] = {get_t(X, 0, n, g), get_t(X, 1, n, g)};
                const Tile2 dn = tile_load(tiles + (int64_t)(oDN + v) * kTile, lane);
                f32x4 d0 = dn.a, d1 = dn.b;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    bool any;
                    const Tile2 s = list_sum(tiles + (int64_t)oSI * kTile, G.off[2 + side], G.items[2 + side], v, lane, any);
                    if (any) {
                        tmul32(WIE.l1 + 1024 * side, s.a, s.b, lane, d0, d1);
                        f32x4 a[2];
                        transpose(scr, s.a, s.b, lane, a);
                        wgrad32(a, xt, AIE.w1[side]);
                    }
                }
                tile_store(tiles + (int64_t)(oDN + v) * kTile, lane, d0, d1);
            }
            flush<2>(red, AIE, P.mlp[1], 65, 32, 64, wave, lane);
        }
        __syncthreads();
        GNNB_STAMP(7);
        // ---- E2: initial node: nodes0 = MLP(features); the pipeline rows of d features are the state's ---------------------------------
        {
            Acc<1> A;
            A.clear();
            const NicGnnPeriodBwdMlp& M = P.mlp[0];
            for (int item = wave; item < subs * N; item += kWaves) {
                const int v = item / subs, sub = item - v * subs;   // (the sub-blocks of an entity side by side)
                int64_t b0, hblk;
                bool live;
                float* tiles;
                block_of(sub, b0, live, hblk, tiles);
                const Tile2 dy = tile_load(tiles + (int64_t)(oDN + v) * kTile, lane);
                const Rows Y = rows_at(M.Y + (int64_t)v * ldb + b0, M.row_stride);
                const f32x4 y0 = get_d(Y, 0, g, n, live), y1 = get_d(Y, 1, g, n, live);
                const Rows H2 = rows_at(M.H2 + v * hent + hblk, 32), H1 = rows_at(M.H1 + v * hent + hblk, 32);
                const Rows X = rows_at(P.feat + (int64_t)v * ldb + b0, P.node_row_stride);
                f32x4 xt[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {   // feature rows past Dn do not exist: a clamped row, dropped by a select
                    const bool has = 16 * cb + n < P.Dn;
                    const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(X.r, (has ? 16 * cb + n : 0) * X.row_bytes + 16 * g, 0, 0));
                    xt[cb] = has ? v4 : zero4();
                }
                f32x4 dz1[2], a1[2];
                mlp_bwd_tile<false, 1>(WIN, scr, lane, live, live ? elu_grad4(dy.a, y0) : zero4(), live ? elu_grad4(dy.b, y1) : zero4(), H2,
                                       H1, A, dz1, a1);
                wgrad32(a1, xt, A.w1[0]);
                f32x4 d0 = zero4(), d1 = zero4();
                tmul32(WIN.l1, dz1[0], dz1[1], lane, d0, d1);
                const int row0 = uni(G.row0, v), slots = uni(G.slots, v);
                if (live) {
                    float* gs = P.g_state + b0 + n;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (4 * g + i < slots) add_rows(gs, ldb, row0 + 4 * g + i, d0[i]);
                        if (16 + 4 * g + i < slots) add_rows(gs, ldb, row0 + 16 + 4 * g + i, d1[i]);
                    }
                }
            }
            flush<1>(red, A, M, P.Dn, 32, -1, wave, lane);
        }
        __syncthreads();
        GNNB_STAMP(8);
    }
}

int lds_bytes(const NicGnnPeriodBwd& p) {
    const int packs = packt_floats(32, 1) + packt_floats(32, 2) + 2 * packt_floats(32, 3) + packt_floats(1, 1);
    const int tab = (tab_words(p.n_nodes, p.n_edges, p.n_items) + 3) / 4 * 4;
    return (packs + kWaves * kScr + kRed + tab) * 4;
}
constexpr int kLdsLimit = 160 * 1024;
constexpr int kMaxGrid = 512;   // = nic_mlp3_bwd_hist_slots(): the slabs of the per-MLP backward serve both

int check(const NicGnnPeriodBwd* p, const char* who) {
    NIC_REQUIRE(p != nullptr, "%s: descriptor is null", who);
    NIC_REQUIRE(p->n_nodes > 0 && p->n_edges > 0 && p->n_live > 0 && p->n_live <= p->n_edges, "%s: bad graph sizes", who);
    NIC_REQUIRE(p->n_scenarios > 0 && p->ldb >= p->n_scenarios && p->ldb % 32 == 0, "%s: bad scenario sizes (ldb a multiple of 32)", who);
    NIC_REQUIRE(p->Dn > 0 && p->Dn <= 32, "%s: node feature rows outside [1,32]", who);
    NIC_REQUIRE(p->n_sub >= 1 && p->n_sub <= 4, "%s: n_sub outside [1,4]", who);
    NIC_REQUIRE(p->src && p->tgt && p->lead && p->node_row0 && p->node_slots && p->agg_scale, "%s: null graph table", who);
    for (int k = 0; k < 4; ++k)
        NIC_REQUIRE(p->list_off[k] && p->list_items[k] && p->n_items[k] >= 0 && p->n_items[k] <= p->n_edges, "%s: bad edge list %d", who, k);
    NIC_REQUIRE(p->feat && p->nodes0 && p->nodes1 && p->edges0 && p->edges1 && p->agg && p->d_out && p->g_state && p->scratch,
                "%s: null buffer", who);
    NIC_REQUIRE(p->node_row_stride >= (int64_t)p->n_nodes * p->ldb && p->edge_row_stride >= (int64_t)p->n_edges * p->ldb &&
                    32 * 2 * p->node_row_stride < (1ll << 29) && 32 * p->edge_row_stride < (1ll << 29),
                "%s: row strides (a 32-row block must stay below 2 GiB)", who);
    for (int i = 0; i < 5; ++i) {
        const NicGnnPeriodBwdMlp& m = p->mlp[i];
        NIC_REQUIRE(m.wpk_t && m.Y && m.H1 && m.H2 && m.slab1 && m.slab2 && m.slab3, "%s: MLP %d: null buffer", who, i);
        NIC_REQUIRE(m.row_stride > 0 && 32 * m.row_stride < (1ll << 29) && m.lds1 >= (i == 0 ? p->Dn : (i == 1 ? 65 : (i == 4 ? 32 : 96))) + 1 &&
                        m.lds2 >= 33 && m.lds3 >= 33, "%s: MLP %d: strides", who, i);
    }
    NIC_REQUIRE(lds_bytes(*p) <= kLdsLimit, "%s: the graph's tables do not fit in LDS", who);
    if (p->fuse_env) {
        const NicEnvDims& d = p->io.dims;
        NIC_REQUIRE(d.n_warehouses == 1 && d.n_echelons == 0, "%s: fused env / allocation adjoint: one supplying warehouse, no extra echelons", who);
        NIC_REQUIRE(d.n_scenarios == p->n_scenarios && d.ldb == p->ldb && p->n_edges > d.n_stores, "%s: env dims differ from the policy's", who);
        NIC_REQUIRE(d.store_slots >= 2 && d.store_slots <= NIC_MAX_SLOTS && d.warehouse_slots >= 2 && d.warehouse_slots <= NIC_MAX_SLOTS,
                    "%s: pipeline lengths outside [2,%d]", who, NIC_MAX_SLOTS);
        NIC_REQUIRE(p->io.store_inv && p->io.wh_inv && p->io.demand.p && p->io.store_orders.p && p->io.wh_orders.p && p->io.underage.p &&
                        p->io.holding.p && p->io.lead_times.p && p->io.wh_holding.p && p->io.wh_lead_times.p,
                    "%s: null env table", who);
        NIC_REQUIRE(p->sums && p->ratio && p->scale && p->g_reward.p && p->g_store_in && p->g_wh_in && p->g_orders && p->e_supplier >= 0,
                    "%s: null buffer of the fused env / allocation adjoint", who);
    }
    return 0;
}
}  // namespace

extern "C" {

int64_t nic_gnn_period_bwd_scratch_floats(int32_t n_nodes, int32_t n_edges, int32_t n_live, int32_t n_scenarios, int32_t n_sub) {
    const int64_t n_blocks = (n_scenarios + NB - 1) / NB;
    int64_t grid = (n_blocks + n_sub - 1) / n_sub;
    if (grid > kMaxGrid) grid = kMaxGrid;
    return grid * n_sub * (2 * (int64_t)n_edges + n_live + 3 * (int64_t)n_nodes) * kTile;
}

int nic_gnn_period_bwd(const NicGnnPeriodBwd* p, void* stream) {
    if (int e = check(p, "nic_gnn_period_bwd")) return e;
    const int bytes = lds_bytes(*p);
    const int n_blocks = nic::ceil_div(p->n_scenarios, NB);
    int grid = nic::ceil_div(n_blocks, p->n_sub);
    if (grid > kMaxGrid) grid = kMaxGrid;
    hipStream_t s = nic::as_stream(stream);
    int m = 4;
    if (p->fuse_env) {
        const int sl = p->io.dims.store_slots > p->io.dims.warehouse_slots ? p->io.dims.store_slots : p->io.dims.warehouse_slots;
        m = sl <= 4 ? 4 : (sl <= 8 ? 8 : NIC_MAX_SLOTS);
    }
    nic::note_kernelf("gnn_period_bwd_kernel<%d>%s", m, p->fuse_env ? " (+ env / allocation adjoint)" : "");
#define NIC_GPB(MW)                                                                                                                  \
    do {                                                                                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gnn_period_bwd_kernel<MW>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                kLdsLimit) != hipSuccess)                                                                            \
            return nic::fail("nic_gnn_period_bwd: cannot raise the dynamic LDS limit");                                              \
        hipLaunchKernelGGL(gnn_period_bwd_kernel<MW>, dim3(grid), dim3(kWaves * 64), bytes, s, *p);                                  \
    } while (0)
    if (m == 4) NIC_GPB(4);
    else if (m == 8) NIC_GPB(8);
    else NIC_GPB(NIC_MAX_SLOTS);
#undef NIC_GPB
    return nic::check_launch("nic_gnn_period_bwd");
}

#ifdef NIC_TUNING_BUILD
int nic_tuning_set_gnn_bwd_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gnnb_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif
}
