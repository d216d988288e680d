// GNN policy: ONE launch per period for the forward of the whole policy (+ allocation head + env step on one-warehouse graphs)
// (include/nic_rollout.h: nic_gnn_period_fwd; the reference's neural_networks.py:1105-1192 initial embeddings, :1229-1340 message
// passing, :1367-1392 output, :1435-1492 allocation, environment.py:110-299 dynamics).  Round 4's route launches the five
// gather-MLP kernels of csrc/mlp3.hip, a segment sum, a row copy and the allocation + env kernel per period: every embedding makes
// a round trip through HBM between two launches and is gathered again (twice for the edge MLPs).  Here a workgroup owns 16
// scenarios (the N of v_mfma_f32_16x16x4_f32) and keeps the node and edge embeddings of its block in LDS across
//     initial_node -> initial_edge -> [aggregation] -> node_update -> edge_update -> output -> allocation + env step;
// HBM sees the state, the demand and - for a training run - the histories the backward reads.
//
// Tiles.  One (entity, 16 scenarios) tile is a 32 x 16 block of an MLP's output: two accumulators of the 16x16x4 MFMA (row blocks
// rb = 0, 1).  Lane (n = lane & 15, g = lane >> 4) of accumulator rb holds rows 16 rb + 4 g + i (i = 0..3) of scenario n - and that
// IS the B operand of the next layer's contraction steps (step 4 rb + i contracts over k = 16 rb + 4 g + i, g = 0..3), so the three
// layers chain in registers and an embedding is kept in LDS exactly as the lanes hold it: [entity][rb][lane] float4 (2 KB per
// entity, conflict-free 16-byte accesses).  The weights come pre-packed in that fragment order (`nic_gnn_period_pack_size`: per
// layer [rb][group of 4 steps][lane][4]) and are staged in LDS per stage; eight wavefronts walk the entities of a stage round robin.
// FP32 throughout (exact products, f32 accumulate), ELU / softplus as in csrc/mlp3.hip.  Accumulation ORDER differs from the
// mlp3 kernels' (other MFMA shape): results agree to rounding, not bit for bit - the golden fixtures bound both.
#include "env_step_body.h"
#include "gnn_alloc_body.h"
#include "nic_common.h"
#include "small_rollout_body.h"

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NB = 16;          // scenarios per workgroup
// wavefronts per workgroup: 16 (four per SIMD: a tile's layers are a dependent MFMA -> ELU -> store chain, so a SIMD needs several
// wavefronts in different phases to keep its matrix pipe fed) where the kernel fits in 128 registers, else 8
constexpr int kTile = 512;      // floats of one entity's embedding tile
constexpr int kEnvChunk = 1;                       // (the fused allocation + env step is for ONE warehouse)
constexpr int kEnvScratch = (kEnvChunk + 1) * 4 * NB + NB;   // floats: quad partial sums, store costs, the warehouse's cost

#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_gnn_stamps = nullptr;
#define GNN_STAMP(point)                                                                          \
    do {                                                                                          \
        if (g_gnn_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0)                \
            g_gnn_stamps[(threadIdx.x >> 6) * 16 + (point)] = wall_clock64();                     \
    } while (0)
// ... and inside ONE tile (the first initial-edge tile of every wavefront): points 8..15, pinned by scheduling barriers
#define GNN_TILE_STAMP(on, point)                                                                 \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((on) && g_gnn_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0)        \
            g_gnn_stamps[(threadIdx.x >> 6) * 16 + (point)] = wall_clock64();                     \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define GNN_STAMP(point) do { } while (0)
#define GNN_TILE_STAMP(on, point) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 elu4(f32x4 v) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = nic::elu1(v[i]);
    return o;
}
__device__ __forceinline__ float softplus1(float z) { return z > 20.f ? z : log1pf(expf(z)); }   // nn.Softplus(beta=1, threshold=20)

// Packed weights of one MLP (host: ops.gnn_period_pack): [L1: 2 x s1q x 256][b1 32][L2: 2 x 2 x 256][b2 32][L3: nrb3 x 2 x 256][b3 32]
__host__ __device__ inline int pack_floats(int s1q, int nrb3) { return 512 * s1q + 32 + 1024 + 32 + 512 * nrb3 + 32; }
struct MlpL {
    const float *l1, *b1, *l2, *b2, *l3, *b3;
    int s1q, nrb3;
};
__device__ __forceinline__ MlpL mlp_at(const float* base, int s1q, int nrb3) {
    MlpL W;
    W.s1q = s1q;
    W.nrb3 = nrb3;
    W.l1 = base;
    W.b1 = W.l1 + 512 * s1q;
    W.l2 = W.b1 + 32;
    W.b2 = W.l2 + 1024;
    W.l3 = W.b2 + 32;
    W.b3 = W.l3 + 512 * nrb3;
    return W;
}

struct Acc2 {
    f32x4 a0, a1;
};
__device__ __forceinline__ Acc2 l1_begin(const MlpL& W, int g) { return Acc2{lds4(W.b1 + 4 * g), lds4(W.b1 + 16 + 4 * g)}; }
// four contraction steps of the first layer (group q of the packed fragments) on the B values x[0..3]
__device__ __forceinline__ void l1_group(const MlpL& W, int q, f32x4 x, int lane, Acc2& A) {
    const f32x4 w0 = lds4(W.l1 + (q * 64 + lane) * 4), w1 = lds4(W.l1 + ((W.s1q + q) * 64 + lane) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A.a0 = mfma4(w0[j], x[j], A.a0);
        A.a1 = mfma4(w1[j], x[j], A.a1);
    }
}
// ... only its first step (the edge MLP's lead-time row)
__device__ __forceinline__ void l1_first_step(const MlpL& W, int q, float x, int lane, Acc2& A) {
    const f32x4 w0 = lds4(W.l1 + (q * 64 + lane) * 4), w1 = lds4(W.l1 + ((W.s1q + q) * 64 + lane) * 4);
    A.a0 = mfma4(w0[0], x, A.a0);
    A.a1 = mfma4(w1[0], x, A.a1);
}
// a 32-row source held as an embedding tile (two float4 fragments per lane): groups q0, q0 + 1
__device__ __forceinline__ void l1_tile(const MlpL& W, int q0, const float* tile, int lane, Acc2& A) {
    const f32x4 x0 = lds4(tile + lane * 4), x1 = lds4(tile + 256 + lane * 4);
    l1_group(W, q0, x0, lane, A);
    l1_group(W, q0 + 1, x1, lane, A);
}

// Where a tile's rows go in HBM (training runs): a buffer descriptor based at (row 0, the entity's column, the block's first
// scenario); a store adds the row as a SCALAR offset and the lane's (4 g rows + scenario) as one 32-bit offset - no 64-bit address
// per row in vector registers.  An absent output has zero records (its stores are dropped by the range check), and so is the
// lane offset of a scenario past the batch.
struct RowOut {
    __amdgpu_buffer_rsrc_t r;
    int row_bytes;
};
__device__ __forceinline__ RowOut row_out(float* p, int64_t row_stride) {
    static __device__ float sink;
    RowOut o;
    o.r = __builtin_amdgcn_make_buffer_rsrc(p ? p : &sink, 0, p ? 0x7fffffff : 0, 0x00020000);
    o.row_bytes = (int)row_stride * 4;
    return o;
}
constexpr int kDeadLane = 0x7fffffff;
__device__ __forceinline__ void put_rows(const RowOut& o, int rb, f32x4 v, int g, int n, bool live) {
    const int voff = live ? 4 * g * o.row_bytes + 4 * n : kDeadLane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), o.r, voff, (16 * rb + i) * o.row_bytes, 0);
}

// Layers 2 and 3 behind the first layer's pre-activations; h1 / h2 go to the history blocks (if any), y is returned activated.
template <bool TRAIN>
__device__ __forceinline__ void finish_mlp(const MlpL& W, int lane, int softplus_out, const Acc2& Z1, const RowOut& H1, const RowOut& H2,
                                           bool live, f32x4 (&y)[2], bool stamp = false) {
    const int g = lane >> 4, n = lane & 15;
    GNN_TILE_STAMP(stamp, 9);
    const f32x4 h10 = elu4(Z1.a0), h11 = elu4(Z1.a1);
    GNN_TILE_STAMP(stamp, 10);
    if (TRAIN) {
        put_rows(H1, 0, h10, g, n, live);
        put_rows(H1, 1, h11, g, n, live);
    }
    GNN_TILE_STAMP(stamp, 11);
    f32x4 c0 = lds4(W.b2 + 4 * g), c1 = lds4(W.b2 + 16 + 4 * g);
    {
        const f32x4 w00 = lds4(W.l2 + (0 * 64 + lane) * 4), w01 = lds4(W.l2 + (1 * 64 + lane) * 4);
        const f32x4 w10 = lds4(W.l2 + (2 * 64 + lane) * 4), w11 = lds4(W.l2 + (3 * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = mfma4(w00[j], h10[j], c0);
            c1 = mfma4(w10[j], h10[j], c1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = mfma4(w01[j], h11[j], c0);
            c1 = mfma4(w11[j], h11[j], c1);
        }
    }
    GNN_TILE_STAMP(stamp, 12);
    const f32x4 h20 = elu4(c0), h21 = elu4(c1);
    if (TRAIN) {
        put_rows(H2, 0, h20, g, n, live);
        put_rows(H2, 1, h21, g, n, live);
    }
    GNN_TILE_STAMP(stamp, 13);
    f32x4 d0 = lds4(W.b3 + 4 * g), d1 = lds4(W.b3 + 16 + 4 * g);
    {
        const f32x4 w00 = lds4(W.l3 + (0 * 64 + lane) * 4), w01 = lds4(W.l3 + (1 * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) d0 = mfma4(w00[j], h20[j], d0);
#pragma unroll
        for (int j = 0; j < 4; ++j) d0 = mfma4(w01[j], h21[j], d0);
        if (W.nrb3 == 2) {
            const f32x4 w10 = lds4(W.l3 + (2 * 64 + lane) * 4), w11 = lds4(W.l3 + (3 * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) d1 = mfma4(w10[j], h20[j], d1);
#pragma unroll
            for (int j = 0; j < 4; ++j) d1 = mfma4(w11[j], h21[j], d1);
        }
    }
    GNN_TILE_STAMP(stamp, 14);
    if (softplus_out) {   // one output row: lane group 0, register 0
        y[0] = f32x4{softplus1(d0[0]), 0.f, 0.f, 0.f};
        y[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        y[0] = elu4(d0);
        y[1] = elu4(d1);
    }
}

__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n_floats) {
    for (int i = threadIdx.x * 4; i < n_floats; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
}
__device__ __forceinline__ void copy_words(int* dst, const void* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = reinterpret_cast<const int*>(src)[i];
}
// a wave-uniform word of an LDS table as a scalar (every index that steers a wavefront - its entity, the entity's endpoints, its
// aggregation list - is uniform: as scalars they cost no vector registers, branch without exec masks and address LDS by immediates)
__device__ __forceinline__ int uni(const int* tab, int i) { return __builtin_amdgcn_readfirstlane(tab[i]); }

// The static graph in LDS (ints; floats bit-cast): [src E][tgt E][lead E][node_row0 N][node_slots N][agg_scale 2N][agg_off 2N+1][agg_items]
struct GraphTabs {
    const int *src, *tgt, *lead, *row0, *slots, *scale, *off, *items;
};
__host__ __device__ inline int graph_words(int N, int E, int n_items) { return 3 * E + 2 * N + 2 * N + (2 * N + 1) + n_items; }

// SPILL (round 6): graphs whose node + edge embeddings do not fit in LDS (the 3 x 16 dense many-warehouse graph: 19 + 70 tiles = 178 KB)
// keep the NODE tiles in LDS and the EDGE tiles in a per-workgroup scratch area in global memory, in the same lane order (2 KB
// contiguous per tile, L2-resident: written by the initial-edge stage, read by the aggregation and the edge update); every other
// line of the kernel is the same - the tile pointer's address space is a compile-time property of the instantiation.
template <int MAXW, bool TRAIN, int kWaves, bool SPILL>
__global__ __launch_bounds__(kWaves * 64) void gnn_period_fwd_kernel(const NicGnnPeriod P) {
    extern __shared__ __align__(16) float lds[];
    const int N = P.n_nodes, E = P.n_edges;
    float* nodes = lds;
    float* edges = SPILL ? P.edge_scratch + (int64_t)blockIdx.x * E * kTile : nodes + N * kTile;
    float* wb0 = nodes + (SPILL ? N : N + E) * kTile;
    float* wb1 = wb0 + P.wb0_floats;
    float* scratch = wb1 + P.wb1_floats;   // allocation + env step: [kEnvChunk + 1][4][16] + [1][16]
    int* tabs = reinterpret_cast<int*>(scratch + kEnvScratch);
    GraphTabs G;
    G.src = tabs;
    G.tgt = G.src + E;
    G.lead = G.tgt + E;
    G.row0 = G.lead + E;
    G.slots = G.row0 + N;
    G.scale = G.slots + N;
    G.off = G.scale + 2 * N;
    G.items = G.off + 2 * N + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int64_t b0 = (int64_t)blockIdx.x * NB, ldb = P.ldb;
    const bool live = b0 + n < P.n_scenarios;
    const int s1q_in = (P.Dn + 15) / 16;
    const int sz_in = pack_floats(s1q_in, 2), sz_ie = pack_floats(5, 2), sz_nu = pack_floats(6, 2), sz_eu = pack_floats(6, 2),
              sz_out = pack_floats(2, 1);
    GNN_STAMP(0);
    copy_to_lds(wb0, P.mlp[0].wpk, sz_in);
    copy_to_lds(wb0 + sz_in, P.mlp[1].wpk, sz_ie);
    copy_to_lds(wb1, P.mlp[2].wpk, sz_nu);
    copy_words(const_cast<int*>(G.src), P.src, E);
    copy_words(const_cast<int*>(G.tgt), P.tgt, E);
    copy_words(const_cast<int*>(G.lead), P.lead, E);
    copy_words(const_cast<int*>(G.row0), P.node_row0, N);
    copy_words(const_cast<int*>(G.slots), P.node_slots, N);
    copy_words(const_cast<int*>(G.scale), P.agg_scale, 2 * N);
    copy_words(const_cast<int*>(G.off), P.agg_off, 2 * N + 1);
    copy_words(const_cast<int*>(G.items), P.agg_items, P.n_agg_items);
    __syncthreads();
    GNN_STAMP(1);
    const int64_t hblk = (b0 >> 5) * 1024 + (b0 & 31);   // the block's columns inside an entity's native history blocks
    const int64_t hent = (ldb / 32) * 1024;               // floats between entities there
    auto hist = [&](float* H, int ent) { return row_out(TRAIN && H ? H + ent * hent + hblk : nullptr, 32); };
    auto rows = [&](float* Y, const NicGnnPeriodMlp& m, int ent) {
        return row_out(TRAIN && Y ? Y + (int64_t)ent * ldb + b0 : nullptr, m.row_stride);
    };
    auto put_tile = [&](float* tile, const f32x4 (&y)[2]) {
        *reinterpret_cast<f32x4*>(tile + lane * 4) = y[0];
        *reinterpret_cast<f32x4*>(tile + 256 + lane * 4) = y[1];
    };

    // ---- initial node embeddings: features = [pipeline slots padded to max_inv | static rows] (neural_networks.py:846-905)
    {
        const MlpL W = mlp_at(wb0, s1q_in, 2);
        const NicGnnPeriodMlp& M = P.mlp[0];
        for (int v = wave; v < N; v += kWaves) {
            const int row0 = uni(G.row0, v), slots = uni(G.slots, v);
            Acc2 A = l1_begin(W, g);
            for (int q = 0; q < s1q_in; ++q) {
                f32x4 x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // every lane loads (a clamped address), a select drops what is not a feature
                    const int k = 16 * q + 4 * j + g;
                    const bool pipe = k < P.max_inv;
                    const bool has = live && k < P.Dn && (!pipe || k < slots);
                    const float* src = pipe ? P.state + (int64_t)(row0 + (k < slots ? k : 0)) * ldb
                                            : P.feat + ((int64_t)(k < P.Dn ? k : 0) * N + v) * ldb;
                    const float val = src[live ? b0 + n : 0];
                    x[j] = has ? val : 0.f;
                    if (TRAIN && P.store_feat && has && pipe) P.feat[((int64_t)k * N + v) * ldb + b0 + n] = val;
                }
                l1_group(W, q, x, lane, A);
            }
            f32x4 y[2];
            finish_mlp<TRAIN>(W, lane, 0, A, hist(M.H1, v), hist(M.H2, v), live, y);
            put_tile(nodes + v * kTile, y);
            if (TRAIN) {
                const RowOut o = rows(M.Y, M, v);
                put_rows(o, 0, y[0], g, n, live);
                put_rows(o, 1, y[1], g, n, live);
            }
        }
    }
    __syncthreads();
    GNN_STAMP(2);
    // ---- initial edge embeddings: [source node | target node | lead time] (:984-1062); a missing endpoint is the all-zero node
    {
        const MlpL W = mlp_at(wb0 + sz_in, 5, 2);
        const NicGnnPeriodMlp& M = P.mlp[1];
        for (int e = wave; e < E; e += kWaves) {
            const int s_ = uni(G.src, e), t_ = uni(G.tgt, e);
            const float lead = __int_as_float(uni(G.lead, e));
            GNN_TILE_STAMP(e == wave, 8);
            Acc2 A = l1_begin(W, g);
            if (s_ >= 0) l1_tile(W, 0, nodes + s_ * kTile, lane, A);
            if (t_ >= 0) l1_tile(W, 2, nodes + t_ * kTile, lane, A);
            l1_first_step(W, 4, g == 0 ? lead : 0.f, lane, A);
            f32x4 y[2];
            finish_mlp<TRAIN>(W, lane, 0, A, hist(M.H1, e), hist(M.H2, e), live, y, e == wave);
            put_tile(edges + e * kTile, y);
            if (TRAIN) {
                const RowOut o = rows(M.Y, M, e);
                put_rows(o, 0, y[0], g, n, live);
                put_rows(o, 1, y[1], g, n, live);
            }
            GNN_TILE_STAMP(e == wave, 15);
        }
    }
    __syncthreads();   // (also: wb0 is free from here on)
    GNN_STAMP(3);
    copy_to_lds(wb0, P.mlp[3].wpk, sz_eu);
    copy_to_lds(wb0 + sz_eu, P.mlp[4].wpk, sz_out);
    // ---- node update: [node | sum over incoming edges / sqrt(in degree) | sum over outgoing edges / sqrt(out degree)] (:1229-1320),
    // the sums taken in the reference's edge order; nodes1 = nodes0 + update, in place
    {
        const MlpL W = mlp_at(wb1, 6, 2);
        const NicGnnPeriodMlp& M = P.mlp[2];
        for (int v = wave; v < N; v += kWaves) {
            Acc2 A = l1_begin(W, g);
            const f32x4 old0 = lds4(nodes + v * kTile + lane * 4), old1 = lds4(nodes + v * kTile + 256 + lane * 4);
            l1_group(W, 0, old0, lane, A);
            l1_group(W, 1, old1, lane, A);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int list = side * N + v;
                const int lo = uni(G.off, list), hi = uni(G.off, list + 1);
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
                for (int p0 = lo; p0 < hi; p0 += 4) {   // four edges' tiles in flight, added in list order
                    f32x4 t0[4], t1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* tile = edges + uni(G.items, p0 + u < hi ? p0 + u : p0) * kTile;
                        t0[u] = lds4(tile + lane * 4);
                        t1[u] = lds4(tile + 256 + lane * 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (p0 + u < hi) {
                            s0 += t0[u];
                            s1 += t1[u];
                        }
                }
                const float sc = __int_as_float(uni(G.scale, list));
                s0 *= sc;
                s1 *= sc;
                if (TRAIN && P.agg) {
                    const RowOut o = row_out(P.agg + (int64_t)list * ldb + b0, (int64_t)2 * N * ldb);
                    put_rows(o, 0, s0, g, n, live);
                    put_rows(o, 1, s1, g, n, live);
                }
                l1_group(W, 2 + 2 * side, s0, lane, A);
                l1_group(W, 3 + 2 * side, s1, lane, A);
            }
            f32x4 y[2];
            finish_mlp<TRAIN>(W, lane, 0, A, hist(M.H1, v), hist(M.H2, v), live, y);
            if (TRAIN) {
                const RowOut o = rows(M.Y, M, v);
                put_rows(o, 0, y[0], g, n, live);
                put_rows(o, 1, y[1], g, n, live);
            }
            y[0] += old0;
            y[1] += old1;
            put_tile(nodes + v * kTile, y);
            if (TRAIN) {
                const RowOut os = rows(M.Ysum, M, v);
                put_rows(os, 0, y[0], g, n, live);
                put_rows(os, 1, y[1], g, n, live);
            }
        }
    }
    __syncthreads();
    GNN_STAMP(4);
    // ---- edge update [edge | updated source | updated target] (:1322-1340) and, on the sum still in registers, the output MLP
    // (:1367-1392) - for the live edges only (a demand edge's update feeds nothing, csrc/mlp3.hip)
    float* outl = scratch + kEnvScratch + P.tab_words;   // [n_live][16] desired quantities
    {
        const MlpL W = mlp_at(wb0, 6, 2), WO = mlp_at(wb0 + sz_eu, 2, 1);
        const NicGnnPeriodMlp &M = P.mlp[3], &MO = P.mlp[4];
        for (int e = wave; e < P.n_live; e += kWaves) {
            const int s_ = uni(G.src, e), t_ = uni(G.tgt, e);
            Acc2 A = l1_begin(W, g);
            const f32x4 old0 = lds4(edges + e * kTile + lane * 4), old1 = lds4(edges + e * kTile + 256 + lane * 4);
            l1_group(W, 0, old0, lane, A);
            l1_group(W, 1, old1, lane, A);
            if (s_ >= 0) l1_tile(W, 2, nodes + s_ * kTile, lane, A);
            if (t_ >= 0) l1_tile(W, 4, nodes + t_ * kTile, lane, A);
            f32x4 y[2];
            finish_mlp<TRAIN>(W, lane, 0, A, hist(M.H1, e), hist(M.H2, e), live, y);
            if (TRAIN) {
                const RowOut o = rows(M.Y, M, e);
                put_rows(o, 0, y[0], g, n, live);
                put_rows(o, 1, y[1], g, n, live);
            }
            y[0] += old0;
            y[1] += old1;
            if (TRAIN) {
                const RowOut os = rows(M.Ysum, M, e);
                put_rows(os, 0, y[0], g, n, live);
                put_rows(os, 1, y[1], g, n, live);
            }
            Acc2 B = l1_begin(WO, g);
            l1_group(WO, 0, y[0], lane, B);
            l1_group(WO, 1, y[1], lane, B);
            f32x4 z[2];
            finish_mlp<TRAIN>(WO, lane, 1, B, hist(MO.H1, e), hist(MO.H2, e), live, z);
            if (g == 0) {
                outl[e * NB + n] = z[0][0];
                if (live && (TRAIN || !P.fuse_env)) MO.Y[(int64_t)e * ldb + b0 + n] = z[0][0];
            }
        }
    }
    GNN_STAMP(5);
    if (!P.fuse_env) return;
    __syncthreads();
    // ---- proportional allocation + one period of dynamics (csrc/gnn_alloc_env.hip's bodies, on 16 scenarios x 4 quad lanes =
    // wavefront 0).  The desired quantities are read from LDS (the head's `out` with a row stride of 16), the orders go through
    // global memory and are read by the other lanes of the quad behind a barrier.
    {
        const NicEnvStepIO& io = P.io;
        const int x = n, q = g, Wn = io.dims.n_warehouses;
        const int64_t b = b0 + x;
        const bool on = wave == 0, act = on && live;
        float(*part)[4][NB] = reinterpret_cast<float(*)[4][NB]>(scratch);
        float(*rq)[NB] = reinterpret_cast<float(*)[NB]>(scratch + kEnvChunk * 4 * NB);
        float(*cw)[NB] = reinterpret_cast<float(*)[NB]>(scratch + (kEnvChunk + 1) * 4 * NB);
        if (act && q == 0)
            nic::gnn_alloc_fwd_one(outl, NB, x, io.wh_inv, P.orders, P.sums, P.ratio, P.scale, io.dims.n_stores, P.e_self, P.e_supplier,
                                   P.cap_at_one, b, ldb);
        __syncthreads();
        if (on) rq[q][x] = act ? nic::env_fwd_stores<MAXW>(io, P.store_out, b, q) : 0.f;
        for (int wc = 0; wc < Wn; wc += kEnvChunk) {
            if (on)
                for (int i = 0; i < kEnvChunk && wc + i < Wn; ++i) part[i][q][x] = act ? nic::env_ship_partial(io, wc + i, b, q) : 0.f;
            __syncthreads();
            if (on)
                for (int i = q; i < kEnvChunk && wc + i < Wn; i += nic::kQuad) {
                    const float shipped = nic::combine4(part[i][0][x], part[i][1][x], part[i][2][x], part[i][3][x]);
                    cw[wc + i][x] = act ? nic::env_fwd_warehouse<MAXW>(io, P.wh_out, wc + i, shipped, b) : 0.f;
                }
            __syncthreads();
        }
        if (act && q == 0) {
            const float total = nic::combine4(rq[0][x], rq[1][x], rq[2][x], rq[3][x]);
            float r_wh = 0.f;
            for (int w = 0; w < Wn; ++w) r_wh += cw[w][x];
            P.reward[b] = total + r_wh;
        }
    }
    GNN_STAMP(6);
}

int lds_floats(const NicGnnPeriod& p, int* wb0, int* wb1, bool spill = false) {
    const int s1q_in = (p.Dn + 15) / 16;
    const int a = pack_floats(s1q_in, 2) + pack_floats(5, 2), b = pack_floats(6, 2) + pack_floats(2, 1);
    *wb0 = a > b ? a : b;
    *wb1 = pack_floats(6, 2);
    const int tab = (graph_words(p.n_nodes, p.n_edges, p.n_agg_items) + 3) / 4 * 4;
    return (p.n_nodes + (spill ? 0 : p.n_edges)) * kTile + *wb0 + *wb1 + kEnvScratch + tab + p.n_live * NB;
}
constexpr int kLdsLimit = 160 * 1024;

int check(const NicGnnPeriod* p, const char* who) {
    NIC_REQUIRE(p != nullptr, "%s: descriptor is null", who);
    NIC_REQUIRE(p->n_nodes > 0 && p->n_edges > 0 && p->n_live > 0 && p->n_live <= p->n_edges && p->n_agg_items >= 0 &&
                    p->n_agg_items <= 4 * p->n_edges, "%s: bad graph sizes", who);
    NIC_REQUIRE(p->n_scenarios > 0 && p->ldb >= p->n_scenarios && p->ldb % 32 == 0, "%s: bad scenario sizes (ldb a multiple of 32)", who);
    NIC_REQUIRE(p->Dn > 0 && p->Dn <= 64 && p->max_inv >= 0 && p->max_inv <= p->Dn, "%s: node feature rows outside [1,64]", who);
    NIC_REQUIRE(p->src && p->tgt && p->agg_off && p->agg_items && p->agg_scale && p->lead && p->node_row0 && p->node_slots && p->state &&
                    p->feat, "%s: null graph / feature buffer", who);
    for (int i = 0; i < 5; ++i) {
        const NicGnnPeriodMlp& m = p->mlp[i];
        NIC_REQUIRE(m.wpk != nullptr, "%s: MLP %d has no packed weights", who, i);
        NIC_REQUIRE((m.H1 == nullptr) == (m.H2 == nullptr), "%s: MLP %d keeps one of H1 / H2 only", who, i);
        NIC_REQUIRE(m.Y == nullptr || m.row_stride > 0, "%s: MLP %d: row stride of Y", who, i);
    }
    NIC_REQUIRE(p->mlp[4].Y != nullptr, "%s: the output MLP's desired quantities [n_edges][ldb] are always written", who);
    int wb0, wb1;
    const int need = lds_floats(*p, &wb0, &wb1) * 4;
    if (need > kLdsLimit) {
        const int need2 = lds_floats(*p, &wb0, &wb1, true) * 4;
        NIC_REQUIRE(need2 <= kLdsLimit, "%s: the graph's node embeddings and tables do not fit in LDS (%d bytes of %d); use the per-MLP launches",
                    who, need2, kLdsLimit);
        NIC_REQUIRE(p->edge_scratch != nullptr, "%s: the graph's embeddings do not fit in LDS (%d bytes of %d) and no edge_scratch was given",
                    who, need, kLdsLimit);
    }
    if (p->fuse_env) {
        const NicEnvDims& d = p->io.dims;
        NIC_REQUIRE(d.n_warehouses == 1 && d.n_echelons == 0, "%s: fused allocation + env step: one supplying warehouse, no extra echelons", who);
        NIC_REQUIRE(d.n_scenarios == p->n_scenarios && d.ldb == p->ldb, "%s: env dims differ from the policy's", who);
        NIC_REQUIRE(p->orders && p->sums && p->ratio && p->scale && p->store_out && p->wh_out && p->reward && p->e_supplier >= 0,
                    "%s: null allocation / env buffer", who);
        NIC_REQUIRE(p->io.store_orders.p == p->orders && p->io.wh_orders.p == p->orders + (int64_t)d.n_stores * d.ldb,
                    "%s: io's order tables must be the rows of `orders` ([S + 1][ldb])", who);
    }
    return 0;
}
}  // namespace

extern "C" {

int nic_gnn_period_pack_size(int32_t K_steps_groups, int32_t n_out) { return pack_floats(K_steps_groups, n_out > 16 ? 2 : 1); }

int nic_gnn_period_ok(int32_t n_nodes, int32_t n_edges, int32_t Dn) {
    NicGnnPeriod p{};
    p.n_nodes = n_nodes;
    p.n_edges = n_edges;
    p.n_live = n_edges;
    p.n_agg_items = 2 * n_edges;
    p.Dn = Dn;
    int a, b;
    if (!(n_nodes > 0 && n_edges > 0 && Dn > 0 && Dn <= 64)) return 0;
    if (lds_floats(p, &a, &b) * 4 <= kLdsLimit) return 1;
    return lds_floats(p, &a, &b, true) * 4 <= kLdsLimit ? 2 : 0;   // 2: with the edge tiles in NicGnnPeriod.edge_scratch
}


int nic_gnn_period_fwd(const NicGnnPeriod* p, void* stream) {
    if (int e = check(p, "nic_gnn_period_fwd")) return e;
    NicGnnPeriod q = *p;
    const bool spill = lds_floats(q, &q.wb0_floats, &q.wb1_floats) * 4 > kLdsLimit;
    const int bytes = lds_floats(q, &q.wb0_floats, &q.wb1_floats, spill) * 4;
    q.tab_words = (graph_words(q.n_nodes, q.n_edges, q.n_agg_items) + 3) / 4 * 4;
    const bool train = q.mlp[0].Y != nullptr;
    const dim3 grid(nic::ceil_div(q.n_scenarios, NB));
    hipStream_t s = nic::as_stream(stream);
    int m = 4;
    if (q.fuse_env) {
        const int sl = q.io.dims.store_slots > q.io.dims.warehouse_slots ? q.io.dims.store_slots : q.io.dims.warehouse_slots;
        m = sl <= 4 ? 4 : (sl <= 8 ? 8 : NIC_MAX_SLOTS);
    }
    nic::note_kernelf("gnn_period_fwd_kernel<%d,%s,%d%s>", m, train ? "true" : "false", m == 4 ? 16 : 8, spill ? ",spill" : "");
#define NIC_GP_FWD(MW, TR, SP)                                                                                              \
    do {                                                                                                                    \
        constexpr int NW = (MW) == 4 ? 16 : 8;                                                                              \
        /* (per launch: the attribute belongs to the CURRENT device's copy of the kernel) */                                \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gnn_period_fwd_kernel<MW, TR, NW, SP>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit) != hipSuccess)                       \
            return nic::fail("nic_gnn_period_fwd: cannot raise the dynamic LDS limit");                                     \
        hipLaunchKernelGGL((gnn_period_fwd_kernel<MW, TR, NW, SP>), grid, dim3(NW * 64), bytes, s, q);                      \
    } while (0)
#define NIC_GP_FWD2(MW)                          \
    do {                                         \
        if (train && spill) NIC_GP_FWD(MW, true, true);        \
        else if (train) NIC_GP_FWD(MW, true, false);           \
        else if (spill) NIC_GP_FWD(MW, false, true);           \
        else NIC_GP_FWD(MW, false, false);                     \
    } while (0)
    if (m == 4) NIC_GP_FWD2(4);
    else if (m == 8) NIC_GP_FWD2(8);
    else NIC_GP_FWD2(NIC_MAX_SLOTS);
#undef NIC_GP_FWD2
#undef NIC_GP_FWD
    return nic::check_launch("nic_gnn_period_fwd");
}

#ifdef NIC_TUNING_BUILD
int nic_tuning_set_gnn_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gnn_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif
}
