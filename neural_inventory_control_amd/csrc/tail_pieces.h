// Device pieces shared by the fused per-period tail kernels (period_tail.hip) and the whole-horizon kernels of the wide
// vanilla_warehouse policy (wide_rollout.hip): 32-scenario LDS tiles, the block's NicEnvStepIO over them, the static-table
// staging, and the contraction split over a workgroup's four wavefronts.  Included inside each translation unit's anonymous
// namespace users (everything here is `static` by construction: __device__ __forceinline__ / constexpr).
#pragma once
#include <type_traits>

#include "env_step_body.h"
#include "nic_common.h"
#include "policy_heads_body.h"


namespace {


using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int NB = 32;        // scenarios per block
constexpr int LDT = 36;       // LDS tile row stride (floats): 16-byte aligned rows, conflict-free b128 reads down a column of rows
constexpr int kThreads = 256;
constexpr int kChunk = 8;     // warehouses whose shipment partials are exchanged per barrier round (as head_env.hip)
constexpr int kMaxWh = 16;    // (S + 1) Wn <= 32 rows of logits
constexpr int kStateRows = 52;  // F + 1 <= 52 (thin_in_fwd's 26 steps of two rows)
constexpr int kTabRows = 68;    // static tables of a block: 2 S + S Wn + 3 Wn rows (<= 66 for S <= 16, (S + 1) Wn <= 32)

// In-kernel timestamps (tuning build only: tools/tail_stamp_probe.py compiles its own copy of the library with -DNIC_TUNING_BUILD;
// the product library contains none of this): the 100 MHz wall clock of workgroup 0's four wavefronts at up to 16 points.
#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_tail_stamps = nullptr;
#define TAIL_STAMP(point)                                                                         \
    do {                                                                                          \
        if (g_tail_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0)               \
            g_tail_stamps[(threadIdx.x >> 6) * 16 + (point)] = wall_clock64();                    \
    } while (0)
#else
#define TAIL_STAMP(point) do { } while (0)
#endif

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float elu_f(float x) {   // (csrc/thin_layer.hip::thin_elu = csrc/linear_mfma.hip::elu_f)
    const float xn = fminf(x, 0.f);
    const float series =
        xn * fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float viaexp = __expf(xn) - 1.f;
    const float neg = xn > -0.35f ? series : viaexp;
    return x > 0.f ? x : neg;
}
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int64_t n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}


// the block's NicEnvStepIO: state / demand / orders AND the static cost / lead-time tables in LDS tiles.  (The first version
// left the static tables in HBM / L2, moved to the block's first scenario: every store of the env step then began with a round
// trip to L2 - 2.3 us of a 19 us forward launch, 4.1 us of a 30 us backward launch, tools/tail_stamp_probe.py.)
// Table tile rows: [underage S | holding S | lead times S x Wn | warehouse holding Wn | warehouse lead Wn | edge cost Wn].
__device__ __forceinline__ NicEnvStepIO block_io(const NicEnvStepIO& g, int nlive, float* st, float* dm, float* od, float* tb) {
    NicEnvStepIO io = g;
    const int S = g.dims.n_stores, Wn = g.dims.n_warehouses;
    io.dims.n_scenarios = nlive;
    io.dims.ldb = LDT;
    io.store_inv = st;
    io.wh_inv = st + S * g.dims.store_slots * LDT;
    io.ech_inv = nullptr;
    io.demand = NicTable2{dm, LDT, 1};
    io.store_orders = NicTable3{od, (int64_t)Wn * LDT, LDT, 1};
    io.wh_orders = NicTable2{od + S * Wn * LDT, LDT, 1};
    io.underage = NicTable2{tb, LDT, 1};
    io.holding = NicTable2{tb + S * LDT, LDT, 1};
    io.lead_times = NicTable3{tb + 2 * S * LDT, (int64_t)Wn * LDT, LDT, 1};
    const int r3 = 2 * S + S * Wn;
    io.wh_holding = NicTable2{tb + r3 * LDT, LDT, 1};
    io.wh_lead_times = NicTable2{tb + (r3 + Wn) * LDT, LDT, 1};
    if (g.wh_edge_costs.p) io.wh_edge_costs = NicTable2{tb + (r3 + 2 * Wn) * LDT, LDT, 1};
    return io;
}

// the block's static tables -> registers (thread t: column t % 32, rows t / 32 + 8 i of each table), branch-free: every load is
// unconditional through a clamped row index, so a thread's fourteen loads issue back to back (the first version selected one
// of six pointers per element and hipcc turned the selects into 54 branches with a wait behind each).  Dead columns shadow the
// last live scenario.  Table tile rows: see block_io.
struct TabRegs {
    float u[2], h[2], l[4], wh[2], wl[2], we[2];   // <= 16 stores, <= 32 (store, warehouse) pairs, <= 16 warehouses
};
__device__ __forceinline__ void tables_fetch(const NicEnvStepIO& g, int c0, int nlive, TabRegs& v) {
    const int S = g.dims.n_stores, Wn = g.dims.n_warehouses;
    const int t = threadIdx.x, col = t & 31;
    int row0 = t >> 5;
    asm volatile("" : "+v"(row0));   // (opaque: the addresses below are recomputed per block, not hoisted out of the caller's loops)
    const int64_t b = c0 + (col < nlive ? col : nlive - 1);
    const NicTable2& we = g.wh_edge_costs.p ? g.wh_edge_costs : g.wh_holding;   // (no edge costs: any valid table, never stored)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rs = min(row0 + 8 * i, S - 1), rw = min(row0 + 8 * i, Wn - 1);
        v.u[i] = g.underage.p[rs * g.underage.loc_stride + b * g.underage.scn_stride];
        v.h[i] = g.holding.p[rs * g.holding.loc_stride + b * g.holding.scn_stride];
        v.wh[i] = g.wh_holding.p[rw * g.wh_holding.loc_stride + b * g.wh_holding.scn_stride];
        v.wl[i] = g.wh_lead_times.p[rw * g.wh_lead_times.loc_stride + b * g.wh_lead_times.scn_stride];
        v.we[i] = we.p[rw * we.loc_stride + b * we.scn_stride];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(row0 + 8 * i, S * Wn - 1), ls = r / Wn, lw = r - ls * Wn;
        v.l[i] = g.lead_times.p[ls * g.lead_times.loc_stride + lw * g.lead_times.sup_stride + b * g.lead_times.scn_stride];
    }
}
__device__ __forceinline__ void tables_put(float* tb, const NicEnvStepIO& g, const TabRegs& v) {
    const int S = g.dims.n_stores, Wn = g.dims.n_warehouses;
    const int t = threadIdx.x, col = t & 31, row0 = t >> 5;
    const int r2 = 2 * S, r3 = r2 + S * Wn;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = row0 + 8 * i;
        if (r < S) {
            tb[r * LDT + col] = v.u[i];
            tb[(S + r) * LDT + col] = v.h[i];
        }
        if (r < Wn) {
            tb[(r3 + r) * LDT + col] = v.wh[i];
            tb[(r3 + Wn + r) * LDT + col] = v.wl[i];
            tb[(r3 + 2 * Wn + r) * LDT + col] = v.we[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 8 * i;
        if (r < S * Wn) tb[(r2 + r) * LDT + col] = v.l[i];
    }
}

// [rows][32] tile <-> [rows][ldb] block, float4 per lane (thread t: row t / 8 (+ 32 per pass), columns 4 (t % 8) ..)
template <int PASSES>
__device__ __forceinline__ void tile_fetch(const float* g, int64_t ldb, int c0, int rows, float4 (&v)[PASSES]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int r = (t >> 3) + 32 * i;
        v[i] = r < rows ? *reinterpret_cast<const float4*>(g + (int64_t)r * ldb + c0 + (t & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int PASSES>
__device__ __forceinline__ void tile_put(float* tile, int rows, const float4 (&v)[PASSES]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int r = (t >> 3) + 32 * i;
        if (r < rows) *reinterpret_cast<float4*>(tile + r * LDT + (t & 7) * 4) = v[i];
    }
}
// live columns of a tile's rows -> global
__device__ __forceinline__ void tile_store(const float* tile, float* g, int64_t ldb, int c0, int rows, int nlive) {
    const int t = threadIdx.x, c = (t & 7) * 4;
    for (int r = t >> 3; r < rows; r += 32) {
        const float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c);
        float* dst = g + (int64_t)r * ldb + c0 + c;
        if (c + 3 < nlive) {
            *reinterpret_cast<float4*>(dst) = v;
        } else {
            if (c + 0 < nlive) dst[0] = v.x;
            if (c + 1 < nlive) dst[1] = v.y;
            if (c + 2 < nlive) dst[2] = v.z;
        }
    }
}

// ---- contraction of a [32 (x MT)] x K operand with a [K][32 scenarios] block, K split over the four wavefronts -----------------
// gemm_wx_stream_kernel<1, 4>'s loop for one column tile: per 16-deep k group a lane fetches its 8 A values as two 16-byte loads
// (row m0 + lane, k contiguous) and its 8 B values as dword loads, D = 4 groups ahead; within a group lanes 0-31 take k = kk and
// lanes 32-63 k = 8 + kk.  MT row tiles share the B values.  Returns the wavefront's partial accumulators.
template <int MT, int PF = 4>
__device__ __forceinline__ void ksplit_contract(const float* A, int64_t lda, int M, const float* Bm, int64_t ldb, int K, int c0,
                                                f32x16 (&acc)[MT]) {
    constexpr int D = 4, KSP = 4;   // D: the stream kernel's group count granularity (which k range a wavefront owns)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (int64_t)M * lda), rB = make_rsrc(Bm, (int64_t)K * ldb);
    const int ng_all = (K + 15) / 16;
    const int per = ((ng_all + KSP - 1) / KSP + D - 1) / D * D, nblk = per / PF;   // PF groups in flight (PF divides D)
    const int g_lo = wave * per;
    const int ldb4 = (int)ldb * 4;
    int offA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) offA[i] = (int)((((int64_t)(32 * i + li)) * lda + g_lo * 16 + h * 8) * 4);
    int offB = (int)((((int64_t)g_lo * 16 + h * 8) * ldb + c0 + li) * 4);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[PF][MT][8], b[PF][8];
    auto load = [&](int d) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float4 lo = buf_load4(rA, offA[i]), hi = buf_load4(rA, offA[i] + 16);
            a[d][i][0] = lo.x; a[d][i][1] = lo.y; a[d][i][2] = lo.z; a[d][i][3] = lo.w;
            a[d][i][4] = hi.x; a[d][i][5] = hi.y; a[d][i][6] = hi.z; a[d][i][7] = hi.w;
            offA[i] += 64;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) b[d][kk] = ldf(rB, offB, kk * ldb4);
        offB += 16 * ldb4;
    };
    auto compute = [&](int d) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i][kk], b[d][kk], acc[i], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) load(d);
    for (int blk = 0; blk + 1 < nblk; ++blk) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            compute(d);
            load(d);
        }
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) compute(d);
}

// wavefronts 1..3 hand their accumulators to wavefront 0 (call on every wavefront, barrier in between)
template <int MT>
__device__ __forceinline__ void ksplit_publish(float* red, const f32x16 (&acc)[MT]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave - 1) * MT + i) * 16 + r) * 64 + lane] = acc[i][r];
    }
}
template <int MT>
__device__ __forceinline__ void ksplit_collect(const float* red, f32x16 (&acc)[MT]) {   // wavefront 0, after the barrier
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += red[((w * MT + i) * 16 + r) * 64 + lane];
}


}  // namespace
