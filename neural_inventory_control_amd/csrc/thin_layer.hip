// Fused backward of a THIN output layer (N <= 32 output rows: the logits layer of the policy MLPs, e.g. 17 x 512 for
// BASELINE cfg3): one pass over the layer's input activations H = X[K][ldb] produces BOTH
//   dX[k][b]            = elu'(X[k][b]) * sum_n W[n][k] dY[n][b]           (what nic_linear_dgrad computes)
//   slab[split][n][k]  += sum_{b in split} dY[n][b] X[k][b],  [n][K] += sum_b dY[n][b]   (what nic_linear_wgrad computes)
// The two separate GEMMs each stream X (134 MB at 512 x 65,536) and are bound by that stream, not by their flops; here X
// is read from HBM once: 2 x K x 4 bytes per scenario (X in, dX out) is the whole algorithmic traffic.
// Reference: the logits layer of MyNeuralNetwork.forward (neural_networks.py:80-106) under autograd (addmm backward +
// elu_backward of the layer below).
//
// One wavefront (= one workgroup, no barriers across waves) owns a scenario range x KG 32-row chunks of X (KG = 1: two
// chunks per wave measured slower - hipcc hoists every fragment read of the unrolled body and spills); registers stay under
// 256 so that two such waves share a SIMD and hide each other's LDS/HBM latency.  Per 64-scenario block the X tile is
// staged in LDS once and used two ways: as the B operand of the weight-gradient MFMAs (lane = row k, float4 over
// scenarios), and row-wise as elu'(X) for the dX tile.  The next X tile and dY tile are fetched into registers while the
// current block is computed.
//   wgrad  D[n][k]   += sum_b dY[n][b] X[k][b]   : A = dY tile (lane i = n), B = X tile (lane j = k); contraction index
//                                                  of MFMA step (q, e) and lane half h is b = 8q + 4h + e (b128 LDS reads)
//   dgrad  D[k][b]    = sum_n W[n][k] dY[n][b]   : A = W^T chunk (lane i = k, resident in VGPRs), B = dY tile (lane j = b)
#include <stdint.h>

#include "nic_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CB = 64;        // scenarios per block
constexpr int LD = CB + 4;    // LDS row stride (floats): 16-byte aligned rows, conflict-free b128 reads down a column of rows
constexpr int KG = 1;         // 32-row chunks per wavefront

struct ThinParams {
    const float* W;   // [N][ldw]
    int64_t ldw;
    const float* dY;  // [N][ldb]
    const float* X;   // [K][ldb]
    float* dX;        // [K][ldb]
    float* slab;      // [n_splits][N][lds]
    int64_t lds_, ldb;
    int N, K, nB, chunk, n_splits;  // chunk = scenarios per split (multiple of 64)
    int act_prev;
};

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }
// zero the scenarios >= nB of a float4 that starts at scenario `col` (select, no branch: the caller applies it when the value
// is consumed, so the load itself stays an unconditional, early-issued global_load_dwordx4)
__device__ __forceinline__ float4 mask4(float4 v, int col, int nB) {
    v.x = col < nB ? v.x : 0.f;
    v.y = col + 1 < nB ? v.y : 0.f;
    v.z = col + 2 < nB ? v.z : 0.f;
    v.w = col + 3 < nB ? v.w : 0.f;
    return v;
}

// Ordering point for LDS traffic inside a ONE-wave workgroup.  __syncthreads() would also wait for every outstanding global
// load and STORE (s_waitcnt vmcnt(0)), i.e. drain the dX stores of a chunk before the next chunk may start; the LDS queue of
// a wave is in order, so all that is needed is that the compiler keeps LDS writes and reads on their side of this point.
__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// NS = MFMA steps over the N output rows (2 rows per step).  FULL: ldb % 64 == 0 and K % 64 == 0 — every wave has KG whole
// chunks and stores its dX tile unconditionally (the padding scenarios nB..ldb receive exact zeros: their dY is masked), so
// the chunk body is straight-line code and hipcc's s_waitcnt counts stay exact: the wait for a prefetched tile does not
// also drain the dX stores issued after it.  Otherwise every chunk / store is guarded.
template <int NS, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void thin_bwd_kernel(ThinParams p) {
    __shared__ __attribute__((aligned(16))) float lds_dy[2 * NS * LD];
    __shared__ __attribute__((aligned(16))) float lds_x[32 * LD];
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    const int n_kg = (p.K + 32 * KG - 1) / (32 * KG);
    // split fastest: the row groups of one split run on the same XCD (workgroups go round-robin over the 8 XCDs) and share
    // its dY tile in that L2
    const int split = blockIdx.x % p.n_splits, kg = blockIdx.x / p.n_splits;
    const int k_base = kg * 32 * KG;
    const int n_chunks = FULL ? KG : min(KG, (p.K - k_base) / 32);
    // (splits that interleave their 64-scenario blocks, so that concurrent waves read neighbouring pieces of each row,
    // measured 20 % slower than contiguous ranges)
    const int col_lo = split * p.chunk, col_hi = min(col_lo + p.chunk, p.nB);
    if (col_lo >= col_hi) return;  // (the slab rows of an empty split stay as they are)

    // W^T fragments of this wave's row chunks: A[i = k][kk = n], n = 2s + h
    float aW[KG][NS];
#pragma unroll
    for (int c = 0; c < KG; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = 2 * s + h;
            aW[c][s] = (c < n_chunks && n < p.N) ? p.W[(int64_t)n * p.ldw + k_base + c * 32 + li] : 0.f;
        }

    f32x16 wacc[KG];
#pragma unroll
    for (int c = 0; c < KG; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[c][r] = 0.f;
    float bias_acc = 0.f;

    const float* xg = p.X + (int64_t)k_base * p.ldb;
    float* dxg = p.dX + (int64_t)k_base * p.ldb;
    const int ldb32 = (int)p.ldb;
    // staging map of a [32][64] tile: lane -> row (lane >> 4) + 4q, columns (lane & 15) * 4 .. +3
    const int srow = lane >> 4, scol = (lane & 15) * 4;
    // the column is clamped into the row (ldb % 4 == 0), so the loads are unconditional; what lies beyond the last scenario
    // (always a clamped or padding column) is masked when the tile is written to LDS
    // addresses are (uniform base: SGPR arithmetic) + (one per-lane 32-bit offset), so the unrolled body keeps ONE address
    // register per access pattern instead of one per access (K * ldb < 2^31 is checked by the launcher)
    const int lane_ld = srow * ldb32 + scol;
    const int lane_st = 4 * h * ldb32 + li;   // dX stores: row (uniform) + 4h, scenario li  // staging loads: row srow (+ 4q uniform), columns scol..scol+3
    auto load_x = [&](int col0, int c, float4 (&v)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float* rowbase = xg + (c * 32 + 4 * q) * ldb32 + col0;  // uniform
            v[q] = FULL ? *reinterpret_cast<const float4*>(rowbase + lane_ld)
                        : *reinterpret_cast<const float4*>(rowbase + srow * ldb32 + min(scol, ldb32 - 4 - col0));
        }
    };

    constexpr int DQ = (2 * NS + 3) / 4;  // float4 loads per lane that cover the 2 NS staged dY rows
    auto load_dy = [&](int col0, float4 (&v)[DQ]) {
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            if (FULL && 4 * q + 3 < p.N) {  // (uniform) all four rows of this step exist
                v[q] = *reinterpret_cast<const float4*>(p.dY + 4 * q * ldb32 + col0 + lane_ld);
            } else {
                const int n = min(srow + 4 * q, p.N - 1);  // rows N .. 2 NS - 1 are zero padding (masked at the LDS write)
                v[q] = *reinterpret_cast<const float4*>(p.dY + n * ldb32 + min(col0 + scol, ldb32 - 4));
            }
        }
    };
    const bool act_elu = p.act_prev == NIC_ACT_ELU;

    float4 xv[8], dyv[DQ];
    load_dy(col_lo, dyv);
    load_x(col_lo, 0, xv);
    for (int col0 = col_lo; col0 < col_hi; col0 += CB) {
        // ---- dY tile of this block (fetched during the previous block): rows >= N and columns >= nB are zero
        lds_order();  // previous block's reads of lds_dy are done
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int n = srow + 4 * q;
            if (n < 2 * NS) {
                float4 v = mask4(dyv[q], col0 + scol, p.nB);
                if (n >= p.N) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(lds_dy + n * LD + scol) = v;
            }
        }
        lds_order();
        // the bias gradient rides on the first row group: lane (i = n, h) sums its 32 scenarios of the block
        const bool arow = li < 2 * NS;  // dY rows staged in LDS
        if (kg == 0 && arow) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(lds_dy + li * LD + 8 * q + 4 * h);
                bias_acc += (a.x + a.y) + (a.z + a.w);
            }
        }

#pragma unroll
        for (int c = 0; c < KG; ++c) {
            if (!FULL && c >= n_chunks) break;
            // ---- X chunk: registers (fetched one chunk ahead) -> LDS
            lds_order();  // previous chunk's reads of lds_x are done
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(lds_x + (srow + 4 * q) * LD + scol) = mask4(xv[q], col0 + scol, p.nB);
            lds_order();
            {  // fetch the next chunk (or the first chunk and the dY tile of the next block) while this one is computed
                const bool last = FULL ? c + 1 == KG : c + 1 >= n_chunks;
                const int coln = last ? col0 + CB : col0;
                if (coln < col_hi) {
                    load_x(coln, last ? 0 : c + 1, xv);
                    if (last) load_dy(coln, dyv);
                }
            }
            // ---- weight gradient: 32 MFMA steps over the 64 scenarios of the block
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // A operand: lane (i = n, h) holds dY[n][8q + 4h + e] (rows >= 2 NS are zero and not staged)
                const float4 ay = arow ? *reinterpret_cast<const float4*>(lds_dy + li * LD + 8 * q + 4 * h)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 bx = *reinterpret_cast<const float4*>(lds_x + li * LD + 8 * q + 4 * h);
                wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.x, bx.x, wacc[c], 0, 0, 0);
                wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.y, bx.y, wacc[c], 0, 0, 0);
                wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.z, bx.z, wacc[c], 0, 0, 0);
                wacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay.w, bx.w, wacc[c], 0, 0, 0);
            }
            // ---- input gradient of the chunk, two 32-scenario halves
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s)  // B operand: lane (j = scenario, h) holds dY[2s + h][cb*32 + j]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aW[c][s], lds_dy[(2 * s + h) * LD + cb * 32 + li], acc, 0, 0, 0);
                const int col = col0 + cb * 32 + li;
                if (FULL || col < p.nB) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ku = (r & 3) + 8 * (r >> 2);  // uniform part of the row; the lane half adds 4h
                        const float g = elu_grad_from_out(lds_x[(ku + 4 * h) * LD + cb * 32 + li]);
                        float* rowbase = dxg + (c * 32 + ku) * ldb32 + col0 + cb * 32;  // uniform
                        rowbase[lane_st] = acc[r] * (act_elu ? g : 1.f);
                    }
                }
            }
        }
    }

    // ---- slab update (read-modify-write: the slab accumulates over the periods of a rollout).  D layout: lane (j = k, h),
    // register r <-> output row n = crow(r, h)
    float* slab = p.slab + (int64_t)split * p.N * p.lds_;
#pragma unroll
    for (int c = 0; c < KG; ++c) {
        if (!FULL && c >= n_chunks) break;
        // (old values of the 16 rows fetched together, then added and stored: as `slab[..] += acc` per row this was sixteen
        // dependent load - wait - store round trips at the tail of every wavefront)
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = crow(r, h);
            old[r] = slab[(int64_t)(n < p.N ? n : 0) * p.lds_ + k_base + c * 32 + li];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = crow(r, h);
            if (n < p.N) slab[(int64_t)n * p.lds_ + k_base + c * 32 + li] = old[r] + wacc[c][r];
        }
    }
    if (kg == 0) {
        const float total = bias_acc + __shfl_xor(bias_acc, 32);
        if (h == 0 && li < p.N) slab[(int64_t)li * p.lds_ + p.K] += total;
    }
}


// ---- forward of a layer with a SHORT contraction and many output rows (the policy's first layer: 512 x 51 at cfg3) -----------
// Y[N][ldb] = act(W X + b) with K <= 64: the launch is its output write (134 MB at 512 x 65,536 against 13 MB of input), and the
// tiled GEMM - one 256 x 256 tile per CU at a time, 256 VGPRs - spends it in an epilogue that nothing overlaps (56 us = 2.4 TB/s;
// the store pattern alone reaches 5.4 TB/s, tools/layout_probe.hip "first layer").  Here a wavefront owns 32 scenarios for ALL
// output rows: its X tile stays in registers as the B operands (k = 2s + h), the weights stream through as A fragments from
// the TRANSPOSED copy (lane i reads consecutive words: two 128-B lines per load, L1 / L2 hits shared by the CU's wavefronts),
// 32 rows at a time, the next block's fragments in flight while this block's MFMAs run; ~130 VGPRs, so several wavefronts per
// SIMD cover each other's stores and ELUs.  Same contraction order and the same ELU as the GEMM epilogue (elu_f): same bits.
__device__ __forceinline__ float thin_elu(float x) {
    const float xn = fminf(x, 0.f);
    const float series =
        xn * fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float viaexp = __expf(xn) - 1.f;
    const float neg = xn > -0.35f ? series : viaexp;
    return x > 0.f ? x : neg;
}
constexpr int kThinInWaves = 4;
#ifndef NIC_THIN_IN_SPLIT
#define NIC_THIN_IN_SPLIT 2
#endif
constexpr int kThinInSplit = NIC_THIN_IN_SPLIT;   // wavefronts that share a scenario chunk, each taking a slice of the output rows
template <int KS, bool BIAS>
__global__ __launch_bounds__(64 * kThinInWaves) __attribute__((amdgpu_waves_per_eu(4, 4))) void thin_in_fwd_kernel(const float* __restrict__ Wt, int64_t ldwt,
                                                                        const float* __restrict__ bias, const float* __restrict__ X,
                                                                        float* __restrict__ Y, int N, int K, int n_cols, int64_t ldb,
                                                                        int act) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, i = j;
    const int chunk = blockIdx.x * kThinInWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int64_t)chunk * 32 >= n_cols) return;
    const int64_t col_raw = (int64_t)chunk * 32 + j;
    const bool live = col_raw < n_cols;
    const int64_t col = live ? col_raw : 0;
    const bool full = (int64_t)(chunk + 1) * 32 <= n_cols;
    // every operand through raw buffer descriptors: one 32-bit lane offset per buffer, rows as scalar offsets, and the rows past
    // K (an odd K, or K below the template's step count) simply lie beyond the descriptor's extent and read as zero
    auto rsrc_n = [](const float* p, int64_t n_floats) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
    };
    auto ldf = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    const int ld4 = (int)ldb * 4, lw4 = (int)ldwt * 4;
    const __amdgpu_buffer_rsrc_t rX = rsrc_n(X, (int64_t)K * ldb), rW = rsrc_n(Wt, (int64_t)K * ldwt),
                                 rB = rsrc_n(bias ? bias : Wt, bias ? N : 0),
                                 rY = __builtin_amdgcn_make_buffer_rsrc(Y + (int64_t)chunk * 32, 0, 0x7fffffff, 0x00020000);
    float x[KS];
    const int vx = h * ld4 + (int)col * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) x[s] = ldf(rX, vx, 2 * s * ld4);
    const int vo = 4 * h * ld4 + j * 4;
    // this wavefront's slice of the 32-row blocks (blockIdx.y)
    const int all_blocks = N / 32, n_split = (int)gridDim.y, per = (all_blocks + n_split - 1) / n_split;
    const int nb_lo = (int)blockIdx.y * per, n_blocks = (nb_lo + per < all_blocks ? nb_lo + per : all_blocks);
    if (nb_lo >= n_blocks) return;
    auto load_block = [&](int nb, float (&a)[KS]) {
        const int vw = h * lw4 + (nb * 32 + i) * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = ldf(rW, vw, 2 * s * lw4);
    };
    // a block's 26 MFMAs; the register each step has just consumed is refilled with the NEXT block's fragment (one resident
    // fragment set instead of two: 26 registers less, and every load is issued in front of this block's stores)
    auto mfma_block = [&](float (&a)[KS], int nb_next, f32x16& acc) {
        const int vw = h * lw4 + (nb_next * 32 + i) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x[s], acc, 0, 0, 0);
            a[s] = ldf(rW, vw, 2 * s * lw4);
        }
    };
    // bias + activation.  (The rollout engine passes NO bias: its first layer's bias rides in the contraction - the layer input
    // carries a row of ones behind its K rows and the transposed weights the bias as row K, rollout.py - so that a block issues no
    // load that is younger than the previous block's stores.)
    auto finish_block = [&](int nb, f32x16& y) {
        if constexpr (BIAS) {
            const int vb = (nb * 32 + 4 * h) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] += ldf(rB, vb + ((r & 3) + 8 * (r >> 2)) * 4, 0);
        }
        if (act == NIC_ACT_ELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y[r] = thin_elu(y[r]);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four at a time: sixteen ELUs in flight spill registers)
            }
        }
    };
    auto store_block = [&](int nb, const f32x16& y) {
        if (full || live) {   // (`full` is wave-uniform: no lane masks on the common path)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[r]), rY, vo, (nb * 32 + (r & 3) + 8 * (r >> 2)) * ld4, 0);
        }
    };
    // Order of a block's work (round 4).  vmcnt counts loads AND stores on this part and retires them in order, so waiting for a
    // load also waits for every store issued before it: with the next block's fragment loads issued behind this block's stores
    // (round 2's loop, two fragment sets) each block's first MFMA waited for the PREVIOUS block's 16 stores to be acknowledged -
    // a wavefront's MFMA phase and store phase never overlapped (52 us = 22 us of MFMA + 27 us of HBM, added).  Now:
    //     MFMAs(b), each followed by the load of block b+1's fragment into the register it consumed -> activation
    //     -> s_waitcnt vmcnt(0) [stores(b-1): a block old; fragments(b+1): requested during the MFMA chain] -> stores(b)
    // so the stores of a block retire under the next block's MFMAs and nothing waits for a store that was just issued.
    float a[KS];
    f32x16 y;
    load_block(nb_lo, a);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (nothing pending at the loop head: the compiler then adds no waits of its own inside)
    for (int nb = nb_lo; nb < n_blocks; ++nb) {
        mfma_block(a, nb + 1 < n_blocks ? nb + 1 : nb, y);   // (the last block re-reads its own fragments: no branch in the chain)
        finish_block(nb, y);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the previous block's stores and the fragments just requested
        store_block(nb, y);
    }
}

}  // namespace

extern "C" {

int nic_linear_fwd_thin_in_ok(int32_t N, int32_t K) { return N >= 128 && N % 32 == 0 && K >= 1 && K <= 52; }

int nic_linear_fwd_thin_in(const float* Wt, int64_t ldwt, const float* bias, const float* X, float* Y, int32_t N, int32_t K,
                           int32_t n_scenarios, int32_t ldb, int32_t act, void* stream) {
    NIC_REQUIRE(Wt && X && Y, "nic_linear_fwd_thin_in: null buffer");
    NIC_REQUIRE(nic_linear_fwd_thin_in_ok(N, K), "nic_linear_fwd_thin_in: N (%d) must be a multiple of 32 >= 128 and K (%d) <= 52", N, K);
    NIC_REQUIRE(ldwt >= N, "nic_linear_fwd_thin_in: ldwt (%lld) < N", (long long)ldwt);
    NIC_REQUIRE(act == NIC_ACT_NONE || act == NIC_ACT_ELU, "nic_linear_fwd_thin_in: unknown activation %d", act);
    NIC_REQUIRE(n_scenarios > 0 && ldb >= n_scenarios && ldb % 4 == 0,
                "nic_linear_fwd_thin_in: ldb (%d) must be a multiple of 4 and >= n_scenarios (%d)", ldb, n_scenarios);
    NIC_REQUIRE((int64_t)N * ldb * 4 < (1ll << 31), "nic_linear_fwd_thin_in: the output must span less than 2 GiB");
    const int n_cols = (n_scenarios + 3) / 4 * 4;   // like nic_linear_fwd: the scenario count rounded up to 4 columns is written
    // wavefronts that share a 32-scenario chunk, each taking a slice of the 32-row output blocks: two at BASELINE's sizes (4,096
    // wavefronts at 65,536 scenarios); with few scenarios the rows are split further so that the launch still has ~4,096
    // wavefronts and a wavefront's serial chain of row blocks gets shorter (8,192 scenarios: 19 -> 9 us, round 4)
    const int chunks = nic::ceil_div(n_cols, 32);
    int split = (16 * nic::cu_count() + chunks - 1) / chunks;
    if (split < kThinInSplit) split = kThinInSplit;
    if (split > N / 32) split = N / 32;
    const dim3 grid(nic::ceil_div(n_cols, 32 * kThinInWaves), split), block(64 * kThinInWaves);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernel("thin_in_fwd_kernel<26>");
    if (bias != nullptr) hipLaunchKernelGGL((thin_in_fwd_kernel<26, true>), grid, block, 0, s, Wt, ldwt, bias, X, Y, N, K, n_cols, (int64_t)ldb, act);
    else hipLaunchKernelGGL((thin_in_fwd_kernel<26, false>), grid, block, 0, s, Wt, ldwt, bias, X, Y, N, K, n_cols, (int64_t)ldb, act);
    return nic::check_launch("nic_linear_fwd_thin_in");
}

int nic_linear_bwd_thin(const float* W, int64_t ldw, const float* dY, const float* X, float* dX, float* slab, int64_t lds_,
                        int32_t N, int32_t K, int32_t n_scenarios, int32_t ldb, int32_t act_prev, int32_t n_splits,
                        void* stream) {
    NIC_REQUIRE(W && dY && X && dX && slab, "nic_linear_bwd_thin: null buffer");
    NIC_REQUIRE(N >= 1 && N <= NIC_THIN_MAX_ROWS, "nic_linear_bwd_thin: N (%d) must be 1..%d", N, NIC_THIN_MAX_ROWS);
    NIC_REQUIRE(K >= 32 && K % 32 == 0, "nic_linear_bwd_thin: K (%d) must be a positive multiple of 32", K);
    NIC_REQUIRE(ldw >= K && lds_ >= K + 1, "nic_linear_bwd_thin: bad ldw/lds (%lld/%lld)", (long long)ldw, (long long)lds_);
    NIC_REQUIRE(n_scenarios > 0 && ldb >= n_scenarios && ldb % 4 == 0,
                "nic_linear_bwd_thin: ldb (%d) must be a multiple of 4 and >= n_scenarios (%d)", ldb, n_scenarios);
    NIC_REQUIRE(((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0,
                "nic_linear_bwd_thin: dY/X must be 16-byte aligned");
    NIC_REQUIRE(n_splits >= 1, "nic_linear_bwd_thin: n_splits must be >= 1");
    NIC_REQUIRE((int64_t)K * ldb < (1ll << 31), "nic_linear_bwd_thin: K * ldb must be below 2^31 elements");
    NIC_REQUIRE(act_prev == NIC_ACT_NONE || act_prev == NIC_ACT_ELU, "nic_linear_bwd_thin: unknown activation %d", act_prev);
    int chunk = (n_scenarios + n_splits - 1) / n_splits;
    chunk = (chunk + CB - 1) / CB * CB;
    ThinParams p{W, ldw, dY, X, dX, slab, lds_, ldb, N, K, n_scenarios, chunk, n_splits, act_prev};
    const int n_kg = (K + 32 * KG - 1) / (32 * KG);
    const dim3 grid(n_kg * n_splits), block(64);
    hipStream_t s = nic::as_stream(stream);
    const int steps = (N + 1) / 2;
    const bool full = ldb % CB == 0 && K % (32 * KG) == 0;
    nic::note_kernelf("thin_bwd_kernel<%d,%s>", steps <= 4 ? 4 : (steps <= 9 ? 9 : 16), full ? "true" : "false");
#define NIC_THIN_LAUNCH(NS_)                                                                    \
    do {                                                                                        \
        if (full) hipLaunchKernelGGL((thin_bwd_kernel<NS_, true>), grid, block, 0, s, p);       \
        else hipLaunchKernelGGL((thin_bwd_kernel<NS_, false>), grid, block, 0, s, p);           \
    } while (0)
    if (steps <= 4) NIC_THIN_LAUNCH(4);
    else if (steps <= 9) NIC_THIN_LAUNCH(9);
    else NIC_THIN_LAUNCH(16);
#undef NIC_THIN_LAUNCH
    return nic::check_launch("nic_linear_bwd_thin");
}

}  // extern "C"
