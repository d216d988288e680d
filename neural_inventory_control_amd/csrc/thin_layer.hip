// Fused backward of a THIN output layer (N <= 32 output rows: the logits layer of the policy MLPs, e.g. 17 x 512 for
// BASELINE cfg3): one pass over the layer's input activations H = X[K][ldb] produces BOTH
//   dX[k][b]            = elu'(X[k][b]) * sum_n W[n][k] dY[n][b]           (what nic_linear_dgrad computes)
//   slab[split][n][k]  += sum_{b in split} dY[n][b] X[k][b],  [n][K] += sum_b dY[n][b]   (what nic_linear_wgrad computes)
// The two separate GEMMs each stream X (134 MB at 512 x 65,536) and are bound by that stream, not by their flops; here X
// is read from HBM once: 2 x K x 4 bytes per scenario (X in, dX out) is the whole algorithmic traffic.
// Reference: the logits layer of MyNeuralNetwork.forward (neural_networks.py:80-106) under autograd (addmm backward +
// elu_backward of the layer below).
//
// One wavefront (= one workgroup, no barriers across waves) owns a scenario range x a group of KG row chunks (64 rows of
// X); registers are kept under 256 so that two such waves share a SIMD and hide each other's LDS/HBM latency.  Per 64-scenario block and 32-row chunk the X tile is staged in LDS once and used three ways: as the B operand of
// the weight-gradient MFMAs (lane = row k, float4 over scenarios), and row-wise as elu'(X) for the dX tile.
//   wgrad  D[n][k]   += sum_b dY[n][b] X[k][b]   : A = dY tile (lane i = n), B = X tile (lane j = k); contraction index
//                                                  of MFMA step (q, e) and lane half h is b = 8q + 4h + e (b128 LDS reads)
//   dgrad  D[k][b]    = sum_n W[n][k] dY[n][b]   : A = W^T chunk (lane i = k, resident in VGPRs), B = dY tile (lane j = b)
#include <stdint.h>

#include "nic_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CB = 64;        // scenarios per block
constexpr int LD = CB + 4;    // LDS row stride (floats): 16-byte aligned rows, conflict-free b128 reads down a column of rows
constexpr int KG = 2;         // 32-row chunks per wavefront

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
// 4 scenarios col .. col+3 of one row, zero beyond the last scenario (rows are 16-byte aligned and ldb % 4 == 0, so the
// load itself never leaves the row)
__device__ __forceinline__ float4 load4_masked(const float* row, int col, int nB) {
    if (col >= nB) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = *reinterpret_cast<const float4*>(row + col);
    if (col + 3 >= nB) {
        if (col + 1 >= nB) v.y = 0.f;
        if (col + 2 >= nB) v.z = 0.f;
        v.w = 0.f;
    }
    return v;
}

template <int NS>  // MFMA steps over the N output rows (2 rows per step)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void thin_bwd_kernel(ThinParams p) {
    __shared__ __attribute__((aligned(16))) float lds_dy[2 * NS * LD];
    __shared__ __attribute__((aligned(16))) float lds_x[32 * LD];
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    const int n_kg = (p.K + 32 * KG - 1) / (32 * KG);
    // split fastest: the row groups of one split run on the same XCD (workgroups go round-robin over the 8 XCDs) and share
    // its dY tile in that L2
    const int split = blockIdx.x % p.n_splits, kg = blockIdx.x / p.n_splits;
    const int k_base = kg * 32 * KG;
    const int n_chunks = min(KG, (p.K - k_base) / 32);
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
    auto load_x = [&](int col0, int c, float4 (&v)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // 32-bit element offsets from one uniform base (K * ldb < 2^31 is checked by the launcher): one VGPR per address
            v[q] = (c < n_chunks) ? load4_masked(xg + (c * 32 + srow + 4 * q) * ldb32, col0 + scol, p.nB)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float4 xv[8];
    load_x(col_lo, 0, xv);
    for (int col0 = col_lo; col0 < col_hi; col0 += CB) {
        // ---- dY tile of this block: rows >= N and columns >= nB are zero
        __syncthreads();  // previous block's reads of lds_dy are done
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = srow + 4 * q;
            if (n < 2 * NS) {
                const float4 v = (n < p.N) ? load4_masked(p.dY + n * ldb32, col0 + scol, p.nB) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(lds_dy + n * LD + scol) = v;
            }
        }
        __syncthreads();
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
            if (c >= n_chunks) break;
            // ---- X chunk: registers (fetched one chunk ahead) -> LDS
            __syncthreads();  // previous chunk's reads of lds_x are done
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(lds_x + (srow + 4 * q) * LD + scol) = xv[q];
            __syncthreads();
            {  // fetch the next chunk (or the first chunk of the next block) while this one is computed
                const int cn = (c + 1 < n_chunks) ? c + 1 : 0;
                const int coln = (c + 1 < n_chunks) ? col0 : col0 + CB;
                if (coln < col_hi) load_x(coln, cn, xv);
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
                if (col < p.nB) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = crow(r, h);
                        float y = acc[r];
                        if (p.act_prev == NIC_ACT_ELU) y *= elu_grad_from_out(lds_x[kl * LD + cb * 32 + li]);
                        dxg[(c * 32 + kl) * ldb32 + col] = y;
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
        if (c >= n_chunks) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = crow(r, h);
            if (n < p.N) slab[(int64_t)n * p.lds_ + k_base + c * 32 + li] += wacc[c][r];
        }
    }
    if (kg == 0) {
        const float total = bias_acc + __shfl_xor(bias_acc, 32);
        if (h == 0 && li < p.N) slab[(int64_t)li * p.lds_ + p.K] += total;
    }
}

}  // namespace

extern "C" {

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
    if (steps <= 4) hipLaunchKernelGGL(thin_bwd_kernel<4>, grid, block, 0, s, p);
    else if (steps <= 9) hipLaunchKernelGGL(thin_bwd_kernel<9>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(thin_bwd_kernel<16>, grid, block, 0, s, p);
    return nic::check_launch("nic_linear_bwd_thin");
}

}  // extern "C"
