// Policy-MLP layers on the gfx950 FP32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products and accumulation,
// bit-for-bit an fmaf chain, 256 FLOP/clk/CU = 157 TFLOP/s peak) — the reference's aten::addmm/mm + elu /
// elu_backward (neural_networks.py:80-106) re-designed around feature-major activations.
//
// Formulation.  Activations are stored feature-major, X[k][b] (scenario index contiguous), so
//     forward   Y[n][b]  = act( sum_k W[n][k]  X[k][b] + bias[n] )         "wx" kernel, A = W   (N x K)
//     dgrad     dX[k][b] = ( sum_n Wt[k][n] dY[n][b] ) * act'(H[k][b])     "wx" kernel, A = W^T (K x N)
//     wgrad     dW[n][k] += sum_b dY[n][b] X[k][b]                          "wgrad" kernel, contraction over scenarios
// In the wx kernel the MFMA's lane index is the scenario: the B operand (lane l holds B[k=l>>5][j=l&31]) and the
// C/D tile (col = lane&31) are read / written as contiguous 128-byte scenario runs, i.e. coalesced with no
// transposition anywhere between the env-step kernels and the GEMMs.
//
// Tiling (256 threads = 4 waves): block tile BM x BN x 32, wave tile (MT*32) x (NT*32) of 32x32x2 MFMAs.
//   A tile  : LDS [BM][36]  (rows padded by 4 floats: ds_read_b128 of 16 rows then hits 16 distinct bank quads)
//   B tile  : LDS [32][BN]  for the wx kernel (ds_read_b32: each 32-lane half reads one contiguous row segment)
//             LDS [BN][36]  for the wgrad kernel (same shape as A: the contraction index is the contiguous one)
//   k order : within a 16-deep k group, MFMA step kk consumes k = kk (lanes 0-31) and k = 8+kk (lanes 32-63), so a
//             lane's eight A values of a group are CONTIGUOUS in LDS (two ds_read_b128) — summation order over k is
//             free because parity is defined to 1e-5, not bitwise, for the policy GEMMs.
//   pipeline: register-staged double buffering — global loads of tile t+1 are issued before the MFMAs of tile t and
//             written to the other LDS buffer after them; one barrier per k tile.
// Roofline: MFMA-bound (2*N*K flops per scenario per layer; 128x128x32 tiles read 32 flop/byte from L2).
#include "nic_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;       // k depth of one LDS tile
constexpr int LDA_S = 36;    // padded LDS row (floats) for k-contiguous tiles
constexpr int kThreads = 256;

enum { EPI_BIAS_ACT = 0, EPI_DGRAD = 1 };

struct WxParams {
    const float* A;      // [M][lda]
    int64_t lda;
    const float* Bm;     // [K][ldb]
    float* C;            // [M][ldb]
    const float* bias;   // [M] or null            (EPI_BIAS_ACT)
    const float* Hprev;  // [M][ldb] or null        (EPI_DGRAD: multiply by act'(Hprev))
    int M, K, ncols;     // ncols = columns to produce (multiple of 4, <= ldb)
    int64_t ldb;
    int act;             // NIC_ACT_*
    int accumulate;      // EPI_DGRAD: C += result
};

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }
// derivative of ELU expressed with its OUTPUT y: x > 0 -> 1, else exp(x) = y + 1
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

// ---- tile loaders (global -> registers) -----------------------------------------------------------------------
// k-contiguous tile: ROWS x 32 floats, element (r, k) at src[(row0 + r) * ld + k0 + k]; rows >= nrows and k >= kmax
// are zero-filled.  Each thread moves ROWS/32 float4 (thread t: row = t/8 + 32*i, k4 = t%8).
template <int ROWS>
struct KTile {
    static constexpr int N4 = ROWS / 32;
    float4 v[N4];
    __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int row0, int nrows, int k0, int kmax,
                                         bool vec_ok, int ones_row = -1, int ones_cols = 0) {
        const int t = threadIdx.x;
        const int k = k0 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < N4; ++i) {
            const int r = row0 + (t >> 3) + 32 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                const float* p = src + (int64_t)r * ld + k;
                if (vec_ok && k + 3 < kmax) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < kmax) x.x = p[0];
                    if (k + 1 < kmax) x.y = p[1];
                    if (k + 2 < kmax) x.z = p[2];
                    if (k + 3 < kmax) x.w = p[3];
                }
            } else if (r == ones_row) {  // virtual row of ones: its wgrad column is the bias gradient
                x.x = k + 0 < ones_cols ? 1.f : 0.f;
                x.y = k + 1 < ones_cols ? 1.f : 0.f;
                x.z = k + 2 < ones_cols ? 1.f : 0.f;
                x.w = k + 3 < ones_cols ? 1.f : 0.f;
            }
            v[i] = x;
        }
    }
    __device__ __forceinline__ void store(float* lds /* [ROWS][LDA_S] */) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < N4; ++i)
            *reinterpret_cast<float4*>(lds + ((t >> 3) + 32 * i) * LDA_S + (t & 7) * 4) = v[i];
    }
};

// column-contiguous tile: 32 (k) x COLS floats, element (k, c) at src[(k0 + k) * ld + c0 + c]; k >= kmax and
// c >= ncols zero-filled.  Thread t: c4 = t % (COLS/4), k = t / (COLS/4) + (256*4/COLS) * i.
template <int COLS>
struct CTile {
    static constexpr int TPR = COLS / 4;            // threads per row
    static constexpr int RPI = kThreads / TPR;      // rows per iteration
    static constexpr int N4 = BK / RPI;
    float4 v[N4];
    __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int k0, int kmax, int c0, int ncols) {
        const int t = threadIdx.x;
        const int c = c0 + (t % TPR) * 4;
#pragma unroll
        for (int i = 0; i < N4; ++i) {
            const int k = k0 + t / TPR + RPI * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kmax && c < ncols) x = *reinterpret_cast<const float4*>(src + (int64_t)k * ld + c);
            v[i] = x;
        }
    }
    __device__ __forceinline__ void store(float* lds /* [32][COLS] */) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < N4; ++i)
            *reinterpret_cast<float4*>(lds + (t / TPR + RPI * i) * COLS + (t % TPR) * 4) = v[i];
    }
};

// eight k-contiguous operand values of one lane for k group g (see "k order" above)
__device__ __forceinline__ void read_frag8(const float* lds_row, int g, int h, float (&f)[8]) {
    const float4 lo = *reinterpret_cast<const float4*>(lds_row + g * 16 + h * 8);
    const float4 hi = *reinterpret_cast<const float4*>(lds_row + g * 16 + h * 8 + 4);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
}

// ---------------------------------------------------------------------------------------------------------------
// wx kernel: C[M][ncols] = epilogue( A[M][K] * Bm[K][ncols] )
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int MT, int NT, int EPI>
__global__ __launch_bounds__(kThreads) void gemm_wx_kernel(WxParams p) {
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int STAGE = BM * LDA_S + BK * BN;  // floats per pipeline stage: A tile then B tile
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    // row tiles vary fastest so that the workgroups sharing one column panel of Bm are dispatched back to back
    const int tiles_m = (p.M + BM - 1) / BM;
    const int m0 = (blockIdx.x % tiles_m) * BM;
    const int c0 = (blockIdx.x / tiles_m) * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;
    const bool vecA = (p.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    KTile<BM> ta;
    CTile<BN> tb;
    const int nk = (p.K + BK - 1) / BK;
    ta.load(p.A, p.lda, m0, p.M, 0, p.K, vecA);
    tb.load(p.Bm, p.ldb, 0, p.K, c0, p.ncols);
    ta.store(lds);
    tb.store(lds + BM * LDA_S);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            ta.load(p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, vecA);
            tb.load(p.Bm, p.ldb, (kt + 1) * BK, p.K, c0, p.ncols);
        }
        const float* a_base = lds + cur * STAGE + (wm * MT * 32 + li) * LDA_S;
        const float* b_base = lds + cur * STAGE + BM * LDA_S + wn * NT * 32 + li;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a[MT][8];
#pragma unroll
            for (int i = 0; i < MT; ++i) read_frag8(a_base + i * 32 * LDA_S, g, h, a[i]);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float b[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = b_base[(g * 16 + h * 8 + kk) * BN + j * 32];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            ta.store(lds + (cur ^ 1) * STAGE);
            tb.store(lds + (cur ^ 1) * STAGE + BM * LDA_S);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            float bias = 0.f;
            if (EPI == EPI_BIAS_ACT && p.bias) bias = p.bias[row];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = c0 + (wn * NT + j) * 32 + li;
                if (col >= p.ncols) continue;
                const int64_t off = (int64_t)row * p.ldb + col;
                float y = acc[i][j][r];
                if (EPI == EPI_BIAS_ACT) {
                    y += bias;
                    if (p.act == NIC_ACT_ELU) y = elu_f(y);
                } else {
                    if (p.Hprev && p.act == NIC_ACT_ELU) y *= elu_grad_from_out(p.Hprev[off]);
                    if (p.accumulate) y += p.C[off];
                }
                p.C[off] = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad kernel: slab[split][n][k] += sum_{b in split} dY[n][b] * X[k][b]   (k == K: bias column, X row of ones)
// ---------------------------------------------------------------------------------------------------------------
struct WgParams {
    const float* dY;  // [N][ldb]
    const float* X;   // [K][ldb]
    float* slab;      // [n_splits][N][lds]
    int64_t lds_, ldb;
    int N, K, nB, chunk;  // chunk = scenarios per split (multiple of 32)
};

template <int WAVES_M, int WAVES_N, int MT, int NT>
__global__ __launch_bounds__(kThreads) void gemm_wgrad_kernel(WgParams p) {
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    constexpr int STAGE = (BM + BN) * LDA_S;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int n0 = blockIdx.y * BM;   // output rows  (features of dY)
    const int k0 = blockIdx.x * BN;   // output cols  (features of X, plus the bias column K)
    const int split = blockIdx.z;
    const int b_begin = split * p.chunk;
    const int b_end = min(b_begin + p.chunk, p.nB);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;
    float* slab = p.slab + (int64_t)split * p.N * p.lds_;

    // accumulators start from the slab (the running sum over the periods of the rollout)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = k0 + (wn * NT + j) * 32 + li;
                acc[i][j][r] = (row < p.N && col <= p.K) ? slab[(int64_t)row * p.lds_ + col] : 0.f;
            }

    if (b_begin < b_end) {
        const bool vec = (p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.dY) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);
        KTile<BM> ta;
        KTile<BN> tb;
        const int nt = (b_end - b_begin + BK - 1) / BK;
        ta.load(p.dY, p.ldb, n0, p.N, b_begin, b_end, vec);
        tb.load(p.X, p.ldb, k0, p.K, b_begin, b_end, vec, p.K, b_end);
        ta.store(lds);
        tb.store(lds + BM * LDA_S);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) {
                ta.load(p.dY, p.ldb, n0, p.N, b_begin + (t + 1) * BK, b_end, vec);
                tb.load(p.X, p.ldb, k0, p.K, b_begin + (t + 1) * BK, b_end, vec, p.K, b_end);
            }
            const float* a_base = lds + cur * STAGE + (wm * MT * 32 + li) * LDA_S;
            const float* b_base = lds + cur * STAGE + (BM + wn * NT * 32 + li) * LDA_S;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float a[MT][8], b[NT][8];
#pragma unroll
                for (int i = 0; i < MT; ++i) read_frag8(a_base + i * 32 * LDA_S, g, h, a[i]);
#pragma unroll
                for (int j = 0; j < NT; ++j) read_frag8(b_base + j * 32 * LDA_S, g, h, b[j]);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            }
            if (t + 1 < nt) {
                ta.store(lds + (cur ^ 1) * STAGE);
                tb.store(lds + (cur ^ 1) * STAGE + BM * LDA_S);
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = k0 + (wn * NT + j) * 32 + li;
                if (row < p.N && col <= p.K) slab[(int64_t)row * p.lds_ + col] = acc[i][j][r];
            }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int64_t lds_, int n_splits, float* __restrict__ dW,
                                    int64_t lddw, float* __restrict__ db, int N, int K, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * (K + 1);
    if (idx >= total) return;
    const int n = (int)(idx / (K + 1)), k = (int)(idx % (K + 1));
    float s = 0.f;
    for (int sp = 0; sp < n_splits; ++sp) s += slab[((int64_t)sp * N + n) * lds_ + k];
    s *= scale;
    if (k < K) dW[(int64_t)n * lddw + k] = s;
    else if (db) db[n] = s;
}

template <int WM, int WN, int MT, int NT, int EPI>
void launch_wx(const WxParams& p, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.ncols + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_wx_kernel<WM, WN, MT, NT, EPI>), dim3(tiles_m * tiles_n), dim3(kThreads), 0, s, p);
}

template <int EPI>
void dispatch_wx(const WxParams& p, hipStream_t s) {
    if (p.M > 64) launch_wx<2, 2, 2, 2, EPI>(p, s);        // 128 x 128
    else if (p.M > 32) launch_wx<1, 4, 2, 1, EPI>(p, s);   //  64 x 128
    else launch_wx<1, 4, 1, 2, EPI>(p, s);                 //  32 x 256
}

template <int WM, int WN, int MT, int NT>
void launch_wg(const WgParams& p, int n_splits, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    dim3 grid((p.K + 1 + BN - 1) / BN, (p.N + BM - 1) / BM, n_splits);
    hipLaunchKernelGGL((gemm_wgrad_kernel<WM, WN, MT, NT>), grid, dim3(kThreads), 0, s, p);
}

// the tile shape is a function of (N, K) only so that nic_wgrad_num_splits and the launch agree
void wgrad_tile(int N, int K, int* bm, int* bn) {
    if (N > 64) { *bm = 128; *bn = (K + 1 > 64) ? 128 : 64; }
    else if (N > 32) { *bm = 64; *bn = 128; }
    else { *bm = 32; *bn = 256; }
}

int require_ld(const char* who, int32_t n_scenarios, int32_t ldb) {
    NIC_REQUIRE(n_scenarios > 0 && ldb >= n_scenarios && ldb % 4 == 0, "%s: ldb (%d) must be a multiple of 4 and >= n_scenarios (%d)",
                who, ldb, n_scenarios);
    return 0;
}

}  // namespace

extern "C" {

int nic_linear_fwd(const float* W, int64_t ldw, const float* bias, const float* X, float* Y, int32_t N, int32_t K,
                   int32_t n_scenarios, int32_t ldb, int32_t act, void* stream) {
    NIC_REQUIRE(W && X && Y, "nic_linear_fwd: null buffer");
    NIC_REQUIRE(N > 0 && K > 0 && ldw >= K, "nic_linear_fwd: bad N/K/ldw (%d/%d/%lld)", N, K, (long long)ldw);
    NIC_REQUIRE(act == NIC_ACT_NONE || act == NIC_ACT_ELU, "nic_linear_fwd: unknown activation %d", act);
    if (int e = require_ld("nic_linear_fwd", n_scenarios, ldb)) return e;
    NIC_REQUIRE((reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0,
                "nic_linear_fwd: X/Y must be 16-byte aligned");
    WxParams p{W, ldw, X, Y, bias, nullptr, N, K, (n_scenarios + 3) / 4 * 4, ldb, act, 0};
    dispatch_wx<EPI_BIAS_ACT>(p, nic::as_stream(stream));
    return nic::check_launch("nic_linear_fwd");
}

int nic_linear_dgrad(const float* Wt, int64_t ldwt, const float* dY, const float* Hprev, float* dX, int32_t N, int32_t K,
                     int32_t n_scenarios, int32_t ldb, int32_t act_prev, int32_t accumulate, void* stream) {
    NIC_REQUIRE(Wt && dY && dX, "nic_linear_dgrad: null buffer");
    NIC_REQUIRE(N > 0 && K > 0 && ldwt >= N, "nic_linear_dgrad: bad N/K/ldwt (%d/%d/%lld)", N, K, (long long)ldwt);
    if (int e = require_ld("nic_linear_dgrad", n_scenarios, ldb)) return e;
    NIC_REQUIRE((reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(dX) & 15) == 0,
                "nic_linear_dgrad: dY/dX must be 16-byte aligned");
    // dX[K][b] = Wt[K][N] * dY[N][b]: output rows = K, contraction = N
    WxParams p{Wt, ldwt, dY, dX, nullptr, Hprev, K, N, (n_scenarios + 3) / 4 * 4, ldb, act_prev, accumulate};
    dispatch_wx<EPI_DGRAD>(p, nic::as_stream(stream));
    return nic::check_launch("nic_linear_dgrad");
}

int nic_wgrad_num_splits(int32_t N, int32_t K, int32_t n_scenarios) {
    if (N <= 0 || K <= 0 || n_scenarios <= 0) return 0;
    int bm, bn;
    wgrad_tile(N, K, &bm, &bn);
    const int tiles = ((N + bm - 1) / bm) * ((K + 1 + bn - 1) / bn);
    int splits = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU in total
    const int max_splits = (n_scenarios + 255) / 256;  // at least 256 scenarios (8 k-tiles) per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

int nic_linear_wgrad(const float* dY, const float* X, float* slab, int64_t lds_, int32_t N, int32_t K, int32_t n_scenarios,
                     int32_t ldb, int32_t n_splits, void* stream) {
    NIC_REQUIRE(dY && X && slab, "nic_linear_wgrad: null buffer");
    NIC_REQUIRE(N > 0 && K > 0 && lds_ >= K + 1, "nic_linear_wgrad: bad N/K/lds (%d/%d/%lld)", N, K, (long long)lds_);
    NIC_REQUIRE(n_splits >= 1, "nic_linear_wgrad: n_splits must be >= 1");
    if (int e = require_ld("nic_linear_wgrad", n_scenarios, ldb)) return e;
    int chunk = (n_scenarios + n_splits - 1) / n_splits;
    chunk = (chunk + BK - 1) / BK * BK;
    WgParams p{dY, X, slab, lds_, ldb, N, K, n_scenarios, chunk};
    hipStream_t s = nic::as_stream(stream);
    int bm, bn;
    wgrad_tile(N, K, &bm, &bn);
    if (bm == 128 && bn == 128) launch_wg<2, 2, 2, 2>(p, n_splits, s);
    else if (bm == 128) launch_wg<2, 2, 2, 1>(p, n_splits, s);
    else if (bm == 64) launch_wg<1, 4, 2, 1>(p, n_splits, s);
    else launch_wg<1, 4, 1, 2>(p, n_splits, s);
    return nic::check_launch("nic_linear_wgrad");
}

int nic_wgrad_reduce(const float* slab, int64_t lds_, int32_t n_splits, float* dW, int64_t lddw, float* db, int32_t N,
                     int32_t K, float scale, void* stream) {
    NIC_REQUIRE(slab && dW, "nic_wgrad_reduce: null buffer");
    NIC_REQUIRE(N > 0 && K > 0 && lds_ >= K + 1 && lddw >= K && n_splits >= 1, "nic_wgrad_reduce: bad sizes");
    const int64_t total = (int64_t)N * (K + 1);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nic::ceil_div(total, 256)), dim3(256), 0, nic::as_stream(stream), slab, lds_,
                       n_splits, dW, lddw, db, N, K, scale);
    return nic::check_launch("nic_wgrad_reduce");
}
}
