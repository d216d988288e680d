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
// Production kernels (16-byte aligned operands — everything the rollout engine allocates):
//   gemm_wx_dma_kernel     forward + dgrad; 128 x 128 x 32 block tile of 8 waves (2 x 4, wave tile 64 x 32), two workgroups
//                          co-resident per CU; 64 x 128 (8 waves) and 32 x 128 (4 waves) for small launches / thin or ragged
//                          outputs (pick_wx_tile; round 4: co-resident workgroups beat one 256 x 256 workgroup per CU)
//   gemm_wgrad_dma_kernel  weight gradient, 128 x 256 output tile per (scenario chunk, period group), 8 waves (2 x 4)
//   both: tiles go HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` (no VGPR round trip), double-buffered, one barrier per
//   k tile; A-style tiles [rows][32] are unpadded with a 16-byte-chunk XOR swizzle applied on the DMA source address
//   and on the ds_read_b128 fragment reads; the [32][BN] B tile of the wx kernel is linear (ds_read_b32 rows);
//   epilogues are staged through the dead LDS tiles and leave as whole-row float4 accesses; logical tile order is
//   XCD-aware.
//   k order : within a 16-deep k group, MFMA step kk consumes k = kk (lanes 0-31) and k = 8+kk (lanes 32-63), so a
//             lane's eight A values of a group are CONTIGUOUS in LDS (two ds_read_b128) — summation order over k is
//             free because parity is defined to 1e-5, not bitwise, for the policy GEMMs.
// Fallback kernels (gemm_wx_kernel, gemm_wgrad_kernel): register-staged, per-element guards, for unaligned operands,
// scenario counts that are not a multiple of 32, and the thin layers of the weight gradient.
// Roofline: MFMA-bound.  Measured on MI355X (512 x 512 x 65,536): 102-114 TFLOP/s; MFMA pipe 70 % busy at a 2.26 GHz
// sustained clock (profiles/r01_gemm_pmc_dma256.txt); the remainder is the per-tile DMA wait + barrier (a no-DMA timing
// build runs 117 TFLOP/s) and the unoverlapped prologue / epilogue of a 1-workgroup-per-CU kernel.
#include <stdlib.h>

#include "nic_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;       // k depth of one LDS tile
constexpr int LDA_S = 36;    // padded LDS row (floats) for k-contiguous tiles
constexpr int kThreads = 256;

enum { EPI_BIAS_ACT = 0, EPI_DGRAD = 1 };
const char* epi_name(int epi) { return epi == EPI_BIAS_ACT ? "EPI_BIAS_ACT" : "EPI_DGRAD"; }

// In-kernel timestamps and timing-only switches (tuning build only: tools/gemm_stamp_probe.py compiles its own copy of the
// library with -DNIC_TUNING_BUILD; the product library contains none of this).  Slot s of workgroup w: {s_memtime, 100 MHz wall
// clock} at kernel entry (0), in front of the k loop (1), behind it (2) and at exit (3).
#ifdef NIC_TUNING_BUILD
__device__ unsigned long long* g_nic_stamps = nullptr;
#define NIC_STAMP(slot)                                                                  \
    do {                                                                                 \
        if (g_nic_stamps != nullptr && threadIdx.x == 0) {                               \
            g_nic_stamps[(blockIdx.x * 4 + (slot)) * 2] = __builtin_readcyclecounter();  \
            g_nic_stamps[(blockIdx.x * 4 + (slot)) * 2 + 1] = wall_clock64();            \
        }                                                                                \
    } while (0)
#define NIC_TUNE(bit) (p.tune & (bit))
#else
#define NIC_STAMP(slot) do { } while (0)
#define NIC_TUNE(bit) false
#endif

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
    int tune;            // tuning build only, timing experiments (results invalid): 1 = no A-tile copies inside the k loop,
                         // 2 = no B-tile copies, 4 = no barrier per k tile, 32 = s_setprio 1 for waves 4-7 (valid results),
                         // 8 = dgrad without the Hprev prefetch, 16 = stamp 1 behind the plain k tiles, 64 = forward epilogue
                         // straight from the accumulators, 128 = 256 x 256 tiling whatever the shape (8-128: valid results)
};

// ELU.  libm's expm1f is ~40 VALU instructions and the epilogue applies it to every output element (33.5 M per 512-wide
// layer at 65,536 scenarios: ~20 us of pure VALU per launch); this branch-free form is a 6-term series near zero (relative
// error < 2e-7 for x > -0.35) and exp(x) - 1 on the hardware exponential below that (absolute error ~1e-7).
__device__ __forceinline__ float elu_f(float x) {
    const float xn = fminf(x, 0.f);
    const float series =
        xn * fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float viaexp = __expf(xn) - 1.f;
    const float neg = xn > -0.35f ? series : viaexp;
    return x > 0.f ? x : neg;
}
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

// ---- branch-free tile loaders (buffer loads: hardware range check returns 0 past num_records) ------------------------
// The guarded loaders above branch per element and hipcc serialises their loads behind s_waitcnt; in the k loop that
// stalls every tile.  When the operand rows are 16-byte aligned the tile is fetched with raw buffer loads instead: rows
// past the end of the matrix are out of range of the descriptor and read as 0.0 without a branch, so the loads of a
// tile issue back to back.  Columns past K inside an A row read whatever follows (row padding or the next row — finite
// weights) and are multiplied by B rows that ARE out of range (zero), so no k guard is needed either.
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int64_t n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
}

// LDS-DMA issued from inline asm.  With the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc cannot prove that the
// ds_reads of pipeline stage `cur` do not alias the DMA writes into stage `cur ^ 1` (same __shared__ array) and puts an
// s_waitcnt vmcnt(0) in front of the first fragment read of EVERY tile: the copy it was meant to overlap is waited for
// before the MFMAs start (measured: 13 % of the kernel).  An asm statement is outside hipcc's vmcnt bookkeeping; the
// kernels drain the DMA themselves (dma_wait) right before the barrier that publishes the tile.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 make_desc(const float* p, int64_t n_floats) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    u32x4 d;
    d.x = (unsigned)a;
    d.y = (unsigned)(a >> 32) & 0xffffu;  // stride 0: raw buffer
    d.z = (unsigned)(n_floats * 4);       // num_records in bytes: loads past it return 0
    d.w = 0x00020000u;
    return d;
}
// lds_addr: wave-uniform LDS byte address; lane l's 16 bytes land at lds_addr + 16 * l.  M0 is saved / restored inside the
// statement (it is compiler-reserved); s_nop 0 covers the SALU-write-M0 -> LDS-DMA hazard.
__device__ __forceinline__ void dma16(u32x4 desc, int voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(desc), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int ROWS>
struct KTileBuf {
    static constexpr int N4 = ROWS / 32;
    float4 v[N4];
    int off[N4];  // byte offsets of this thread's float4s in the current tile
    __device__ __forceinline__ void init(int64_t ld, int row0, int k0) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < N4; ++i) off[i] = (int)((((int64_t)row0 + (t >> 3) + 32 * i) * ld + k0 + (t & 7) * 4) * 4);
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc) {
#pragma unroll
        for (int i = 0; i < N4; ++i) v[i] = buf_load4(rsrc, off[i]);
    }
    __device__ __forceinline__ void advance(int tiles = 1) {
#pragma unroll
        for (int i = 0; i < N4; ++i) off[i] += tiles * BK * 4;
    }
    // wgrad: zero the scenarios at or past b_end; row `ones_row` is a virtual row of ones (bias gradient column)
    __device__ __forceinline__ void mask_cols(int b0, int b_end, int row0, int ones_row) {
        const int t = threadIdx.x;
        const int b = b0 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < N4; ++i) {
            const bool one = (row0 + (t >> 3) + 32 * i) == ones_row;
            float4 x = v[i];
            x.x = b + 0 < b_end ? (one ? 1.f : x.x) : 0.f;
            x.y = b + 1 < b_end ? (one ? 1.f : x.y) : 0.f;
            x.z = b + 2 < b_end ? (one ? 1.f : x.z) : 0.f;
            x.w = b + 3 < b_end ? (one ? 1.f : x.w) : 0.f;
            v[i] = x;
        }
    }
    __device__ __forceinline__ void store(float* lds) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < N4; ++i)
            *reinterpret_cast<float4*>(lds + ((t >> 3) + 32 * i) * LDA_S + (t & 7) * 4) = v[i];
    }
};

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2).  Remap the linear workgroup id so that every
// XCD gets a CONTIGUOUS range of logical tiles: the tiles that share an operand panel (the row tiles of one scenario
// panel in the wx kernel, the 16 output tiles of one scenario chunk in the wgrad kernel) then run on one XCD at about
// the same time and the panel is fetched from HBM once instead of once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// eight k-contiguous operand values of one lane for k group g (see "k order" above)
__device__ __forceinline__ void read_frag8(const float* lds_row, int g, int h, float (&f)[8]) {
    const float4 lo = *reinterpret_cast<const float4*>(lds_row + g * 16 + h * 8);
    const float4 hi = *reinterpret_cast<const float4*>(lds_row + g * 16 + h * 8 + 4);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
}

// ---------------------------------------------------------------------------------------------------------------
// wx kernel, guarded form: C[M][ncols] = epilogue( A[M][K] * Bm[K][ncols] ) for operands that are NOT 16-byte aligned /
// row-padded (e.g. an unpadded [N][51] weight handed over by HipLinear).  Register-staged double buffering with
// per-element guards; correctness path, not the performance path (aligned operands use gemm_wx_dma_kernel below).
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int MT, int NT, int EPI>
__global__ __launch_bounds__(kThreads) void gemm_wx_kernel(WxParams p) {
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int STAGE = BM * LDA_S + BK * BN;  // floats per pipeline stage: A tile then B tile
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

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

    const int nk = (p.K + BK - 1) / BK;
    KTile<BM> ta;
    CTile<BN> tb;
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

    // epilogue straight from the accumulators: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
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
// wx kernel, LDS-DMA form (the production path): tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds`, no VGPR
// round trip and no ds_write.  Measured on MI355X at 512x512x65536: the register-staged loop spends 13 % of its time
// in the ds_write_b128 transfer and 5 % in the loads; both disappear here.
//   A tile [BM][32] floats, UNPADDED (a DMA wave-instruction writes 1 KiB linearly = 8 rows), 16-byte chunks XOR-
//     swizzled with ((row >> 1) & 7) on the SOURCE address; the ds_read_b128 fragment reads apply the same XOR, which
//     makes every 16-lane read group hit 16 distinct bank quads (rows r, r+1 differ in the 32-bank half).
//   B tile [32][BN] floats, linear (row reads by ds_read_b32 are conflict-free as they are).
//   Rows past the matrix end are out of range of the buffer descriptor: the DMA writes zeros, no branches.
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int MT, int NT, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_wx_dma_kernel(WxParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // LDS address-space pointers only exist in the device pass
    constexpr int NW = WAVES_M * WAVES_N, NTHREADS = 64 * NW;
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
    constexpr int LDC = BN + 4;
    // epilogue staging: the whole C tile if it fits next to nothing else, otherwise one wave-row (MT*32 rows) per pass
    constexpr int PASSES = (BM * LDC * 4 <= 72 * 1024) ? 1 : WAVES_M;
    constexpr int PASS_ROWS = BM / PASSES;
    constexpr int LDS_FLOATS = (2 * STAGE > PASS_ROWS * LDC) ? 2 * STAGE : PASS_ROWS * LDC;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    NIC_STAMP(0);
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile % tiles_m) * BM;
    const int c0 = (tile / tiles_m) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;

    const u32x4 ra = make_desc(p.A, (int64_t)p.M * p.lda);
    const u32x4 rb = make_desc(p.Bm, (int64_t)p.K * p.ldb);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;  // LDS byte address of the staging array
    constexpr int A_INSTR = BM / 8 / NW;           // 1-KiB DMA instructions per wave for the A tile (8 rows each)
    constexpr int B_INSTR = BN / 8 / NW;           // ... for the B tile (256/BN rows each)
    constexpr int B_LPR = BN / 4;                  // lanes per B row
    static_assert(A_INSTR >= 1 && B_INSTR >= 1 && B_LPR <= 64, "tile too small for this wave count");
    int offA[A_INSTR], offB[B_INSTR];
#pragma unroll
    for (int q = 0; q < A_INSTR; ++q) {
        const int row = (wave * A_INSTR + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // LDS position (lane & 7) holds source chunk `chunk`
        offA[q] = (int)((((int64_t)m0 + row) * p.lda + chunk * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < B_INSTR; ++q) {
        const int row = (wave * B_INSTR + q) * (64 / B_LPR) + lane / B_LPR;
        offB[q] = (int)(((int64_t)row * p.ldb + c0 + (lane % B_LPR) * 4) * 4);
    }
    const int stepB = (int)(p.ldb * BK * 4);
    bool in_loop = false;  // (tuning build: the timing-only switches leave the prologue's copy alone)
    auto issue = [&](int stage) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(stage * STAGE) * 4u);
        const unsigned wbase_a = __builtin_amdgcn_readfirstlane(base + (unsigned)(wave * A_INSTR) * 1024u);
        const unsigned wbase_b = __builtin_amdgcn_readfirstlane(base + (unsigned)A_FLOATS * 4u + (unsigned)(wave * B_INSTR) * 1024u);
        if (!(NIC_TUNE(1) && in_loop)) {
#pragma unroll
            for (int q = 0; q < A_INSTR; ++q) dma16(ra, offA[q], wbase_a + q * 1024u);
        }
        if (!(NIC_TUNE(2) && in_loop)) {
#pragma unroll
            for (int q = 0; q < B_INSTR; ++q) dma16(rb, offB[q], wbase_b + q * 1024u);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int q = 0; q < A_INSTR; ++q) offA[q] += BK * 4;
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q) offB[q] += stepB;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // swizzled chunk positions of this lane's A rows: row = wm*MT*32 + i*32 + li -> ((row >> 1) & 7) == ((li >> 1) & 7)
    const int sw = (li >> 1) & 7;
    const int nk = (p.K + BK - 1) / BK;
    // epilogue geometry (needed here already: the dgrad epilogue's Hprev rows are fetched during the last k tiles)
    constexpr int TPR = BN / 4, RPI = NTHREADS / TPR, ITER = PASS_ROWS / RPI;
    static_assert(PASS_ROWS % RPI == 0, "pass rows must divide evenly over the threads");
    const int t = threadIdx.x;
    const int col = c0 + (t % TPR) * 4;
    // EPI_DGRAD multiplies the tile by act'(Hprev): read in the epilogue, those 256 KB per workgroup arrive as one burst from
    // every CU at once, on top of the write burst.  The kernel has ~80 free VGPRs, so the rows of the first pass are fetched
    // into registers during the last PFT k tiles (each batch completes under that tile's MFMAs, before its dma_wait), and
    // the rows of the second pass are fetched into the same registers as the first pass consumes them.
    constexpr int PFT = (ITER % 4 == 0) ? 4 : 2;
    static_assert(ITER % PFT == 0, "prefetch batches must divide the pass");
    // (the 7-row-tile wave holds 112 accumulators + 56 fragment registers: no room for the prefetch registers as well)
    constexpr bool PREFETCH_H = EPI == EPI_DGRAD && MT <= 4;
    float4 hq[PREFETCH_H ? ITER : 1];  // (dead in the forward instantiation)
    const bool pf_on = PREFETCH_H && p.Hprev != nullptr && p.act == NIC_ACT_ELU && nk >= PFT && !p.accumulate && !NIC_TUNE(8);
    auto load_h = [&](int pass, int it) {
        const int row = m0 + pass * PASS_ROWS + t / TPR + RPI * it;
        return (row < p.M && col < p.ncols) ? *reinterpret_cast<const float4*>(p.Hprev + (int64_t)row * p.ldb + col)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    auto ktile = [&](int kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {  // stage cur^1 was last read in tile kt-1; every wave is past that barrier
            advance();
            issue(cur ^ 1);
        }
        const float* a_base = lds + cur * STAGE + (wm * MT * 32 + li) * BK;
        const float* b_base = lds + cur * STAGE + A_FLOATS + wn * NT * 32 + li;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a[MT][8];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float4 lo = *reinterpret_cast<const float4*>(a_base + i * 32 * BK + (((g * 4 + h * 2) ^ sw) << 2));
                const float4 hi = *reinterpret_cast<const float4*>(a_base + i * 32 * BK + (((g * 4 + h * 2 + 1) ^ sw) << 2));
                a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = lo.z; a[i][3] = lo.w;
                a[i][4] = hi.x; a[i][5] = hi.y; a[i][6] = hi.z; a[i][7] = hi.w;
            }
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
        dma_wait();  // tile kt+1 has landed (this wave's share); the barrier publishes every wave's share
        if (!NIC_TUNE(4)) __syncthreads();
    };
    issue(0);
    dma_wait();
    __syncthreads();
    NIC_STAMP(1);
    in_loop = true;
#ifdef NIC_TUNING_BUILD
    // static priority for the younger half of the workgroup (cdna_hip_programming.md T5, static form)
    if (NIC_TUNE(32) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= NTHREADS / 2) __builtin_amdgcn_s_setprio(1);
#endif
    const int nk_plain = pf_on ? nk - PFT : nk;
    for (int kt = 0; kt < nk_plain; ++kt) ktile(kt);
    if (NIC_TUNE(16)) NIC_STAMP(1);  // (tuning build: the "k loop" phase then covers the prefetch tiles alone)
    if constexpr (EPI == EPI_DGRAD) {
        if (pf_on) {
#pragma unroll
            for (int b = 0; b < PFT; ++b) {
#pragma unroll
                for (int q = 0; q < ITER / PFT; ++q) hq[PREFETCH_H ? b * (ITER / PFT) + q : 0] = load_h(0, b * (ITER / PFT) + q);
                ktile(nk_plain + b);
            }
        }
    }

    NIC_STAMP(2);
    // epilogue: stage the block tile through LDS and write whole rows
    float* cs = lds;
    // Production case first (whole tile inside the matrix; forward: bias + ELU, dgrad: prefetched Hprev): straight-line row
    // loop, no per-row guards or option branches, so hipcc hoists the bias loads and keeps exact s_waitcnt counts — the
    // refill loads and the stores of one row do not stall the next row.
    const bool full_tile = m0 + BM <= p.M && c0 + BN <= p.ncols;
    const bool fast_epi = full_tile && (EPI == EPI_DGRAD ? pf_on : (p.bias != nullptr && p.act == NIC_ACT_ELU));
#ifdef NIC_TUNING_BUILD
    if (EPI == EPI_BIAS_ACT && fast_epi && NIC_TUNE(64)) {
        // experiment: forward epilogue straight from the accumulators (no LDS staging, no barriers): one dword store per
        // accumulator register, 32 lanes = 128 contiguous bytes of a row, the wave's two halves four rows apart
        const int64_t base = (int64_t)(m0 + wm * MT * 32 + 4 * h) * p.ldb + c0 + wn * NT * 32 + li;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = i * 32 + (r & 3) + 8 * (r >> 2);
                const float bias = p.bias[m0 + wm * MT * 32 + 4 * h + rl];
#pragma unroll
                for (int j = 0; j < NT; ++j) p.C[base + (int64_t)rl * p.ldb + j * 32] = elu_f(acc[i][j][r] + bias);
            }
        NIC_STAMP(3);
        return;
    }
#endif
    if (fast_epi) {
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            if (PASSES == 1 || wm == pass) {
                const int wrow = (PASSES == 1) ? wm * MT * 32 : 0;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cs[(wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC + (wn * NT + j) * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
            const int row0 = m0 + pass * PASS_ROWS + t / TPR;
            float bias_r[ITER];
            if (EPI == EPI_BIAS_ACT) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) bias_r[it] = p.bias[row0 + RPI * it];
            }
            // 32-bit element offset, advanced row group by row group (M * ldb < 2^28 on this path): one address register
            int off = row0 * (int)p.ldb + col;
            const int off_step = RPI * (int)p.ldb, off_pass = PASS_ROWS * (int)p.ldb;
#pragma unroll
            for (int it = 0; it < ITER; ++it, off += off_step) {
                const int row_l = t / TPR + RPI * it;
                float4 y = *reinterpret_cast<const float4*>(cs + row_l * LDC + (t % TPR) * 4);
                if (EPI == EPI_BIAS_ACT) {
                    y.x = elu_f(y.x + bias_r[it]); y.y = elu_f(y.y + bias_r[it]);
                    y.z = elu_f(y.z + bias_r[it]); y.w = elu_f(y.w + bias_r[it]);
                } else {
                    const float4 hv = hq[PREFETCH_H ? it : 0];
                    y.x *= elu_grad_from_out(hv.x); y.y *= elu_grad_from_out(hv.y);
                    y.z *= elu_grad_from_out(hv.z); y.w *= elu_grad_from_out(hv.w);
                    if (pass + 1 < PASSES)  // next pass's row into the freed register
                        hq[PREFETCH_H ? it : 0] = *reinterpret_cast<const float4*>(p.Hprev + off + off_pass);
                }
                *reinterpret_cast<float4*>(p.C + off) = y;
                // keep the scheduler from hoisting all 16 staged rows (64 VGPRs on top of 128 accumulators + 64 Hprev
                // registers spills): rows are scheduled in groups of four
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
        NIC_STAMP(3);
        return;
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (PASSES == 1 || wm == pass) {
            const int wrow = (PASSES == 1) ? wm * MT * 32 : 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        cs[(wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC + (wn * NT + j) * 32 + li] = acc[i][j][r];
        }
        __syncthreads();
        float bias_g[ITER];   // (fetched together: one load - wait - store round trip per row otherwise)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = m0 + pass * PASS_ROWS + t / TPR + RPI * it;
            bias_g[it] = (EPI == EPI_BIAS_ACT && p.bias) ? p.bias[row < p.M ? row : 0] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row_l = t / TPR + RPI * it;
            const int row = m0 + pass * PASS_ROWS + row_l;
            if (row >= p.M || col >= p.ncols) continue;
            float4 y = *reinterpret_cast<const float4*>(cs + row_l * LDC + (t % TPR) * 4);
            const int64_t off = (int64_t)row * p.ldb + col;
            if (EPI == EPI_BIAS_ACT) {
                const float bias = bias_g[it];
                y.x += bias; y.y += bias; y.z += bias; y.w += bias;
                if (p.act == NIC_ACT_ELU) { y.x = elu_f(y.x); y.y = elu_f(y.y); y.z = elu_f(y.z); y.w = elu_f(y.w); }
            } else {
                if (p.Hprev && p.act == NIC_ACT_ELU) {
                    const float4 hv = pf_on ? hq[PREFETCH_H ? it : 0] : *reinterpret_cast<const float4*>(p.Hprev + off);
                    y.x *= elu_grad_from_out(hv.x); y.y *= elu_grad_from_out(hv.y);
                    y.z *= elu_grad_from_out(hv.z); y.w *= elu_grad_from_out(hv.w);
                    if (pf_on && pass + 1 < PASSES) hq[PREFETCH_H ? it : 0] = load_h(pass + 1, it);  // next pass's row into the freed register
                }
                if (p.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(p.C + off);
                    y.x += o.x; y.y += o.y; y.z += o.z; y.w += o.w;
                }
            }
            *reinterpret_cast<float4*>(p.C + off) = y;
        }
        if (pass + 1 < PASSES) __syncthreads();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// wx kernel, streamed form for SMALL launches (round 4): the reference's shipped batch of 1,024 scenarios, the thin logits /
// first-layer-gradient GEMMs at every batch size.  A launch with a few hundred 32 x 32 output tiles cannot fill the chip with
// workgroup tiles, and a lone workgroup's LDS pipeline pays ~0.45 us of copy round trip + barrier per 32-deep k tile that
// nothing overlaps (16 k tiles: 13-25 us for 0.5 GFLOP).  Here ONE WAVEFRONT owns a 32 x (32 NT) output tile and reads its
// operands straight from L2 into registers - no LDS tile, no barrier in the k loop: per 16-deep k group a lane fetches its 8
// A values as two 16-byte loads (row m0 + lane, k contiguous) and its 8 NT B values as dword loads (rows of X: 128 contiguous
// bytes per half wave), D = 4 groups ahead of the MFMAs that consume them (counted vmcnt, no waits on the critical path).
// KS wavefronts of a workgroup split the contraction (K / KS each) and add their accumulators through LDS once, so even a
// 17-row logits layer at 1,024 scenarios is 32 tiles x 4 = 128 wavefronts with 8 k groups each instead of one 16-tile chain.
// Operand traffic is tiles x K x 32 (1 + NT) x 4 B from L2 (weights and the activations of a small batch are L2-resident):
// only worth it while that stays below ~100 MB - pick_wx_stream.
// Same k order as the LDS-DMA kernel within a group (lanes 0-31: k = kk, lanes 32-63: k = 8 + kk).
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int KS, int EPI>
__global__ __launch_bounds__(64 * KS) void gemm_wx_stream_kernel(WxParams p) {
    constexpr int D = 4;
    __shared__ float red[(KS > 1 ? KS - 1 : 1) * NT * 16 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 31, h = lane >> 5;
    const int tiles_m = (p.M + 31) / 32;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);   // the row tiles of one column panel run on one XCD (shared L2)
    const int m0 = (tile % tiles_m) * 32, c0 = (tile / tiles_m) * (32 * NT);
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)p.M * p.lda), rB = make_rsrc(p.Bm, (int64_t)p.K * p.ldb);
    // this wavefront's k groups: [g_lo, g_lo + per), per a multiple of D; groups past K read B rows beyond the descriptor (zeros)
    const int ng_all = (p.K + 15) / 16;
    const int per = ((ng_all + KS - 1) / KS + D - 1) / D * D, nblk = per / D;
    const int g_lo = wave * per;
    const int ldb4 = (int)p.ldb * 4;
    int offA = (int)((((int64_t)m0 + li) * p.lda + g_lo * 16 + h * 8) * 4);
    int offB = (int)((((int64_t)g_lo * 16 + h * 8) * p.ldb + c0 + li) * 4);

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a[D][8], b[D][NT][8];
    auto load = [&](int d) {
        const float4 lo = buf_load4(rA, offA), hi = buf_load4(rA, offA + 16);
        a[d][0] = lo.x; a[d][1] = lo.y; a[d][2] = lo.z; a[d][3] = lo.w;
        a[d][4] = hi.x; a[d][5] = hi.y; a[d][6] = hi.z; a[d][7] = hi.w;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[d][j][kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, offB + j * 128, kk * ldb4, 0));
        offA += 64;
        offB += 16 * ldb4;
    };
    auto compute = [&](int d) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][kk], b[d][j][kk], acc[j], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d);
    for (int blk = 0; blk + 1 < nblk; ++blk) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            compute(d);
            load(d);
        }
    }
    // the epilogue's own operands are fetched under the last block's MFMAs (wave 0 writes the tile)
    const int row_base = m0 + 4 * h;   // accumulator register r holds row row_base + (r & 3) + 8 (r >> 2), column c0 + 32 j + li
    float aux[16];   // forward: bias[row]; dgrad: act'(Hprev) of tile column j = 0 (further columns are read in the epilogue)
    const bool use_h = EPI == EPI_DGRAD && p.Hprev != nullptr && p.act == NIC_ACT_ELU;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const bool ok = row < p.M && c0 + li < p.ncols;
            if (EPI == EPI_BIAS_ACT) aux[r] = (p.bias != nullptr && row < p.M) ? p.bias[row] : 0.f;
            else aux[r] = (use_h && ok) ? p.Hprev[(int64_t)row * p.ldb + c0 + li] : 1.f;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) compute(d);

    if constexpr (KS > 1) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave - 1) * NT + j) * 16 + r) * 64 + lane] = acc[j][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += red[((w * NT + j) * 16 + r) * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = c0 + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            if (row >= p.M || col >= p.ncols) continue;
            const int64_t off = (int64_t)row * p.ldb + col;
            float y = acc[j][r];
            if (EPI == EPI_BIAS_ACT) {
                y += aux[r];
                if (p.act == NIC_ACT_ELU) y = elu_f(y);
            } else {
                if (use_h) y *= elu_grad_from_out(j == 0 ? aux[r] : p.Hprev[off]);
                if (p.accumulate) y += p.C[off];
            }
            p.C[off] = y;
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
    int swz;              // XCD-aware tile order (DMA kernel)
    // periods (DMA kernel): the contraction runs over n_periods operand pairs dY + t * pstride_dy, X + t * pstride_x (elements)
    // with the same scenario split in each — one launch for the whole backward sweep of a rollout
    int n_periods;
    int64_t pstride_dy, pstride_x;
    // the register accumulators are added to the slab and cleared every `flush_periods` periods: an fp32 sum of 10^5
    // terms in one register drifts to ~5e-5 relative (measured against the fp32 CPU reference at T = 100 x 1,024 scenarios
    // per split); 8k-term partial sums keep the weight gradients at the per-period accuracy
    int flush_periods;
    // round 4: the slab slots are (period group, scenario split) pairs - slot = group * scen_splits + scenario split; a group
    // contracts periods [group * periods_per_group, ...) only.  Few scenarios x many periods (the reference's shipped batch of
    // 1,024, an 8-GPU shard of 8,192) then still gives every CU a workgroup.  scen_splits = 0: one group (all slots are
    // scenario splits).
    int scen_splits, periods_per_group;
};

template <int WAVES_M, int WAVES_N, int MT, int NT, int FAST>
__global__ __launch_bounds__(kThreads) void gemm_wgrad_kernel(WgParams p) {
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    constexpr int STAGE = (BM + BN) * LDA_S;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    // 1-D grid, tile index fastest: the tiles of one scenario chunk (split) are contiguous logical ids -> one XCD
    const int tiles_k = (p.K + 1 + BN - 1) / BN, tiles_n = (p.N + BM - 1) / BM;
    const int lid = blockIdx.x;  // (an XCD remap of this id measured 15 % SLOWER here: keep dispatch order)
    const int k0 = (lid % tiles_k) * BN;                  // output cols  (features of X, plus the bias column K)
    const int n0 = ((lid / tiles_k) % tiles_n) * BM;      // output rows  (features of dY)
    const int split = lid / (tiles_k * tiles_n);
    // slab slot = (period group, scenario split) when scen_splits > 0 (see WgParams): this workgroup's periods
    const int ssplit = p.scen_splits > 0 ? split % p.scen_splits : split;
    const int p_first = p.scen_splits > 0 ? (split / p.scen_splits) * p.periods_per_group : 0;
    const int p_end = p.scen_splits > 0 ? min(p_first + p.periods_per_group, p.n_periods) : p.n_periods;
    const int b_begin = ssplit * p.chunk;
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

    // periods: the accumulators persist over n_periods operand pairs (one slab read-modify-write per launch)
    for (int period = p_first; period < p_end && b_begin < b_end; ++period) {
        const float* dYp = p.dY + period * p.pstride_dy;
        const float* Xp = p.X + period * p.pstride_x;
        if (period > p_first) __syncthreads();  // the previous period's last tile has been read by every wave
        const bool vec = (p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(dYp) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Xp) & 15) == 0);
        KTile<BM> ta;
        KTile<BN> tb;
        KTileBuf<BM> fa;
        KTileBuf<BN> fb;
        __amdgpu_buffer_rsrc_t ra, rb;
        const int nt = (b_end - b_begin + BK - 1) / BK;
        if constexpr (FAST) {
            ra = make_rsrc(dYp, (int64_t)p.N * p.ldb);
            rb = make_rsrc(Xp, (int64_t)p.K * p.ldb);
            fa.init(p.ldb, n0, b_begin);
            fb.init(p.ldb, k0, b_begin);
            fa.load(ra);
            fb.load(rb);
            fa.mask_cols(b_begin, b_end, n0, -1);
            fb.mask_cols(b_begin, b_end, k0, p.K);
            fa.store(lds);
            fb.store(lds + BM * LDA_S);
        } else {
            ta.load(dYp, p.ldb, n0, p.N, b_begin, b_end, vec);
            tb.load(Xp, p.ldb, k0, p.K, b_begin, b_end, vec, p.K, b_end);
            ta.store(lds);
            tb.store(lds + BM * LDA_S);
        }
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) {
                if constexpr (FAST) {
                    fa.advance();
                    fb.advance();
                    fa.load(ra);
                    fb.load(rb);
                } else {
                    ta.load(dYp, p.ldb, n0, p.N, b_begin + (t + 1) * BK, b_end, vec);
                    tb.load(Xp, p.ldb, k0, p.K, b_begin + (t + 1) * BK, b_end, vec, p.K, b_end);
                }
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
                if constexpr (FAST) {
                    fa.mask_cols(b_begin + (t + 1) * BK, b_end, n0, -1);
                    fb.mask_cols(b_begin + (t + 1) * BK, b_end, k0, p.K);
                    fa.store(lds + (cur ^ 1) * STAGE);
                    fb.store(lds + (cur ^ 1) * STAGE + BM * LDA_S);
                } else {
                    ta.store(lds + (cur ^ 1) * STAGE);
                    tb.store(lds + (cur ^ 1) * STAGE + BM * LDA_S);
                }
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

// ---------------------------------------------------------------------------------------------------------------
// wgrad kernel, LDS-DMA form: both operands are row tiles with the contraction index (scenario) contiguous, so both use
// the swizzled 1-KiB DMA of the wx kernel's A tile.  Output tile BM x BN of dW for one scenario chunk (split); the tile is
// ADDED to the split's slab through the LDS-staged epilogue (whole-row float4 read-add-write).  The bias gradient needs
// no virtual row of ones here: the waves of the first column tile sum the dY fragments they read anyway.
// Requires n_scenarios % 32 == 0 (no partial k tiles) — the launcher falls back to gemm_wgrad_kernel otherwise.
// ---------------------------------------------------------------------------------------------------------------
// SKIP_ROWS: row tiles of a wave that lie entirely past N are not computed (N = 195 in a 256-row tile: the wave rows hold 4 and 3
// of the 7 row tiles with rows, and the two wavefronts of a SIMD are (wm = 0, wn) and (wm = 1, wn)).
template <int WAVES_M, int WAVES_N, int MT, int NT, bool SKIP_ROWS = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_wgrad_dma_kernel(WgParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WAVES_M * WAVES_N, NTHREADS = 64 * NW;
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int LDC = BN + 4;
    constexpr int PASSES = (BM * LDC * 4 <= 72 * 1024) ? 1 : WAVES_M;
    constexpr int PASS_ROWS = BM / PASSES;
    constexpr int LDS_FLOATS = (2 * STAGE > PASS_ROWS * LDC) ? 2 * STAGE : PASS_ROWS * LDC;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tiles_k = (p.K + BN - 1) / BN, tiles_n = (p.N + BM - 1) / BM;
    // the output tiles of one scenario chunk read the same dY / X rows: keep them on one XCD (shared L2)
    const int lid = p.swz ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int k0 = (lid % tiles_k) * BN;              // output cols (features of X)
    const int n0 = ((lid / tiles_k) % tiles_n) * BM;  // output rows (features of dY)
    const int split = lid / (tiles_k * tiles_n);
    const int ssplit = p.scen_splits > 0 ? split % p.scen_splits : split;   // scenario split / period group of this slab slot
    const int pgroup = p.scen_splits > 0 ? split / p.scen_splits : 0;
    const int p_first = pgroup * p.periods_per_group;
    const int n_per = p.scen_splits > 0 ? min(p.periods_per_group, p.n_periods - p_first) : p.n_periods;
    const int b_begin = ssplit * p.chunk;
    const int b_end = min(b_begin + p.chunk, p.nB);
    const int nt = (b_end - b_begin) / BK;            // whole tiles only (nB % 32 == 0)
    if (nt <= 0 || n_per <= 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;
    float* slab = p.slab + (int64_t)split * p.N * p.lds_;

    u32x4 ra = make_desc(p.dY + p_first * p.pstride_dy, (int64_t)p.N * p.ldb);
    u32x4 rb = make_desc(p.X + p_first * p.pstride_x, (int64_t)p.K * p.ldb);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    int offA[A_INSTR], offB[B_INSTR];
#pragma unroll
    for (int q = 0; q < A_INSTR; ++q) {
        const int row = (wave * A_INSTR + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        offA[q] = (int)((((int64_t)n0 + row) * p.ldb + b_begin + chunk * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < B_INSTR; ++q) {
        const int row = (wave * B_INSTR + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        offB[q] = (int)((((int64_t)k0 + row) * p.ldb + b_begin + chunk * 4) * 4);
    }
    auto issue = [&](int stage) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(stage * STAGE) * 4u);
        const unsigned wbase_a = __builtin_amdgcn_readfirstlane(base + (unsigned)(wave * A_INSTR) * 1024u);
        const unsigned wbase_b = __builtin_amdgcn_readfirstlane(base + (unsigned)A_FLOATS * 4u + (unsigned)(wave * B_INSTR) * 1024u);
#pragma unroll
        for (int q = 0; q < A_INSTR; ++q) dma16(ra, offA[q], wbase_a + q * 1024u);
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q) dma16(rb, offB[q], wbase_b + q * 1024u);
    };
    auto advance = [&]() {
#pragma unroll
        for (int q = 0; q < A_INSTR; ++q) offA[q] += BK * 4;
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q) offB[q] += BK * 4;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float rowsum[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rowsum[i] = 0.f;
    const bool bias_owner = (k0 == 0) && (wn == 0);  // these waves also produce the bias-gradient column
    const int rows_left = p.N - n0 - wm * MT * 32;
    const bool full_rows = rows_left > (MT - 1) * 32, some_rows = rows_left > 0;

    const int sw = (li >> 1) & 7;
    issue(0);
    dma_wait();
    __syncthreads();
    // flat sequence of (period, k tile): the copy of the next tile crosses period boundaries by moving the two buffer
    // descriptors to the next period's operands and rewinding the offsets, so the pipeline never drains
    const int total = nt * n_per;
    const int group = nt * (p.flush_periods > 0 ? p.flush_periods : n_per);  // tiles between two slab updates
    int t_in = 0, period = p_first;
    auto next_tile = [&]() {  // moves descriptors / offsets to tile t+1 (possibly the first tile of the next period)
        if (++t_in == nt) {
            t_in = 0;
            ++period;
            ra = make_desc(p.dY + period * p.pstride_dy, (int64_t)p.N * p.ldb);
            rb = make_desc(p.X + period * p.pstride_x, (int64_t)p.K * p.ldb);
#pragma unroll
            for (int q = 0; q < A_INSTR; ++q) offA[q] -= (nt - 1) * BK * 4;
#pragma unroll
            for (int q = 0; q < B_INSTR; ++q) offB[q] -= (nt - 1) * BK * 4;
        } else {
            advance();
        }
    };
    // threads per output row, rounded up to a power of two (BN = 448: 112 -> 128, the last 16 threads of a row group idle)
    constexpr int TPR_USED = BN / 4;
    constexpr int TPR = TPR_USED <= 16 ? 16 : TPR_USED <= 32 ? 32 : TPR_USED <= 64 ? 64 : 128;
    constexpr int RPI = NTHREADS / TPR, ITER = PASS_ROWS / RPI;
    static_assert(TPR_USED <= 128 && PASS_ROWS % RPI == 0, "epilogue row groups must tile the pass");
    const int tt = threadIdx.x;
    const bool col_thread = (tt % TPR) < TPR_USED;
    const int col = k0 + (tt % TPR) * 4;
    float* cs = lds;
    auto flush = [&]() {  // slab += accumulators (all waves; uses the whole staging area, so no copy may be in flight)
    // bias gradient: lanes l and l+32 hold the two k halves of row (wm*MT + i)*32 + li
        if (bias_owner) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float s = rowsum[i] + __shfl_xor(rowsum[i], 32);
                const int row = n0 + (wm * MT + i) * 32 + li;
                if (h == 0 && row < p.N) slab[(int64_t)row * p.lds_ + p.K] += s;
            }
        }

        // slab += tile, staged through LDS (whole-row float4 read-add-write)
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            if (PASSES == 1 || wm == pass) {
                const int wrow = (PASSES == 1) ? wm * MT * 32 : 0;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cs[(wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC + (wn * NT + j) * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int row_l = tt / TPR + RPI * it;
                const int row = n0 + pass * PASS_ROWS + row_l;
                if (row >= p.N || col >= p.K || !col_thread) continue;
                const float4 y = *reinterpret_cast<const float4*>(cs + row_l * LDC + (tt % TPR) * 4);
                float* dst = slab + (int64_t)row * p.lds_ + col;
                if (col + 3 < p.K && (p.lds_ & 3) == 0) {
                    float4 o = *reinterpret_cast<const float4*>(dst);
                    o.x += y.x; o.y += y.y; o.z += y.z; o.w += y.w;
                    *reinterpret_cast<float4*>(dst) = o;
                } else {
                    dst[0] += y.x;
                    if (col + 1 < p.K) dst[1] += y.y;
                    if (col + 2 < p.K) dst[2] += y.z;
                    if (col + 3 < p.K) dst[3] += y.w;
                }
            }
            if (pass + 1 < PASSES) __syncthreads();
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            rowsum[i] = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    };
    for (int t = 0; t < total; ++t) {
        const int cur = t & 1;
        const bool boundary = (t + 1) % group == 0 || t + 1 == total;  // this tile ends an accumulation group
        if (!boundary) {
            next_tile();
            issue(cur ^ 1);
        }
        const float* a_base = lds + cur * STAGE + (wm * MT * 32 + li) * BK;
        const float* b_base = lds + cur * STAGE + A_FLOATS + (wn * NT * 32 + li) * BK;
        auto body = [&](auto mtv_c) {
            constexpr int MTV = decltype(mtv_c)::value;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float a[MTV][8], b[NT][8];
#pragma unroll
                for (int i = 0; i < MTV; ++i) {
                    const float4 lo = *reinterpret_cast<const float4*>(a_base + i * 32 * BK + (((g * 4 + h * 2) ^ sw) << 2));
                    const float4 hi = *reinterpret_cast<const float4*>(a_base + i * 32 * BK + (((g * 4 + h * 2 + 1) ^ sw) << 2));
                    a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = lo.z; a[i][3] = lo.w;
                    a[i][4] = hi.x; a[i][5] = hi.y; a[i][6] = hi.z; a[i][7] = hi.w;
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float4 lo = *reinterpret_cast<const float4*>(b_base + j * 32 * BK + (((g * 4 + h * 2) ^ sw) << 2));
                    const float4 hi = *reinterpret_cast<const float4*>(b_base + j * 32 * BK + (((g * 4 + h * 2 + 1) ^ sw) << 2));
                    b[j][0] = lo.x; b[j][1] = lo.y; b[j][2] = lo.z; b[j][3] = lo.w;
                    b[j][4] = hi.x; b[j][5] = hi.y; b[j][6] = hi.z; b[j][7] = hi.w;
                }
                if (bias_owner) {
#pragma unroll
                    for (int i = 0; i < MTV; ++i)
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) rowsum[i] += a[i][kk];
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int i = 0; i < MTV; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            }
        };
        if constexpr (SKIP_ROWS && MT >= 2) {
            if (full_rows) body(std::integral_constant<int, MT>{});
            else if (some_rows) body(std::integral_constant<int, MT - 1>{});
        } else {
            body(std::integral_constant<int, MT>{});
        }
        dma_wait();
        __syncthreads();
        if (boundary) {
            flush();
            if (t + 1 < total) {  // re-prime the pipeline after the slab update
                next_tile();
                issue(cur ^ 1);
                dma_wait();
                __syncthreads();
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad for the SMALL layers (N <= 32, K <= 32; the whole-horizon rollout of the 32-wide policies contracts over
// T*ldb ~ millions of columns): one 32 x 32 accumulator per WAVE over the wave's own column range, four independent
// waves per workgroup (wave-private LDS tiles, no workgroup barrier).  HBM-bound by construction: every dY / X element is
// read exactly once (8 KiB per 16 MFMAs).  slab split index = global wave id; bias column from the dY fragment sums.
// ---------------------------------------------------------------------------------------------------------------
// KC = 32-column chunks of the output (K <= 32 KC input features of X): one accumulator and one wave-private B tile each
template <int KC>
__global__ __launch_bounds__(kThreads) void wgrad_small_kernel(WgParams p, int n_splits) {
    __shared__ __attribute__((aligned(16))) float lds[4 * (1 + KC) * 32 * LDA_S];  // per wave: A tile then KC B tiles, [32][36]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int split = blockIdx.x * 4 + wave;
    if (split >= n_splits) return;  // (tiles are wave-private: no workgroup barrier below)
    const int b_begin = split * p.chunk;
    const int b_end = min(b_begin + p.chunk, p.nB);
    float* As = lds + wave * ((1 + KC) * 32 * LDA_S);
    float* Bs = As + 32 * LDA_S;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.dY, (int64_t)p.N * p.ldb);  // rows >= N / K read as zeros
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.X, (int64_t)p.K * p.ldb);
    // tile = 32 rows x 32 columns; lane l moves rows (l >> 3) + 8 q, columns (l & 7) * 4 .. +3
    int off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = (int)((((int64_t)(lane >> 3) + 8 * q) * p.ldb + b_begin + (lane & 7) * 4) * 4);
    const int chunk_step = (int)(32 * p.ldb * 4);  // byte offset between two 32-row chunks of X

    f32x16 acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float rowsum = 0.f;
    float4 va[4], vb[KC][4];
    const int nt = (b_end - b_begin + BK - 1) / BK;
    if (nt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            va[q] = buf_load4(ra, off[q]);
#pragma unroll
            for (int c = 0; c < KC; ++c) vb[c][q] = buf_load4(rb, off[q] + c * chunk_step);
        }
    }
    for (int t = 0; t < nt; ++t) {
        // columns at or past b_end contribute nothing (only the last tile of the last split can be partial)
        const int cc = b_begin + t * BK + (lane & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 x = va[q];
            const bool tail = cc + 3 >= b_end;
            if (tail) {
                if (cc + 0 >= b_end) x.x = 0.f;
                if (cc + 1 >= b_end) x.y = 0.f;
                if (cc + 2 >= b_end) x.z = 0.f;
                if (cc + 3 >= b_end) x.w = 0.f;
            }
            *reinterpret_cast<float4*>(As + ((lane >> 3) + 8 * q) * LDA_S + (lane & 7) * 4) = x;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float4 y = vb[c][q];
                if (tail) {
                    if (cc + 0 >= b_end) y.x = 0.f;
                    if (cc + 1 >= b_end) y.y = 0.f;
                    if (cc + 2 >= b_end) y.z = 0.f;
                    if (cc + 3 >= b_end) y.w = 0.f;
                }
                *reinterpret_cast<float4*>(Bs + c * 32 * LDA_S + ((lane >> 3) + 8 * q) * LDA_S + (lane & 7) * 4) = y;
            }
        }
        if (t + 1 < nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                off[q] += BK * 4;
                va[q] = buf_load4(ra, off[q]);
#pragma unroll
                for (int c = 0; c < KC; ++c) vb[c][q] = buf_load4(rb, off[q] + c * chunk_step);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed (tiles are wave-private)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float a[8];
            read_frag8(As + li * LDA_S, g, h, a);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) rowsum += a[kk];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float b[8];
                read_frag8(Bs + c * 32 * LDA_S + li * LDA_S, g, h, b);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc[c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();  // fragment reads done before the next tile overwrites the LDS tiles
    }
    float* slab = p.slab + (int64_t)split * p.N * p.lds_;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int col = c * 32 + li;
            if (row < p.N && col < p.K) slab[(int64_t)row * p.lds_ + col] += acc[c][r];
        }
    const float s = rowsum + __shfl_xor(rowsum, 32);
    if (h == 0 && li < p.N) slab[(int64_t)li * p.lds_ + p.K] += s;
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

// fast (buffer-load) path: 16-byte aligned operands, row strides multiples of 4 floats, buffers below 2 GiB
bool wx_fast_ok(const WxParams& p) {
    return p.lda % 4 == 0 && p.ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(p.Bm) & 15) == 0 && (int64_t)p.M * p.lda < (1ll << 28) &&
           (int64_t)p.K * p.ldb < (1ll << 28);
}

#ifdef NIC_TUNING_BUILD
int wx_lds_pad() {  // NIC_WX_LDS_PAD: unused dynamic LDS per workgroup = fewer co-resident workgroups per CU (occupancy experiments)
    const char* e = getenv("NIC_WX_LDS_PAD");
    return e ? atoi(e) : 0;
}
#else
constexpr int wx_lds_pad() { return 0; }
#endif

template <int WM, int WN, int MT, int NT, int EPI>
void launch_wx_dma(const WxParams& p, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.ncols + BN - 1) / BN;
    nic::note_kernelf("gemm_wx_dma_kernel<%d,%d,%d,%d,%s>", WM, WN, MT, NT, epi_name(EPI));
    hipLaunchKernelGGL((gemm_wx_dma_kernel<WM, WN, MT, NT, EPI>), dim3(tiles_m * tiles_n), dim3(64 * WM * WN), wx_lds_pad(), s, p);
}

template <int WM, int WN, int MT, int NT, int EPI>
void launch_wx(const WxParams& p, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.ncols + BN - 1) / BN;
    nic::note_kernelf("gemm_wx_kernel<%d,%d,%d,%d,%s>", WM, WN, MT, NT, epi_name(EPI));
    hipLaunchKernelGGL((gemm_wx_kernel<WM, WN, MT, NT, EPI>), dim3(tiles_m * tiles_n), dim3(kThreads), 0, s, p);
}

// A/B variants of the dispatch (1 = 128 x 128 tiles for the big layers, 2 = register-staged wgrad, 3 = wgrad without the XCD
// tile order) exist only in the tuning build tools/gemm_probe.py makes for itself (-DNIC_TUNING_BUILD, a separate .so
// outside the package); the product library has no environment switches: every variant below is a correct kernel, but
// which one runs must not depend on a stray variable.
#ifdef NIC_TUNING_BUILD
extern "C" int nic_tuning_set_stamps(void* buf) {  // buf: device memory, 8 x u64 per workgroup of the next launches (or null)
    unsigned long long* pbuf = static_cast<unsigned long long*>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nic_stamps), &pbuf, sizeof(pbuf));
}
int tune_flags() {  // NIC_GEMM_TUNE: timing experiments inside the wx kernel (see WxParams::tune); read at every launch
    const char* e = getenv("NIC_GEMM_TUNE");
    return e ? atoi(e) : 0;
}
int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NIC_GEMM_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}
#else
constexpr int gemm_variant() { return 0; }
constexpr int tune_flags() { return 0; }
#endif

// Tile shape of the LDS-DMA wx kernel.  Round 4 measured seventeen tilings from 1,024 to 65,536 scenarios
// (tools/gemm_tile_probe.py, profiles/r04_gemm_tile_probe*.json) and what decides is NOT bytes of LDS traffic per flop but how
// many workgroups a CU holds at once: a lone 256 x 256 workgroup (128 KB of LDS, one per CU) spends ~0.5-1.2 us of every k tile
// in the copy's round trip + the barrier with nothing else to issue, while two co-resident 128 x 128 workgroups of 8 waves
// (64 KB each, 65-90 VGPRs) cover each other's stalls: 512 x 512 x 65,536 dgrad 278 -> 262 us (0.786 -> 0.833 of the FP32 MFMA
// peak), forward 267 = 267; at 8,192 scenarios (cfg3's shard on 8 GPUs) 73 -> 38-40 us, at 1,024 (the reference's shipped
// batch) 70 -> 14 us.  Round 3's picker counted whole rounds of 256 CUs only, so every launch of fewer than 256 tiles "cost one
// round" whatever the tile and ran as 16-128 big workgroups.  Three tilings remain:
//   128 x 128, 8 waves (wave tile 64 x 32), 2 per CU   launches that fill both slots of every CU (>= 512 tiles)
//    64 x 128, 8 waves (wave tile 32 x 32), 3 per CU   smaller launches with >= 33 output rows
//    32 x 128, 4 waves (wave tile 32 x 32), 4 per CU   thin outputs (17 logits rows) and ragged heights (393 rows: 13 row
//                                                      blocks of 32 instead of 4 of 128 - measured 229 us against 252 for
//                                                      round 3's 448-row tile)
// chosen by rows computed per row of the matrix (padding to the block height) x the measured relative cost of the tiling.
// The 256 x 256 tiling only exists in tuning builds (A/B reference of the probes).
enum WxTile { WX_128x128W8, WX_64x128W8, WX_32x128, WX_256x256, WX_N };
int pick_wx_tile(int M, int ncols) {
    auto padded = [&](int bm) { return (double)((M + bm - 1) / bm * bm) / M; };
    const int64_t tiles128 = (int64_t)((M + 127) / 128) * ((ncols + 127) / 128);
    double c128 = padded(128), c64 = padded(64) * 1.02, c32 = padded(32) * 1.06;   // (512 x 512 x 65,536: 262 / 265 / 281 us)
    if (tiles128 < 2 * nic::cu_count()) {   // fewer than two workgroups per CU: smaller tiles put more wavefronts on a CU
        c128 = 1e30;
        c64 = padded(64);
        c32 = padded(32) * 1.02;
    }
    if (c128 <= c64 && c128 <= c32) return WX_128x128W8;
    return c64 <= c32 ? WX_64x128W8 : WX_32x128;
}

#ifdef NIC_TUNING_BUILD
int forced_wx_tile() {  // NIC_WX_TILE: tiling id for every wx launch (tools/gemm_tile_probe.py); read at every launch
    const char* e = getenv("NIC_WX_TILE");
    return e ? atoi(e) : -1;
}
#else
constexpr int forced_wx_tile() { return -1; }
#endif

// Streamed form (gemm_wx_stream_kernel) for launches of at most 1,024 output tiles of 32 x 32: NT = 1 (every tile its own
// wavefront), KS = wavefronts that split the contraction, so that the launch has ~2,048 wavefronts where K allows.
// Returns 0 (LDS-DMA kernels) or KS.  Measured (tools/gemm_tile_probe.py --stream, profiles/r04_gemm_stream_probe.json; us per
// launch, LDS-DMA -> streamed): 512 x 512 x 1,024 scenarios 21.7 -> 10.5, x 2,048 22.8 -> 15.7, x 4,096 25.0 -> 29.4 (not taken);
// logits 17 x 512 13.2 -> 7.5 up to 8,192 scenarios, 14.5 -> 9.3 at 16,384; first-layer input gradient 51 x 512 20.6 -> 7.6.
int pick_wx_stream(int M, int K, int ncols) {
    const int64_t tiles = (int64_t)((M + 31) / 32) * ((ncols + 31) / 32);
    const int cus = nic::cu_count();
    if (tiles > 4 * cus || K < 128) return 0;   // (a short contraction is 2-4 k tiles of the LDS pipeline: nothing to gain, measured)
    int ks = 1;
    while (ks < 4 && tiles * ks * 2 <= 8 * cus && K >= 64 * ks * 2) ks *= 2;
    return ks;
}

template <int NT, int KS, int EPI>
void launch_wx_stream(const WxParams& p, hipStream_t s) {
    const int tiles = ((p.M + 31) / 32) * ((p.ncols + 32 * NT - 1) / (32 * NT));
    nic::note_kernelf("gemm_wx_stream_kernel<%d,%d,%s>", NT, KS, epi_name(EPI));
    hipLaunchKernelGGL((gemm_wx_stream_kernel<NT, KS, EPI>), dim3(tiles), dim3(64 * KS), 0, s, p);
}

#ifdef NIC_TUNING_BUILD
int forced_wx_stream() {  // NIC_WX_STREAM: 0 = never, 10 * NT + KS = that instantiation for every wx launch; unset = the picker
    const char* e = getenv("NIC_WX_STREAM");
    return e ? atoi(e) : -1;
}
#else
constexpr int forced_wx_stream() { return -1; }
#endif

template <int EPI>
void dispatch_wx(const WxParams& p, hipStream_t s) {
    if (wx_fast_ok(p)) {
        int stream = 10 + pick_wx_stream(p.M, p.K, p.ncols);
        if (stream == 10) stream = 0;
        if (forced_wx_stream() >= 0) stream = forced_wx_stream();
        switch (stream) {
            case 11: launch_wx_stream<1, 1, EPI>(p, s); return;
            case 12: launch_wx_stream<1, 2, EPI>(p, s); return;
            case 14: launch_wx_stream<1, 4, EPI>(p, s); return;
#ifdef NIC_TUNING_BUILD
            case 21: launch_wx_stream<2, 1, EPI>(p, s); return;
            case 22: launch_wx_stream<2, 2, EPI>(p, s); return;
            case 24: launch_wx_stream<2, 4, EPI>(p, s); return;
#endif
            default: break;
        }
        // production path: LDS-DMA kernels
        int tile = pick_wx_tile(p.M, p.ncols);
#ifdef NIC_TUNING_BUILD
        if (p.tune & 128) tile = WX_256x256;
        if (forced_wx_tile() >= 0 && forced_wx_tile() < WX_N) tile = forced_wx_tile();
        if (tile == WX_256x256) { launch_wx_dma<2, 4, 4, 2, EPI>(p, s); return; }
#endif
        switch (tile) {
            case WX_128x128W8: launch_wx_dma<2, 4, 2, 1, EPI>(p, s); return;
            case WX_64x128W8: launch_wx_dma<2, 4, 1, 1, EPI>(p, s); return;
            default: launch_wx_dma<1, 4, 1, 1, EPI>(p, s); return;
        }
    }
    if (p.M > 64) launch_wx<2, 2, 2, 2, EPI>(p, s);        // unaligned operands: guarded register-staged kernels
    else if (p.M > 32) launch_wx<1, 4, 2, 1, EPI>(p, s);
    else launch_wx<1, 4, 1, 2, EPI>(p, s);
}

template <int WM, int WN, int MT, int NT>
void launch_wg(const WgParams& p, int n_splits, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    dim3 grid(((p.K + 1 + BN - 1) / BN) * ((p.N + BM - 1) / BM) * n_splits);
    const bool fast = p.ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(p.dY) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.X) & 15) == 0 && (int64_t)p.N * p.ldb < (1ll << 28) &&
                      (int64_t)p.K * p.ldb < (1ll << 28);
    nic::note_kernelf("gemm_wgrad_kernel<%d,%d,%d,%d,%d>", WM, WN, MT, NT, fast ? 1 : 0);
    if (fast) hipLaunchKernelGGL((gemm_wgrad_kernel<WM, WN, MT, NT, 1>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((gemm_wgrad_kernel<WM, WN, MT, NT, 0>), grid, dim3(kThreads), 0, s, p);
}

// the tile shape is a function of (N, K) only so that nic_wgrad_num_splits and the launch agree
void wgrad_tile(int N, int K, int* bm, int* bn) {
    if (N > 64) { *bm = 128; *bn = (K + 1 > 64) ? 128 : 64; }
    else if (N > 32) { *bm = 64; *bn = 128; }
    else { *bm = 32; *bn = 256; }
}

// big layers: 256 x 256 LDS-DMA tiles, one workgroup per CU
bool wgrad_big(int N, int K) { return N >= 192 && K >= 129; }
// N >= 192 output rows over 65..128 input rows (the shipped many-warehouse setting's first layer: 512 x 66): 256 x 128 tiles on
// the LDS-DMA pipeline (round 4; the register-staged 128 x 128 kernel ran that layer's all-period gradient at 0.02 of peak)
bool wgrad_half(int N, int K) { return N >= 192 && K > 64 && K <= 128; }
bool wgrad_dma_shape(int N, int K);
bool wgrad_tall(int N, int K);
bool wgrad_wide(int N, int K);
bool wgrad_mid(int N, int K);

// tall, narrow layers (the first layer: 512 x 51): ONE 512 x 64 output tile per scenario chunk, so dZ is read exactly once
// (the 128 x 64 register-staged tiles read it 1.3 x) by the LDS-DMA pipeline; HBM-bound
bool wgrad_tall(int N, int K) { return N >= 384 && K <= 64; }

template <int WM, int WN, int MT, int NT, bool SKIP = false>
void launch_wg_dma(const WgParams& p, int n_splits, hipStream_t s) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    dim3 grid(((p.K + BN - 1) / BN) * ((p.N + BM - 1) / BM) * n_splits);
    nic::note_kernelf(SKIP ? "gemm_wgrad_dma_kernel<%d,%d,%d,%d,skip>" : "gemm_wgrad_dma_kernel<%d,%d,%d,%d>", WM, WN, MT, NT);
    hipLaunchKernelGGL((gemm_wgrad_dma_kernel<WM, WN, MT, NT, SKIP>), grid, dim3(64 * WM * WN), 0, s, p);
}
// K in (256, 448] (cfg5's first layer: 393 input rows, 295 of them live): ONE 320 / 384 / 448-column tile covers it (88 % of the
// tile's columns used at K = 393) where two 256-column tiles compute 512 (77 %); 128 x 448 keeps the 112 accumulator registers
// of the 7-tile wx kernel
bool wgrad_wide(int N, int K) { return wgrad_big(N, K) && K > 256 && K <= 448 && N % 128 == 0; }
int wgrad_wide_nt(int K) { return (K + 63) / 64; }   // 5, 6 or 7 column tiles of 32 per wave column: 320 / 384 / 448 columns

// 96 <= N <= 128 output rows over a wide input (cfg5's compacted logits layer, 98 x 512): 128 x 256 tiles on the LDS-DMA pipeline
// (the register-staged 128 x 128 kernel reaches 0.37 of peak there)
bool wgrad_mid(int N, int K) { return N >= 96 && N <= 128 && K >= 192; }

bool wgrad_dma_shape(int N, int K) { return wgrad_big(N, K) || wgrad_tall(N, K) || wgrad_mid(N, K) || wgrad_half(N, K); }
// output tiles of one scenario chunk under the LDS-DMA kernel launch_wg_dma_for picks (what the split counts divide 256 by)
int wgrad_dma_tiles(int N, int K) {
    if (wgrad_tall(N, K)) return (N + 511) / 512;
    if (wgrad_half(N, K)) return (N + 255) / 256;
    if (wgrad_mid(N, K)) return (K + 255) / 256;
    if (wgrad_wide(N, K)) return (N + 127) / 128;
    return ((N + 127) / 128) * ((K + 255) / 256);
}

// the LDS-DMA weight-gradient kernel for a shape (wgrad_dma_shape)
void launch_wg_dma_for(const WgParams& p, int n_splits, hipStream_t s) {
    if (wgrad_tall(p.N, p.K)) launch_wg_dma<8, 1, 2, 2>(p, n_splits, s);
    else if (wgrad_half(p.N, p.K)) launch_wg_dma<4, 2, 2, 2>(p, n_splits, s);
    else if (wgrad_mid(p.N, p.K)) launch_wg_dma<2, 4, 2, 2>(p, n_splits, s);
    else if (wgrad_wide(p.N, p.K)) {
        const int nt = wgrad_wide_nt(p.K);
        if (nt == 5) launch_wg_dma<4, 2, 1, 5>(p, n_splits, s);
        else if (nt == 6) launch_wg_dma<4, 2, 1, 6>(p, n_splits, s);
        else launch_wg_dma<4, 2, 1, 7>(p, n_splits, s);
    }
    // big layers: 128 x 256 tiles (round 4; wave tile 64 x 64 = 64 accumulator registers, no scratch).  Measured against round 3's
    // 256 x 256 tile (128 accumulators, 516 B of scratch) on 512 x 512 x 16,384 x T=50: 133.3-133.9 against 131.8 TFLOP/s
    // (profiles/r04_gemm_stagger_and_wgrad_tile_probe.json)
    else launch_wg_dma<2, 4, 2, 2>(p, n_splits, s);
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
    WxParams p{W, ldw, X, Y, bias, nullptr, N, K, (n_scenarios + 3) / 4 * 4, ldb, act, 0, tune_flags()};
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
    WxParams p{Wt, ldwt, dY, dX, nullptr, Hprev, K, N, (n_scenarios + 3) / 4 * 4, ldb, act_prev, accumulate, tune_flags()};
    dispatch_wx<EPI_DGRAD>(p, nic::as_stream(stream));
    return nic::check_launch("nic_linear_dgrad");
}

int nic_wgrad_num_splits(int32_t N, int32_t K, int32_t n_scenarios) {
    if (N <= 0 || K <= 0 || n_scenarios <= 0) return 0;
    int bm, bn;
    wgrad_tile(N, K, &bm, &bn);
    int tiles = ((N + bm - 1) / bm) * ((K + 1 + bn - 1) / bn);
    int target = 4 * nic::cu_count();                  // ~4 workgroups per CU in total
    if (wgrad_dma_shape(N, K)) {                       // LDS-DMA tiles, one workgroup per CU, one round
        tiles = wgrad_dma_tiles(N, K);
        target = nic::cu_count();
    }
    if (N <= 32 && (K <= 32 || (K <= 128 && K % 32 != 0))) {  // wgrad_small_kernel: one split per wave, >= 2048 columns each
        int sp = (int)(((int64_t)n_scenarios + 2047) / 2048);
        if (sp > 1024) sp = 1024;  // one wave per SIMD
        return (sp + 3) / 4 * 4;
    }
    if (N <= NIC_THIN_MAX_ROWS && K % 32 == 0) {  // nic_linear_bwd_thin: one wave per (split, 32-row chunk), 2 waves per SIMD
        int sp = 2048 / (K / 32);
        const int cap = (n_scenarios + 63) / 64;  // at least one 64-scenario block per split
        if (sp > cap) sp = cap;
        return sp < 1 ? 1 : sp;
    }
    int splits = (target + tiles - 1) / tiles;
    const int max_splits = (n_scenarios + 255) / 256;  // at least 256 scenarios (8 k-tiles) per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

// Slab slots of the all-period contraction (nic_linear_wgrad_periods) as (scenario splits x period groups): scenario chunks go
// down to 128 scenarios (4 k tiles per period), what is still missing to fill the chip comes from splitting the horizon.
static void wgrad_period_factors(int32_t n_slots, int32_t n_scenarios, int32_t n_periods, int* scen_splits, int* groups) {
    int cap = n_scenarios / 128;
    if (cap < 1) cap = 1;
    int ss = n_slots < cap ? n_slots : cap;
    if (ss < 1) ss = 1;
    int g = n_slots / ss;
    if (g > n_periods) g = n_periods;
    if (g < 1) g = 1;
    *scen_splits = ss;
    *groups = g;
}

int nic_wgrad_periods_num_splits(int32_t N, int32_t K, int32_t n_scenarios, int32_t n_periods) {
    if (N <= 0 || K <= 0 || n_scenarios <= 0 || n_periods <= 0) return 0;
    if (N <= 32 && K <= 32) return nic_wgrad_num_splits(N, K, n_scenarios);   // wgrad_small_kernel: one launch per period
    if (!wgrad_dma_shape(N, K) || n_scenarios % BK != 0) {
        // register-staged kernels (ragged scenario counts, narrow layers: the real-data batches of 72-288 products x 95 weeks):
        // the same (period group x scenario split) slots - ONE launch instead of a serial walk over the horizon by 1-4 workgroups
        int bm, bn;
        wgrad_tile(N, K, &bm, &bn);
        const int tiles = ((N + bm - 1) / bm) * ((K + 1 + bn - 1) / bn);
        int ss, g;
        wgrad_period_factors((4 * nic::cu_count() + tiles - 1) / tiles, n_scenarios, n_periods, &ss, &g);
        const int base = nic_wgrad_num_splits(N, K, n_scenarios);
        return ss * g > base ? ss * g : base;
    }
    const int tiles = wgrad_dma_tiles(N, K);
    const int slots = (nic::cu_count() + tiles - 1) / tiles;   // one workgroup per CU, one round
    int ss, g;
    wgrad_period_factors(slots, n_scenarios, n_periods, &ss, &g);
    return ss * g;
}

// argument checks + dispatch shared by nic_linear_wgrad (one period) and nic_linear_wgrad_periods
static int wgrad_generic(const float* dY, const float* X, float* slab, int64_t lds_, int32_t N, int32_t K, int32_t n_scenarios,
                         int32_t ldb, int32_t n_splits, int32_t n_periods, int64_t pstride_dy, int64_t pstride_x, void* stream,
                         const char* who, int32_t scen_splits = 0, int32_t periods_per_group = 0) {
    NIC_REQUIRE(dY && X && slab, "%s: null buffer", who);
    NIC_REQUIRE(N > 0 && K > 0 && lds_ >= K + 1, "%s: bad N/K/lds (%d/%d/%lld)", who, N, K, (long long)lds_);
    NIC_REQUIRE(n_splits >= 1, "%s: n_splits must be >= 1", who);
    if (int e = require_ld(who, n_scenarios, ldb)) return e;
    // (scen_splits > 0: the n_splits slab slots are period groups x scen_splits scenario splits, register-staged kernels only)
    int chunk = (n_scenarios + (scen_splits > 0 ? scen_splits : n_splits) - 1) / (scen_splits > 0 ? scen_splits : n_splits);
    chunk = (chunk + BK - 1) / BK * BK;
    WgParams p{dY, X, slab, lds_, ldb, N, K, n_scenarios, chunk, gemm_variant() == 3 ? 0 : 1, n_periods, pstride_dy, pstride_x, 0,
               scen_splits, periods_per_group};
    hipStream_t s = nic::as_stream(stream);
    int bm, bn;
    wgrad_tile(N, K, &bm, &bn);
    const bool dma_ok = ldb % 4 == 0 && n_scenarios % BK == 0 && lds_ % 4 == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 &&
                        (int64_t)N * ldb < (1ll << 28) && (int64_t)K * ldb < (1ll << 28);
    const bool small_ok = n_periods == 1 && N <= 32 && K <= 128 && ldb % 4 == 0 &&
                          (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                          (int64_t)N * ldb < (1ll << 28) && (int64_t)K * ldb < (1ll << 28);
    if (small_ok) {  // thin output (<= 32 rows), up to 128 input features: one wave per split, 1..4 accumulators
        const dim3 g((n_splits + 3) / 4), b(kThreads);
        nic::note_kernelf("wgrad_small_kernel<%d>", K <= 32 ? 1 : (K <= 64 ? 2 : (K <= 96 ? 3 : 4)));
        if (K <= 32) hipLaunchKernelGGL(wgrad_small_kernel<1>, g, b, 0, s, p, n_splits);
        else if (K <= 64) hipLaunchKernelGGL(wgrad_small_kernel<2>, g, b, 0, s, p, n_splits);
        else if (K <= 96) hipLaunchKernelGGL(wgrad_small_kernel<3>, g, b, 0, s, p, n_splits);
        else hipLaunchKernelGGL(wgrad_small_kernel<4>, g, b, 0, s, p, n_splits);
    }
    else if (wgrad_dma_shape(N, K) && dma_ok && gemm_variant() != 2 && scen_splits == 0) launch_wg_dma_for(p, n_splits, s);
    else if (bm == 128 && bn == 128) launch_wg<2, 2, 2, 2>(p, n_splits, s);
    else if (bm == 128) launch_wg<2, 2, 2, 1>(p, n_splits, s);
    else if (bm == 64) launch_wg<1, 4, 2, 1>(p, n_splits, s);
    else launch_wg<1, 4, 1, 2>(p, n_splits, s);
    return nic::check_launch(who);
}

int nic_linear_wgrad(const float* dY, const float* X, float* slab, int64_t lds_, int32_t N, int32_t K, int32_t n_scenarios,
                     int32_t ldb, int32_t n_splits, void* stream) {
    return wgrad_generic(dY, X, slab, lds_, N, K, n_scenarios, ldb, n_splits, 1, 0, 0, stream, "nic_linear_wgrad");
}

int nic_linear_wgrad_periods(const float* dY, const float* X, float* slab, int64_t lds_, int32_t N, int32_t K,
                             int32_t n_scenarios, int32_t ldb, int32_t n_splits, int32_t n_periods, int64_t period_stride_dy,
                             int64_t period_stride_x, void* stream) {
    NIC_REQUIRE(n_periods >= 1, "nic_linear_wgrad_periods: n_periods must be >= 1");
    // Periods are accumulated LAST FIRST, the order autograd accumulates them in the reference: pre-activation gradients grow
    // towards the start of the horizon (each period adds the costs it influences later), so this sums the small terms
    // before the large ones.  Measured at 512x3, T = 100: bias-gradient error against an fp64 evaluation 2.9e-5 in
    // ascending order, 1e-6 descending.
    if (n_periods > 1) {
        dY += (int64_t)(n_periods - 1) * period_stride_dy;
        X += (int64_t)(n_periods - 1) * period_stride_x;
        period_stride_dy = -period_stride_dy;
        period_stride_x = -period_stride_x;
    }
    NIC_REQUIRE(period_stride_dy % 4 == 0 && period_stride_x % 4 == 0,
                "nic_linear_wgrad_periods: period strides must be multiples of 4 elements (16-byte aligned operands)");
    const bool dma_ok = dY && X && slab && N > 0 && K > 0 && lds_ >= K + 1 && n_splits >= 1 && n_scenarios > 0 && ldb >= n_scenarios &&
                        wgrad_dma_shape(N, K) && gemm_variant() != 2 && ldb % 4 == 0 && n_scenarios % BK == 0 && lds_ % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (int64_t)N * ldb < (1ll << 28) &&
                        (int64_t)K * ldb < (1ll << 28);
    if (!dma_ok) {  // other shapes: the register-staged kernels loop over the periods themselves; tiny layers one launch each
        const bool small = N <= 32 && K <= 32;
        if (small || n_periods == 1) {
            for (int t = 0; t < n_periods; ++t)
                if (int e = nic_linear_wgrad(dY + t * period_stride_dy, X + t * period_stride_x, slab, lds_, N, K, n_scenarios,
                                             ldb, n_splits, stream))
                    return e;
            return 0;
        }
        // register-staged kernels.  With enough slab slots: ONE launch, slot = (period group, scenario split), as long as no
        // accumulator sums more than ~8k terms (chunk x periods per group)
        {
            int ss, g;
            wgrad_period_factors(n_splits, n_scenarios, n_periods, &ss, &g);
            int chunk_s = (n_scenarios + ss - 1) / ss;
            chunk_s = (chunk_s + BK - 1) / BK * BK;
            const int ppg = (n_periods + g - 1) / g;
            if (g > 1 && (int64_t)chunk_s * ppg <= 8192)
                return wgrad_generic(dY, X, slab, lds_, N, K, n_scenarios, ldb, ss * g, n_periods, period_stride_dy, period_stride_x,
                                     stream, "nic_linear_wgrad_periods", ss, ppg);
        }
        // otherwise: a launch per group of periods, so that no accumulator sums more than ~8k terms
        // before it is added to the slab (see WgParams::flush_periods)
        int chunk_g = (n_scenarios + n_splits - 1) / n_splits;
        const int group = 8192 / (chunk_g > 0 ? chunk_g : 1) > 0 ? 8192 / (chunk_g > 0 ? chunk_g : 1) : 1;
        for (int t = 0; t < n_periods; t += group) {
            const int n = n_periods - t < group ? n_periods - t : group;
            if (int e = wgrad_generic(dY + t * period_stride_dy, X + t * period_stride_x, slab, lds_, N, K, n_scenarios, ldb,
                                      n_splits, n, period_stride_dy, period_stride_x, stream, "nic_linear_wgrad_periods"))
                return e;
        }
        return 0;
    }
    int scen_splits, groups;
    wgrad_period_factors(n_splits, n_scenarios, n_periods, &scen_splits, &groups);
    int chunk = (n_scenarios + scen_splits - 1) / scen_splits;
    chunk = (chunk + BK - 1) / BK * BK;
    const int flush = 8192 / chunk > 0 ? 8192 / chunk : 1;
    const int ppg = (n_periods + groups - 1) / groups;
    WgParams p{dY, X, slab, lds_, ldb, N, K, n_scenarios, chunk, gemm_variant() == 3 ? 0 : 1, n_periods, period_stride_dy,
               period_stride_x, flush, scen_splits, ppg};
    launch_wg_dma_for(p, scen_splits * groups, nic::as_stream(stream));
    return nic::check_launch("nic_linear_wgrad_periods");
}

// many splits, few outputs (the small layers): one wavefront per output element, lanes stride over the splits
__global__ void wgrad_reduce_wide_kernel(const float* __restrict__ slab, int64_t lds_, int n_splits, float* __restrict__ dW,
                                         int64_t lddw, float* __restrict__ db, int N, int K, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (int64_t)N * (K + 1)) return;
    const int n = (int)(idx / (K + 1)), k = (int)(idx % (K + 1));
    float s = 0.f;
    for (int sp = lane; sp < n_splits; sp += 64) s += slab[((int64_t)sp * N + n) * lds_ + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        s *= scale;
        if (k < K) dW[(int64_t)n * lddw + k] = s;
        else if (db) db[n] = s;
    }
}

int nic_wgrad_reduce(const float* slab, int64_t lds_, int32_t n_splits, float* dW, int64_t lddw, float* db, int32_t N,
                     int32_t K, float scale, void* stream) {
    NIC_REQUIRE(slab && dW, "nic_wgrad_reduce: null buffer");
    NIC_REQUIRE(N > 0 && K > 0 && lds_ >= K + 1 && lddw >= K && n_splits >= 1, "nic_wgrad_reduce: bad sizes");
    const int64_t total = (int64_t)N * (K + 1);
    nic::note_kernel(n_splits >= 128 && total <= 65536 ? "wgrad_reduce_wide_kernel" : "wgrad_reduce_kernel");
    if (n_splits >= 128 && total <= 65536)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(nic::ceil_div(total, 4)), dim3(256), 0, nic::as_stream(stream), slab,
                           lds_, n_splits, dW, lddw, db, N, K, scale);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nic::ceil_div(total, 256)), dim3(256), 0, nic::as_stream(stream), slab, lds_,
                           n_splits, dW, lddw, db, N, K, scale);
    return nic::check_launch("nic_wgrad_reduce");
}
}
