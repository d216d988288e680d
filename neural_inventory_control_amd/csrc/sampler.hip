// Batched synthetic demand sampler on gfx950 (replaces the host numpy generators of data_handling.py:178-211 for
// throughput runs).  Counter-based Philox4x32-10: every normal / uniform is a pure function of
// (seed, GLOBAL scenario index, period, variate index), so the traces do not depend on the launch geometry nor on how
// scenarios are sharded across GPUs.  Output is written directly in the [T][S][ldb] scenario-minor layout
// the env-step kernel reads (one coalesced 256-B store per wave and (t, s)).  4*S bytes written per scenario-period.
// Parity with numpy's MT19937 stream is statistical, not bitwise.
//
// One lane = one (scenario, period).  Every Philox block is generated ONCE per lane (the first version regenerated the
// blocks of all earlier variates for every store: O(S^2 / 4) Philox calls per lane):
//   * equicorrelated normal (the reference's only covariance: rho * s_i * s_j off the diagonal, s_i^2 on it,
//     data_handling.py:194-201, 0 <= rho <= 1): d_s = mean_s + std_s * (sqrt(rho) * z_common + sqrt(1 - rho) * z_s) has exactly
//     that covariance and needs S + 1 normals and 3 flops per output — no factor matrix at all;
//   * general covariance (Cholesky factor L): z[0..S) kept in registers, rows of L staged in LDS once per workgroup and
//     read as wave-uniform broadcasts; S (S + 1) / 2 FMAs per lane;
//   * Poisson: one uniform per output, inversion by search in a per-store CDF table built once per workgroup in LDS
//     (no expf / divide per sample).
// The generator itself is the floor: a Philox4x32-10 block is 20 32x32->64-bit multiplies (quarter-rate on the VALU) for 4
// outputs, and Box-Muller adds a log, a sqrt and a sin/cos pair per 2 normals on the transcendental unit.
#include "nic_common.h"

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;  // one v_mad_u64_u32 each (hi and lo together)
        c = U4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
        k0 += W0;
        k1 += W1;
    }
    return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return (x + 0.5f) * 2.3283064365386963e-10f; }  // (0,1)

// four standard normals from one Philox block (two Box-Muller pairs).  Hardware log2 / sin / cos (v_log_f32, v_sin_f32 and
// v_cos_f32 take their argument in revolutions, so 2*pi*u needs no multiply): absolute error ~1e-6 on a unit-variance
// variate, far below the sampling noise of any statistic of the traces.
__device__ __forceinline__ void normal4(U4 r, float (&z)[4]) {
    const float r0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.x)));  // -2 ln u = -2 ln2 log2 u
    const float r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.z)));
    const float a0 = u01(r.y), a1 = u01(r.w);
    z[0] = r0 * __builtin_amdgcn_cosf(a0);
    z[1] = r0 * __builtin_amdgcn_sinf(a0);
    z[2] = r1 * __builtin_amdgcn_cosf(a1);
    z[3] = r1 * __builtin_amdgcn_sinf(a1);
}

constexpr int kBlock = 256;

// ---- equicorrelated normal ---------------------------------------------------------------------------------------------
// The common factors of four consecutive periods are the four normals of ONE block (counter = (scenario, t / 4, kCommon)),
// store s of period t is word s % 4 of block (scenario, t, s / 4) - every (seed, scenario, period, variate) maps to the same
// number regardless of the launch geometry.  One lane = one scenario x PER periods: PER = 4 amortises the common block
// (S / 4 + 1/4 blocks per period; best for few stores and many lanes), PER = 1 keeps more lanes in flight (S / 4 + 1).
constexpr uint32_t kCommon = 0xFFFFFFFFu;
template <int PER>
__global__ __launch_bounds__(kBlock) void sample_equicorrelated_kernel(float* __restrict__ out, int T, int S, int B, int64_t ldb,
                                                                       int64_t scenario_offset, uint32_t k0, uint32_t k1,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ std_, float a_common,
                                                                       float a_own, int clip) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int t0 = blockIdx.y * PER;
    if (b >= B) return;
    const uint64_t gb = (uint64_t)(b + scenario_offset);
    const uint32_t c0 = (uint32_t)gb, c1 = (uint32_t)(gb >> 32);
    float common[4];
    normal4(philox4x32_10(U4{c0, c1, (uint32_t)(t0 >> 2), kCommon}, k0, k1), common);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int t = t0 + u;
        if (t < T) {
            float* dst = out + (int64_t)t * S * ldb + b;
            const int w = t & 3;
            const float cm = a_common * (w == 0 ? common[0] : w == 1 ? common[1] : w == 2 ? common[2] : common[3]);
            for (int blk = 0; blk * 4 < S; ++blk) {
                float z[4];
                normal4(philox4x32_10(U4{c0, c1, (uint32_t)t, (uint32_t)blk}, k0, k1), z);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = blk * 4 + q;
                    if (s < S) {
                        float d = mean[s] + std_[s] * (cm + a_own * z[q]);
                        if (clip && d < 0.f) d = 0.f;
                        dst[(int64_t)s * ldb] = d;
                    }
                }
            }
        }
    }
}

// ---- ONE store: every word of a Philox block is used -----------------------------------------------------------------------
// With one store there is no common factor to share (a correlation has nothing to act on: d ~ N(mean, std^2) whatever rho is), and
// the form above spends a whole block (+ a quarter of the common one) on every output: 0.10-0.12 of the HBM write roofline at
// 10^6 scenarios x T = 100 (round 5) against 0.5 for 16 stores.  Here the four normals of block (scenario, t / 4, kOwn) are the
// demands of FOUR CONSECUTIVE PERIODS of the scenario, and a lane owns four consecutive scenarios: four blocks -> sixteen outputs,
// written as four 16-byte stores (one per period; a wavefront writes 1 KB contiguous per period).  The mapping (seed, global
// scenario, period) -> number is again independent of launch geometry and sharding.
constexpr uint32_t kOwn = 0xFFFFFFFEu;
__global__ __launch_bounds__(kBlock) void sample_one_store_kernel(float* __restrict__ out, int T, int B, int64_t ldb,
                                                                  int64_t scenario_offset, uint32_t k0, uint32_t k1,
                                                                  const float* __restrict__ mean_p, const float* __restrict__ std_p,
                                                                  int clip) {
    const float mean = mean_p[0], std_ = std_p[0];
    const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    const int tg = blockIdx.y;
    if (b >= B) return;
    float z[4][4];   // [scenario][period]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t gb = (uint64_t)(b + i + scenario_offset);
        normal4(philox4x32_10(U4{(uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)tg, kOwn}, k0, k1), z[i]);
    }
    const bool full = b + 3 < B;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = 4 * tg + u;
        if (t < T) {
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = mean + std_ * z[i][u];
                if (clip && d[i] < 0.f) d[i] = 0.f;
            }
            float* dst = out + (int64_t)t * ldb + b;
            if (full) {
                *reinterpret_cast<float4*>(dst) = make_float4(d[0], d[1], d[2], d[3]);
            } else {   // the batch's last, partial group of four: the padding columns stay as they are
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (b + i < B) dst[i] = d[i];
            }
        }
    }
}

// ---- general covariance: d = mean + L z, z in registers, L in LDS ---------------------------------------------------------
template <int SMAX>
__global__ __launch_bounds__(kBlock) void sample_cholesky_kernel(float* __restrict__ out, int T, int S, int B, int64_t ldb,
                                                                 int64_t scenario_offset, uint32_t k0, uint32_t k1,
                                                                 const float* __restrict__ mean, const float* __restrict__ chol,
                                                                 int clip) {
    __shared__ float L[SMAX * SMAX];
    for (int i = threadIdx.x; i < S * S; i += kBlock) L[(i / S) * SMAX + (i % S)] = chol[i];
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int t = blockIdx.y;
    if (b >= B) return;
    const uint64_t gb = (uint64_t)(b + scenario_offset);
    const uint32_t c0 = (uint32_t)gb, c1 = (uint32_t)(gb >> 32);
    float z[SMAX];
#pragma unroll
    for (int blk = 0; blk < SMAX / 4; ++blk) {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (blk * 4 < S) normal4(philox4x32_10(U4{c0, c1, (uint32_t)t, (uint32_t)blk}, k0, k1), q);
        z[blk * 4 + 0] = q[0]; z[blk * 4 + 1] = q[1]; z[blk * 4 + 2] = q[2]; z[blk * 4 + 3] = q[3];
    }
    float* dst = out + (int64_t)t * S * ldb + b;
    for (int s = 0; s < S; ++s) {
        const float* row = L + s * SMAX;  // wave-uniform address: LDS broadcast
        float acc = mean[s];
#pragma unroll
        for (int j = 0; j < SMAX; ++j)
            if (j <= s) acc += row[j] * z[j];
        if (clip && acc < 0.f) acc = 0.f;
        dst[(int64_t)s * ldb] = acc;
    }
}

// ---- Poisson: inversion by search in a per-store CDF table ----------------------------------------------------------------
constexpr int kCdf = 64;  // table entries per store: P(X <= k), k < 64 (lambda up to ~30: the tail beyond is < 1e-7)
__global__ __launch_bounds__(kBlock) void sample_poisson_kernel(float* __restrict__ out, int T, int S, int B, int64_t ldb,
                                                                int64_t scenario_offset, uint32_t k0, uint32_t k1,
                                                                const float* __restrict__ mean) {
    extern __shared__ float cdf[];  // [S][kCdf]
    for (int s = threadIdx.x; s < S; s += kBlock) {
        const float lam = mean[s];
        float p = expf(-lam), F = p;
        for (int k = 0; k < kCdf; ++k) {
            cdf[s * kCdf + k] = F;
            p *= lam / (float)(k + 1);
            F += p;
        }
    }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int t = blockIdx.y;
    if (b >= B) return;
    const uint64_t gb = (uint64_t)(b + scenario_offset);
    const uint32_t c0 = (uint32_t)gb, c1 = (uint32_t)(gb >> 32);
    float* dst = out + (int64_t)t * S * ldb + b;
    for (int blk = 0; blk * 4 < S; ++blk) {
        const U4 r = philox4x32_10(U4{c0, c1, (uint32_t)t, (uint32_t)blk}, k0, k1);
        const uint32_t bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = blk * 4 + q;
            if (s < S) {
                const float u = u01(bits[q]);
                const float* c = cdf + s * kCdf;
                int k = 0;
                while (k < kCdf - 1 && u > c[k]) ++k;  // same comparison sequence as sequential inversion
                float extra = 0.f;
                if (k == kCdf - 1 && u > c[k]) {  // beyond the table (lambda > ~30): continue the recurrence
                    const float lam = mean[s];
                    float F = c[k], p = c[k] - c[k - 1];
                    int kk = k;
                    while (u > F && kk < 1000) {
                        ++kk;
                        p *= lam / (float)kk;
                        F += p;
                    }
                    extra = (float)(kk - k);
                }
                dst[(int64_t)s * ldb] = (float)k + extra;
            }
        }
    }
}

int check_common(const char* who, float* out, const float* mean, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb) {
    NIC_REQUIRE(out && mean, "%s: null buffer", who);
    NIC_REQUIRE(T > 0 && T <= 65535 && S > 0 && n_scenarios > 0 && ldb >= n_scenarios, "%s: bad sizes", who);
    return 0;
}
}  // namespace

extern "C" {

int nic_sample_demand(float* out, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb, int64_t scenario_offset,
                      uint64_t seed, int32_t kind, const float* mean, const float* chol, int32_t clip, void* stream) {
    if (int e = check_common("nic_sample_demand", out, mean, T, S, n_scenarios, ldb)) return e;
    NIC_REQUIRE(kind == 0 || kind == 1, "nic_sample_demand: unknown distribution %d", kind);
    NIC_REQUIRE(kind == 1 || chol, "nic_sample_demand: normal demand needs a Cholesky factor");
    const dim3 grid(nic::ceil_div(n_scenarios, kBlock), T), block(kBlock);
    hipStream_t s = nic::as_stream(stream);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (kind == 1) {
        NIC_REQUIRE((size_t)S * kCdf * 4 <= 160 * 1024, "nic_sample_demand: too many stores for the Poisson CDF table (%d)", S);
        nic::note_kernel("sample_poisson_kernel");
        hipLaunchKernelGGL(sample_poisson_kernel, grid, block, (size_t)S * kCdf * 4, s, out, T, S, n_scenarios, (int64_t)ldb,
                           scenario_offset, k0, k1, mean);
        return nic::check_launch("nic_sample_demand");
    }
    NIC_REQUIRE(S <= 96, "nic_sample_demand: the general-covariance sampler takes up to 96 stores (%d)", S);
    const int smax = S <= 4 ? 4 : (S <= 16 ? 16 : (S <= 32 ? 32 : (S <= 64 ? 64 : 96)));
    nic::note_kernelf("sample_cholesky_kernel<%d>", smax);
#define NIC_CHOL(SM)                                                                                                          \
    hipLaunchKernelGGL(sample_cholesky_kernel<SM>, grid, block, 0, s, out, T, S, n_scenarios, (int64_t)ldb, scenario_offset, k0, \
                       k1, mean, chol, clip)
    if (smax == 4) NIC_CHOL(4);
    else if (smax == 16) NIC_CHOL(16);
    else if (smax == 32) NIC_CHOL(32);
    else if (smax == 64) NIC_CHOL(64);
    else NIC_CHOL(96);
#undef NIC_CHOL
    return nic::check_launch("nic_sample_demand");
}

int nic_sample_demand_equicorrelated(float* out, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb,
                                     int64_t scenario_offset, uint64_t seed, const float* mean, const float* std_, float rho,
                                     int32_t clip, void* stream) {
    if (int e = check_common("nic_sample_demand_equicorrelated", out, mean, T, S, n_scenarios, ldb)) return e;
    NIC_REQUIRE(std_, "nic_sample_demand_equicorrelated: null std");
    NIC_REQUIRE(rho >= 0.f && rho <= 1.f, "nic_sample_demand_equicorrelated: correlation %g outside [0, 1]", (double)rho);
    if (S == 1 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {   // (one store: rho has nothing to act on)
        const dim3 grid1(nic::ceil_div(nic::ceil_div(n_scenarios, 4), kBlock), (T + 3) / 4), block1(kBlock);
        nic::note_kernel("sample_one_store_kernel");
        hipLaunchKernelGGL(sample_one_store_kernel, grid1, block1, 0, nic::as_stream(stream), out, T, n_scenarios, (int64_t)ldb,
                           scenario_offset, (uint32_t)seed, (uint32_t)(seed >> 32), mean, std_, clip);
        return nic::check_launch("nic_sample_demand_equicorrelated");
    }
    const bool per4 = S <= 32 && (int64_t)n_scenarios * T >= (4ll << 20);
    const dim3 grid(nic::ceil_div(n_scenarios, kBlock), per4 ? (T + 3) / 4 : T), block(kBlock);
    nic::note_kernelf("sample_equicorrelated_kernel<%d>", per4 ? 4 : 1);
    if (per4)
        hipLaunchKernelGGL(sample_equicorrelated_kernel<4>, grid, block, 0, nic::as_stream(stream), out, T, S, n_scenarios,
                           (int64_t)ldb, scenario_offset, (uint32_t)seed, (uint32_t)(seed >> 32), mean, std_, sqrtf(rho),
                           sqrtf(1.f - rho), clip);
    else
        hipLaunchKernelGGL(sample_equicorrelated_kernel<1>, grid, block, 0, nic::as_stream(stream), out, T, S, n_scenarios,
                           (int64_t)ldb, scenario_offset, (uint32_t)seed, (uint32_t)(seed >> 32), mean, std_, sqrtf(rho),
                           sqrtf(1.f - rho), clip);
    return nic::check_launch("nic_sample_demand_equicorrelated");
}
}
