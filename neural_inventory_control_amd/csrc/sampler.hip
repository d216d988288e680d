// Batched synthetic demand sampler on gfx950 (replaces the host numpy generators of data_handling.py:178-211 for
// throughput runs).  Counter-based Philox4x32-10: every normal / uniform is a pure function of
// (seed, GLOBAL scenario index, period, variate index), so the traces do not depend on the launch geometry nor on how
// scenarios are sharded across GPUs.  Output is written directly in the [T][S][ldb] scenario-minor layout
// the env-step kernel reads (one coalesced 256-B store per wave and (t, s)).  HBM-write-bound: 4*S bytes per
// scenario-period.  Parity with numpy's MT19937 stream is statistical, not bitwise.
#include "nic_common.h"

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += W0;
        k1 += W1;
    }
    return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return (x + 0.5f) * 2.3283064365386963e-10f; }  // (0,1)

// four standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ void normal4(U4 r, float (&z)[4]) {
    const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
    sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

__global__ void sample_demand_kernel(float* __restrict__ out, int T, int S, int B, int64_t ldb, int64_t scenario_offset,
                                     uint32_t k0, uint32_t k1, int kind, const float* __restrict__ mean,
                                     const float* __restrict__ chol, int clip) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (b >= B) return;
    const uint64_t gb = (uint64_t)(b + scenario_offset);
    const uint32_t c0 = (uint32_t)gb, c1 = (uint32_t)(gb >> 32);
    float* dst = out + (int64_t)t * S * ldb + b;
    if (kind == 1) {  // Poisson by inversion (sequential search); one uniform per (scenario, period, store)
        for (int s = 0; s < S; ++s) {
            const U4 r = philox4x32_10(U4{c0, c1, (uint32_t)t, (uint32_t)(s >> 2)}, k0, k1);
            const uint32_t bits = (s & 3) == 0 ? r.x : (s & 3) == 1 ? r.y : (s & 3) == 2 ? r.z : r.w;
            const float u = u01(bits), lam = mean[s];
            float p = expf(-lam), F = p;
            int k = 0;
            while (u > F && k < 1000) {
                ++k;
                p *= lam / (float)k;
                F += p;
            }
            dst[(int64_t)s * ldb] = (float)k;
        }
        return;
    }
    // normal: d[s] = mean[s] + sum_{j <= s} chol[s][j] z[j]; z regenerated per 4-block (no per-lane arrays)
    for (int s = 0; s < S; ++s) {
        float acc = mean[s];
        for (int blk = 0; blk * 4 <= s; ++blk) {
            float z[4];
            normal4(philox4x32_10(U4{c0, c1, (uint32_t)t, (uint32_t)blk}, k0, k1), z);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = blk * 4 + q;
                if (j <= s) acc += chol[s * S + j] * z[q];
            }
        }
        if (clip && acc < 0.f) acc = 0.f;
        dst[(int64_t)s * ldb] = acc;
    }
}
}  // namespace

extern "C" int nic_sample_demand(float* out, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb, int64_t scenario_offset,
                                 uint64_t seed, int32_t kind, const float* mean, const float* chol, int32_t clip,
                                 void* stream) {
    NIC_REQUIRE(out && mean, "nic_sample_demand: null buffer");
    NIC_REQUIRE(kind == 1 || chol, "nic_sample_demand: normal demand needs a Cholesky factor");
    NIC_REQUIRE(kind == 0 || kind == 1, "nic_sample_demand: unknown distribution %d", kind);
    NIC_REQUIRE(T > 0 && T <= 65535 && S > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_sample_demand: bad sizes");
    dim3 grid(nic::ceil_div(n_scenarios, 256), T);
    nic::note_kernel("sample_demand_kernel");
    hipLaunchKernelGGL(sample_demand_kernel, grid, dim3(256), 0, nic::as_stream(stream), out, T, S, n_scenarios, (int64_t)ldb,
                       scenario_offset, (uint32_t)seed, (uint32_t)(seed >> 32), kind, mean, chol, clip);
    return nic::check_launch("nic_sample_demand");
}
