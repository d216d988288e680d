// Probe (round 5): can ONE workgroup per CU carry a block of 32 scenarios through 512 x 512 layers at the FP32 MFMA rate with the
// weights streamed from L2 as pre-packed MFMA A fragments and the activations in LDS?  (The building block of a whole-horizon
// kernel for the wide vanilla_warehouse policy: every stage of a period is column-local.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o wide_layer_probe.so wide_layer_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float elu_f(float x) {
    const float xn = fminf(x, 0.f);
    const float series =
        xn * fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float viaexp = __expf(xn) - 1.f;
    const float neg = xn > -0.35f ? series : viaexp;
    return x > 0.f ? x : neg;
}
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LDS activation layout ("B layout"): element (row k, column c) at ((k / 8 * 2 + (k & 1)) * 32 + c) * 4 + ((k & 7) >> 1): a lane
// (c, h) reads the rows 8 g + 2 j + h, j = 0..3, of a k group g as ONE b128.
__device__ __forceinline__ int bl(int k, int c) { return (((k >> 3) * 2 + (k & 1)) * 32 + c) * 4 + ((k & 7) >> 1); }

// packed weights: [row tile][k group][lane][4]: lane (r, h) holds W[32 tile + r][8 g + 2 j + h], j = 0..3
// (csrc/wide_rollout.hip::stream_layer)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int64_t n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(n_floats * 4), 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}
template <int RT, int PF>
__device__ __forceinline__ void stream_layer(const float4* __restrict__ Wp, int K, int wave, const float* hin, f32x16 (&acc)[RT]) {
    // Two register sets of PF k groups each: while the MFMAs of one set run (PF x 4 RT x 64 cycles = 1.7 us at RT = 4), the
    // loads of the other set are in flight, all of them issued in FRONT of those MFMAs.  (hipcc waits for EVERYTHING outstanding
    // at the head of a loop whose loads cross the back edge - s_waitcnt vmcnt(0) - so loads issued group by group behind the
    // MFMAs that free their registers, the textbook rotation, expose the whole L2 latency once per trip: measured 0.46 of peak.)
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int ng = K / 8;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(reinterpret_cast<const float*>(Wp), (int64_t)K * K);
    const int vw = lane * 16;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int tile_off[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) tile_off[i] = (wave * RT + i) * ng * 1024;
    float4 a0[PF][RT], a1[PF][RT];
    auto fetch = [&](float4 (&a)[PF][RT], int g0) {   // groups g0 .. g0 + PF - 1 (past the end: the last groups again, unused)
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int g = g0 + d < ng ? g0 + d : ng - 1;
#pragma unroll
            for (int i = 0; i < RT; ++i) a[d][i] = buf_load4(rW, vw + tile_off[i] + g * 1024);
        }
    };
    const float4* hb = reinterpret_cast<const float4*>(hin) + h * 32 + c;
    auto run = [&](const float4 (&a)[PF][RT], int g0) {
        float4 bq[2];   // (two named sets: with one, hipcc issues each group's LDS read behind the previous group's last MFMA)
        bq[0] = hb[g0 * 64];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (d + 1 < PF) bq[(d + 1) & 1] = hb[(g0 + d + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);   // (the next group's LDS read stays in FRONT of this group's MFMAs)
            const float4 b = bq[d & 1];
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i].w, b.w, acc[i], 0, 0, 0);
        }
    };
    fetch(a0, 0);
    for (int g0 = 0; g0 < ng; g0 += 2 * PF) {   // (ng % (2 PF) == 0)
        fetch(a1, g0 + PF);
        __builtin_amdgcn_sched_barrier(0);
        run(a0, g0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(a0, g0 + 2 * PF);
        __builtin_amdgcn_sched_barrier(0);
        run(a1, g0 + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int RT, int PF>
__global__ __launch_bounds__(256) void wide_layer_probe_kernel(const float4* Wp, const float* bias, const float* X, float* Y, int H,
                                                                int64_t ldb, int n_iter, int store_hist, float* hist) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + H * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int c0 = blockIdx.x * 32;
    for (int e = tid; e < H * 32; e += 256) {
        const int k = e >> 5, cc = e & 31;
        buf0[bl(k, cc)] = X[(int64_t)k * ldb + c0 + cc];
    }
    __syncthreads();
    float* hin = buf0;
    float* hout = buf1;
    for (int it = 0; it < n_iter; ++it) {
        f32x16 acc[RT];
        stream_layer<RT, PF>(Wp, H, wave, hin, acc);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row0 = (wave * RT + i) * 32;
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bias[row0 + crow(r, h)];
            const __amdgpu_buffer_rsrc_t rG = make_rsrc(store_hist ? hist + (int64_t)it * H * ldb + c0 : hout, store_hist ? (int64_t)H * ldb : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ku = (r & 3) + 8 * (r >> 2);
                const float y = elu_f(acc[i][r] + bv[r]);
                hout[bl(row0 + ku + 4 * h, c)] = y;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rG, (4 * h * (int)ldb + c) * 4, (row0 + ku) * (int)ldb * 4, 0);
            }
        }
        __syncthreads();
        float* t = hin;
        hin = hout;
        hout = t;
    }
    for (int e = tid; e < H * 32; e += 256) {
        const int k = e >> 5, cc = e & 31;
        Y[(int64_t)k * ldb + c0 + cc] = hin[bl(k, cc)];
    }
}

extern "C" int wide_layer_probe(const void* Wp, const float* bias, const float* X, float* Y, int H, int n_cols, int64_t ldb, int n_iter,
                                int pf, int store_hist, float* hist, void* stream) {
    const dim3 grid(n_cols / 32), block(256);
    const size_t lds = (size_t)2 * H * 32 * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define L(PF_)                                                                                                              \
    do {                                                                                                                    \
        hipFuncSetAttribute(reinterpret_cast<const void*>(wide_layer_probe_kernel<4, PF_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((wide_layer_probe_kernel<4, PF_>), grid, block, lds, s, static_cast<const float4*>(Wp), bias, X, Y, H, ldb, n_iter, \
                           store_hist, hist);                                                                               \
    } while (0)
    if (H != 512) return 2;
    if (pf == 2) L(2);
    else L(4);
#undef L
    return (int)hipGetLastError();
}
