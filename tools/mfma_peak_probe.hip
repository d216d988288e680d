// mfma_peak_probe: the CONTROL for the policy GEMMs' roofline.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_peak_probe tools/mfma_peak_probe.hip && tools/_build/mfma_peak_probe
//
// A register-only loop of v_mfma_f32_32x32x2_f32 (the instruction the 512 x 512 policy GEMMs issue): no LDS, no global memory,
// no barriers; every SIMD of the chip busy.  It reports, per variant,
//   * TFLOP/s from HIP events around the launch (what an ideal k loop could reach on THIS box in THIS power state),
//   * the shader clock two ways: (a) MFMA-issue cycles (64 per instruction, back to back) / wall time of the loop measured
//     in-kernel with the constant 100 MHz counter (s_memrealtime), (b) the s_memtime delta over the same interval.
// Variants: waves per SIMD (1, 2), independent accumulator sets per wave (1, 2, 4, 8), and a "k-loop-like" variant that adds
// the LDS fragment reads (ds_read_b128) of the production loop to the same MFMA stream - if the register-only loop runs at
// ~155 TFLOP/s and the LDS-fed one drops to the production kernel's rate, the gap is the loop's LDS/DMA pattern, not the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ACC, int LDS_FED>  // LDS_FED: 0 = register-only, 1 = reads waited for right before use, 2 = reads issued one iteration ahead
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* sink, unsigned long long* stamps, float seed) {
    extern __shared__ float lds[];
    f16v acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    if (LDS_FED) {  // something to read (contents irrelevant for timing, finite for the sink)
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = a + i * 1e-6f;
        __syncthreads();
    }
    float4 fa = make_float4(a, 0.f, 0.f, 0.f);
    float2 fb = make_float2(b, 0.f);
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (LDS_FED == 1) {
            // per 8 MFMAs the production k loop reads 6 fragments' worth (4 A + 2 B values per lane for 2 k): one b128 + one b64.
            // hipcc puts these reads behind the MFMAs and their s_waitcnt lgkmcnt(0) in front of the next iteration's: the
            // pattern it also produces for the production loop (read -> wait -> 8 MFMAs)
            a = fa.x + fa.y * 1e-9f + fa.z * 1e-9f + fa.w * 1e-9f;
            b = fb.x + fb.y * 1e-9f;
            fa = *reinterpret_cast<const float4*>(&lds[((threadIdx.x * 4 + it * 256) & 16383) & ~3]);
            fb = *reinterpret_cast<const float2*>(&lds[((threadIdx.x * 2 + it * 128 + 8192) & 16383) & ~1]);
        }
        if (LDS_FED == 2) {  // software-pipelined: the reads of iteration it + 1 are ISSUED before the MFMAs of iteration it
            fa = *reinterpret_cast<const float4*>(&lds[((threadIdx.x * 4 + it * 256) & 16383) & ~3]);
            fb = *reinterpret_cast<const float2*>(&lds[((threadIdx.x * 2 + it * 128 + 8192) & 16383) & ~1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 8 / ACC; u++)
#pragma unroll
            for (int i = 0; i < ACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (LDS_FED == 2) {
            __builtin_amdgcn_sched_barrier(0);
            a = fa.x + fa.y * 1e-9f + fa.z * 1e-9f + fa.w * 1e-9f;
            b = fb.x + fb.y * 1e-9f;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) s += acc[i][j];
    if (s == 12345.678f) sink[0] = s;  // keep the chain alive
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

template <int ACC, int LDS_FED>
static void run(const char* name, int waves_per_simd, int iters, int n_cu) {
    const int threads = 256 * waves_per_simd;       // 4 SIMDs x waves_per_simd wavefronts per workgroup, one workgroup per CU
    const size_t lds_bytes = 96 * 1024;             // > half of the 160 KB: at most one workgroup per CU
    const int n_waves = n_cu * threads / 64;
    float* sink;
    unsigned long long* stamps;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * n_waves));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop<ACC, LDS_FED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 2; warm++) hipLaunchKernelGGL((mfma_loop<ACC, LDS_FED>), dim3(n_cu), dim3(threads), lds_bytes, 0, iters, sink, stamps, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    const int reps = 5;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_loop<ACC, LDS_FED>), dim3(n_cu), dim3(threads), lds_bytes, 0, iters, sink, stamps, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    std::vector<unsigned long long> h(2 * n_waves);
    CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * n_waves, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int w = 0; w < n_waves; w++) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
    cyc /= n_waves;
    wall /= n_waves;                                  // ticks of the constant 100 MHz counter
    const double mfma_per_wave = 8.0 * iters;
    const double flop = mfma_per_wave * 4096.0 * n_waves;   // 32 x 32 x 2 x 2 flop per instruction
    const double loop_us = wall / 100.0;
    // issue-limited cycles of one SIMD: waves_per_simd wavefronts x their MFMAs x 64 cycles each
    const double issue_cycles = mfma_per_wave * 64.0 * waves_per_simd;
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"acc_sets\": %d, \"lds_fed\": %s, \"mfma_per_wave\": %.0f, "
           "\"ms_best\": %.4f, \"ms_mean\": %.4f, \"tflops_best\": %.1f, \"tflops_mean\": %.1f, "
           "\"in_kernel_loop_us\": %.1f, \"clock_ghz_from_mfma_issue\": %.3f, \"memtime_ticks_per_us\": %.1f}\n",
           name, waves_per_simd, ACC, LDS_FED == 0 ? "\"no\"" : (LDS_FED == 1 ? "\"read-wait-mfma\"" : "\"prefetched\""), mfma_per_wave, best, sum / reps, flop / best / 1e9, flop / (sum / reps) / 1e9,
           loop_us, issue_cycles / loop_us / 1e3, cyc / loop_us);
    CHECK(hipFree(sink));
    CHECK(hipFree(stamps));
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 40000;   // x 8 MFMAs x 64 cycles = 20.5 M cycles per wave at one wave per SIMD (~9 ms)
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d}\n", p.gcnArchName, n_cu, p.clockRate);
    run<4, 0>("register-only", 1, iters, n_cu);
    run<1, 0>("register-only, dependent chain", 1, iters, n_cu);
    run<8, 0>("register-only", 1, iters, n_cu);
    run<4, 0>("register-only", 2, iters / 2, n_cu);
    run<8, 0>("register-only (the production wave: 8 tiles)", 2, iters / 2, n_cu);
    run<8, 1>("k-loop-like: + LDS fragment reads, read -> wait -> MFMAs (what hipcc emits)", 2, iters / 2, n_cu);
    run<8, 2>("k-loop-like: + LDS fragment reads issued one iteration ahead", 2, iters / 2, n_cu);
    run<8, 1>("k-loop-like, one wave per SIMD: read -> wait -> MFMAs", 1, iters, n_cu);
    run<8, 2>("k-loop-like, one wave per SIMD: reads issued one iteration ahead", 1, iters, n_cu);
    // a short launch of the production kernels' length (2 x 2,048 MFMAs per wave at two waves per SIMD = ~240 us): does the clock
    // the part holds over 0.25 ms differ from the one it holds over 10 ms?
    run<8, 0>("register-only, production launch length", 2, 512, n_cu);
    // how the per-CU rate depends on how many CUs are issuing MFMAs (one workgroup per CU; `cus_active` in the variant name):
    // the production kernels run their k loop 1.8x faster per tile when only a few tiles are in flight (gemm_stamp_probe.py,
    // PROBE_SIZES=256) - is that the matrix pipe itself, i.e. is the chip-wide peak a power / clock limit rather than the pipe's?
    for (int cus : {1, 2, 8, 32, 64, 128, 192, 256}) {
        if (cus > n_cu) break;
        char name[96];
        snprintf(name, sizeof name, "register-only, cus_active=%d", cus);
        run<8, 0>(name, 2, iters / 8, cus);
        snprintf(name, sizeof name, "register-only, one wave per SIMD, cus_active=%d", cus);
        run<8, 0>(name, 1, iters / 4, cus);
    }
    return 0;
}
