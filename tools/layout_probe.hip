// Memory-pattern probe for the fused gather-MLP kernels: one wavefront = 32 scenarios of one entity reads RIN rows and writes
// ROUT rows, (a) feature-major [row][entity][ldb] (what the GNN engine uses), (b) tile-major [entity][chunk][row][32].
// Build: hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/_build/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int RIN, int ROUT, bool TILE>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ in, float* __restrict__ out, int E, int ld) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int ch = blockIdx.x, e = blockIdx.y, chunks = ld / 32;
    auto addr = [&](int row, int R) -> size_t {
        return TILE ? (((size_t)e * chunks + ch) * R + row) * 32 + j : ((size_t)row * E + e) * ld + (size_t)ch * 32 + j;
    };
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < RIN / 2; ++s) acc += in[addr(2 * s + h, RIN)];
#pragma unroll
    for (int r = 0; r < ROUT / 2; ++r) out[addr(2 * r + h, ROUT)] = acc + r;
}
// (c) one wavefront = 64 scenarios of one entity, lane = scenario: every load / store instruction covers ONE row x 256 B instead of
// two rows x 128 B (what a kernel gets that owns two adjacent 32-column MFMA blocks and swaps register halves, v_permlane32_swap)
template <int RIN, int ROUT>
__global__ __launch_bounds__(64) void probe_wide(const float* __restrict__ in, float* __restrict__ out, int E, int ld) {
    const int lane = threadIdx.x, ch = blockIdx.x, e = blockIdx.y;
    auto addr = [&](int row) -> size_t { return ((size_t)row * E + e) * ld + (size_t)ch * 64 + lane; };
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < RIN; ++s) acc += in[addr(s)];
#pragma unroll
    for (int r = 0; r < ROUT; ++r) out[addr(r)] = acc + r;
}
// (d) the same two-rows-x-128-B instructions as (a), but one wavefront owns two ADJACENT 32-column chunks and alternates between them
template <int RIN, int ROUT>
__global__ __launch_bounds__(64) void probe_pair(const float* __restrict__ in, float* __restrict__ out, int E, int ld) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5, ch = blockIdx.x, e = blockIdx.y;
    auto addr = [&](int row, int c) -> size_t { return ((size_t)row * E + e) * ld + (size_t)ch * 64 + c * 32 + j; };
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < RIN / 2; ++s) acc += in[addr(2 * s + h, 0)] + in[addr(2 * s + h, 1)];
#pragma unroll
    for (int r = 0; r < ROUT / 2; ++r) {
        out[addr(2 * r + h, 0)] = acc + r;
        out[addr(2 * r + h, 1)] = acc - r;
    }
}
template <int RIN, int ROUT, int KIND>
float run2(const float* in, float* out, int E, int ld, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(ld / 64, E);
    auto go = [&]() {
        if (KIND == 0) hipLaunchKernelGGL((probe_wide<RIN, ROUT>), grid, dim3(64), 0, 0, in, out, E, ld);
        else hipLaunchKernelGGL((probe_pair<RIN, ROUT>), grid, dim3(64), 0, 0, in, out, E, ld);
    };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}
template <int RIN, int ROUT, bool TILE>
float run(const float* in, float* out, int E, int ld, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(ld / 32, E);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<RIN, ROUT, TILE>), grid, dim3(64), 0, 0, in, out, E, ld);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<RIN, ROUT, TILE>), grid, dim3(64), 0, 0, in, out, E, ld);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}
int main() {
    const int E = 34, ld = 8192;
    float *in, *out;
    hipMalloc(&in, (size_t)160 * E * ld * 4); hipMalloc(&out, (size_t)192 * E * ld * 4);
    hipMemset(in, 0, (size_t)160 * E * ld * 4);
    auto report = [&](const char* name, float ms, int rin, int rout) {
        const double bytes = (double)(rin + rout) * E * ld * 4;
        printf("%-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
    };
    report("feature-major read 96 write 192", run<96, 192, false>(in, out, E, ld, 50), 96, 192);
    report("tile-major    read 96 write 192", run<96, 192, true>(in, out, E, ld, 50), 96, 192);
    report("feature-major read 96 write 96", run<96, 96, false>(in, out, E, ld, 50), 96, 96);
    report("tile-major    read 96 write 96", run<96, 96, true>(in, out, E, ld, 50), 96, 96);
    report("feature-major read 96 write 32", run<96, 32, false>(in, out, E, ld, 50), 96, 32);
    report("tile-major    read 96 write 32", run<96, 32, true>(in, out, E, ld, 50), 96, 32);
    report("wide (256 B rows) read 96 write 192", run2<96, 192, 0>(in, out, E, ld, 50), 96, 192);
    report("pair (2 x 128 B)  read 96 write 192", run2<96, 192, 1>(in, out, E, ld, 50), 96, 192);
    report("wide (256 B rows) read 96 write 96", run2<96, 96, 0>(in, out, E, ld, 50), 96, 96);
    report("pair (2 x 128 B)  read 96 write 96", run2<96, 96, 1>(in, out, E, ld, 50), 96, 96);
    report("wide (256 B rows) read 96 write 32", run2<96, 32, 0>(in, out, E, ld, 50), 96, 32);
    report("wide (256 B rows) read 160 write 96", run2<160, 96, 0>(in, out, E, ld, 50), 160, 96);
    // the 512 x 51 first layer of the cfg3 policy: 52 rows in, 512 rows out, one "entity", 65,536 scenarios
    {
        float *in2, *out2;
        const int E2 = 1, ld2 = 65536;
        hipMalloc(&in2, (size_t)52 * ld2 * 4); hipMalloc(&out2, (size_t)512 * ld2 * 4);
        hipMemset(in2, 0, (size_t)52 * ld2 * 4);
        auto rep2 = [&](const char* name, float ms) {
            printf("%-40s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, (double)(52 + 512) * ld2 * 4 / ms / 1e9);
        };
        rep2("first layer, 128 B rows (2 per instr)", run<52, 512, false>(in2, out2, E2, ld2, 50));
        rep2("first layer, 256 B rows", run2<52, 512, 0>(in2, out2, E2, ld2, 50));
        rep2("first layer, pair (2 x 128 B)", run2<52, 512, 1>(in2, out2, E2, ld2, 50));
    }
    return 0;
}
