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
    hipMalloc(&in, (size_t)96 * E * ld * 4); hipMalloc(&out, (size_t)192 * E * ld * 4);
    hipMemset(in, 0, (size_t)96 * E * ld * 4);
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
    return 0;
}
