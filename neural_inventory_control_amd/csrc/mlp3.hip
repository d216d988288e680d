// Fused three-layer 32-wide MLP over gathered inputs (include/nic_rollout.h: nic_mlp3_fwd / nic_mlp3_bwd) and the segment
// sum that aggregates messages / scatters gradients over the static supply graph (nic_segment_sum).
//
// One wavefront = 32 scenarios of ONE entity (node or edge), so every gather index is wave-uniform: the K input rows of a
// column are K row pointers computed on the scalar unit from the segment table + the entity's map entry; lanes only add their
// scenario offset.  The layers run on the matrix cores exactly like csrc/small_rollout.hip: A = weights (lane l holds
// W[i = l & 31][kk = l >> 5]), B = activations (lane = scenario), and MFMA step s of layers 2 / 3 is defined to contract over
// k = crow(s, h) - the row the previous layer's accumulator register s holds - so activations never move between layers.
// Both halves of the wave carry the same 32 scenarios and own different feature rows; every history row is written once.
// FP32 (v_mfma_f32_32x32x2_f32: exact products, f32 accumulate); ELU as in the other kernels (small_rollout_body.h).
#include "nic_common.h"
#include "small_rollout_body.h"

namespace {
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// uniform: pointer to input row k of entity e (without the scenario offset) and its scenario stride; nullptr = zeros
struct RowRef {
    const float* p;
    int64_t scn;
};
__device__ __forceinline__ RowRef input_row(const NicMlp3Desc& d, int k, int e) {
    RowRef out{nullptr, 0};
    if (k >= d.K) return out;
    int r = k, sg = 0;
#pragma unroll
    for (int q = 0; q < NIC_MLP3_MAX_SEGS - 1; ++q)
        if (sg == q && q + 1 < d.n_segs && r >= d.seg[q].n_rows) {
            r -= d.seg[q].n_rows;
            sg = q + 1;
        }
    const NicMlp3Seg& S = d.seg[sg];
    const int ent = S.map ? S.map[e] : e;
    if (ent < 0) return out;
    out.p = S.base + (int64_t)r * S.row_stride + (int64_t)ent * S.ent_stride;
    out.scn = S.scn_stride;
    return out;
}

__device__ __forceinline__ float out_act_fwd(int act, float z) {
    if (act == NIC_MLP3_ACT_ELU) return nic::elu1(z);
    if (act == NIC_MLP3_ACT_SOFTPLUS) return z > 20.f ? z : log1pf(expf(z));  // nn.Softplus(beta=1, threshold=20)
    return z;
}
// derivative of the output activation expressed with its OUTPUT y
__device__ __forceinline__ float out_act_grad(int act, float y) {
    if (act == NIC_MLP3_ACT_ELU) return nic::elu1_grad_from_out(y);
    if (act == NIC_MLP3_ACT_SOFTPLUS) return 1.f - expf(-y);  // sigmoid(z) = 1 - exp(-softplus(z))
    return 1.f;
}

// KS = MFMA steps of the first layer (input rows 2s + h): K <= 2 * KS
template <int KS>
__global__ __launch_bounds__(64) void mlp3_fwd_kernel(NicMlp3Desc d, const float* __restrict__ weights, float* __restrict__ Y,
                                                      float* __restrict__ Xh, float* __restrict__ H1, float* __restrict__ H2) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int e = blockIdx.y;
    const int64_t b_raw = (int64_t)blockIdx.x * 32 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;
    const int64_t ent_ld = (int64_t)d.n_entities * d.ldb;   // elements between feature rows of the [rows][E][ldb] buffers
    const int64_t col = (int64_t)e * d.ldb + b;
    const int K = d.K, i = j;

    // layer 1: x rows k = 2s + h gathered straight from the sources; A fragment W1[i][2s + h]
    f32x16 acc;
    const float* W1 = weights;
    const float* b1 = W1 + 32 * K;
    const float* W2 = b1 + 32;
    const float* b2 = W2 + 32 * 32;
    const float* W3 = b2 + 32;
    const float* b3 = W3 + d.n_out * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b1[crow(r, h)];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const RowRef r0 = input_row(d, 2 * s, e), r1 = input_row(d, 2 * s + 1, e);
        const float* p = h ? r1.p : r0.p;
        const int64_t scn = h ? r1.scn : r0.scn;
        const int k = 2 * s + h;
        const float x = p ? p[b * scn] : 0.f;
        const float a = k < K ? W1[i * K + k] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x, acc, 0, 0, 0);
        if (Xh && live && k < K) Xh[(int64_t)k * ent_ld + col] = x;
    }
    float hcur[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hcur[r] = nic::elu1(acc[r]);
    if (H1 && live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) H1[(int64_t)crow(r, h) * ent_ld + col] = hcur[r];
    }
    // layer 2
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b2[crow(r, h)];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W2[i * 32 + crow(s, h)], hcur[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) hcur[r] = nic::elu1(acc[r]);
    if (H2 && live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) H2[(int64_t)crow(r, h) * ent_ld + col] = hcur[r];
    }
    // layer 3 (rows >= n_out carry zero weights)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = crow(r, h) < d.n_out ? b3[crow(r, h)] : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(i < d.n_out ? W3[i * 32 + crow(s, h)] : 0.f, hcur[s], acc, 0, 0, 0);
    if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (crow(r, h) < d.n_out) Y[(int64_t)crow(r, h) * ent_ld + col] = out_act_fwd(d.out_act, acc[r]);
    }
}

// KG = 32-row groups of the input gradient (K <= 32 * KG)
template <int KG>
__global__ __launch_bounds__(64) void mlp3_bwd_kernel(NicMlp3Desc d, const float* __restrict__ weights,
                                                      const float* __restrict__ dY, const float* __restrict__ Yo,
                                                      const float* __restrict__ H1, const float* __restrict__ H2,
                                                      float* __restrict__ dZ3, float* __restrict__ dZ2, float* __restrict__ dZ1,
                                                      float* __restrict__ dX) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int e = blockIdx.y;
    const int64_t b_raw = (int64_t)blockIdx.x * 32 + j;
    const bool live = b_raw < d.n_scenarios;
    const int64_t b = live ? b_raw : 0;
    const int64_t ent_ld = (int64_t)d.n_entities * d.ldb;
    const int64_t col = (int64_t)e * d.ldb + b;
    const int K = d.K, i = j;
    const float* W1 = weights;
    const float* W2 = W1 + 32 * K + 32;
    const float* W3 = W2 + 32 * 32 + 32;

    // dz3 = dY * act'(y) in the C layout (rows crow(r, h)); rows >= n_out are zero
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = crow(r, h);
        float v = 0.f;
        if (row < d.n_out) v = dY[(int64_t)row * ent_ld + col] * out_act_grad(d.out_act, Yo[(int64_t)row * ent_ld + col]);
        dz[r] = live ? v : 0.f;
        if (live && row < d.n_out) dZ3[(int64_t)row * ent_ld + col] = v;
    }
    // dH2 = W3^T dz3: A fragment W3[crow(s, h)][i] (contraction over the output rows)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow(s, h) < d.n_out ? W3[crow(s, h) * 32 + i] : 0.f, dz[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dz[r] = acc[r] * nic::elu1_grad_from_out(H2[(int64_t)crow(r, h) * ent_ld + col]);
        if (live) dZ2[(int64_t)crow(r, h) * ent_ld + col] = dz[r];
    }
    // dH1 = W2^T dz2
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W2[crow(s, h) * 32 + i], dz[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dz[r] = acc[r] * nic::elu1_grad_from_out(H1[(int64_t)crow(r, h) * ent_ld + col]);
        if (live) dZ1[(int64_t)crow(r, h) * ent_ld + col] = dz[r];
    }
    // dX = W1^T dz1, 32 input rows per group
    if (dX) {
#pragma unroll
        for (int g = 0; g < KG; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(32 * g + i < K ? W1[crow(s, h) * K + 32 * g + i] : 0.f, dz[s], acc, 0, 0, 0);
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * g + crow(r, h);
                    if (k < K) dX[(int64_t)k * ent_ld + col] = acc[r];
                }
            }
        }
    }
}

__global__ void segment_sum_kernel(float* __restrict__ dst, int64_t dst_rs, const float* __restrict__ src, int64_t src_rs,
                                   const int32_t* __restrict__ offsets, const int32_t* __restrict__ items,
                                   const float* __restrict__ dst_scale, int R, int B, int64_t ldb, int accumulate) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (b >= B) return;
    const int lo = offsets[n], hi = offsets[n + 1];
    const float sc = dst_scale ? dst_scale[n] : 1.f;
    for (int r = blockIdx.z; r < R; r += gridDim.z) {
        float s = 0.f;
        for (int p = lo; p < hi; ++p) s += src[(int64_t)r * src_rs + (int64_t)items[p] * ldb + b];
        float* o = dst + (int64_t)r * dst_rs + (int64_t)n * ldb + b;
        *o = accumulate ? *o + sc * s : sc * s;
    }
}

int validate(const NicMlp3Desc* d, const char* who) {
    NIC_REQUIRE(d && d->weights, "%s: null descriptor / weights", who);
    NIC_REQUIRE(d->n_entities > 0 && d->n_scenarios > 0 && d->ldb >= d->n_scenarios && d->ldb % 32 == 0,
                "%s: bad sizes (ldb must be a multiple of 32 and >= n_scenarios)", who);
    NIC_REQUIRE(d->K >= 1 && d->K <= NIC_MLP3_MAX_K, "%s: K (%d) must be 1..%d", who, d->K, NIC_MLP3_MAX_K);
    NIC_REQUIRE(d->n_out >= 1 && d->n_out <= 32, "%s: n_out (%d) must be 1..32", who, d->n_out);
    NIC_REQUIRE(d->out_act >= NIC_MLP3_ACT_NONE && d->out_act <= NIC_MLP3_ACT_SOFTPLUS, "%s: unknown output activation", who);
    NIC_REQUIRE(d->n_segs >= 1 && d->n_segs <= NIC_MLP3_MAX_SEGS, "%s: 1..%d input segments", who, NIC_MLP3_MAX_SEGS);
    int rows = 0;
    for (int s = 0; s < d->n_segs; ++s) {
        NIC_REQUIRE(d->seg[s].base && d->seg[s].n_rows > 0, "%s: empty input segment %d", who, s);
        rows += d->seg[s].n_rows;
    }
    NIC_REQUIRE(rows == d->K, "%s: segment rows (%d) do not add up to K (%d)", who, rows, d->K);
    return 0;
}
}  // namespace

extern "C" {

int nic_mlp3_fwd(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, void* stream) {
    if (int e = validate(d, "nic_mlp3_fwd")) return e;
    NIC_REQUIRE(Y, "nic_mlp3_fwd: null output");
    NIC_REQUIRE(!X_hist || (H1 && H2), "nic_mlp3_fwd: incomplete history buffers");
    const dim3 grid(nic::ceil_div(d->n_scenarios, 32), d->n_entities), block(64);
    hipStream_t s = nic::as_stream(stream);
    const int ks = (d->K + 1) / 2;
    const int t = ks <= 4 ? 4 : (ks <= 16 ? 16 : (ks <= 33 ? 33 : 48));
    nic::note_kernelf("mlp3_fwd_kernel<%d>", t);
#define NIC_MLP3_FWD(KS) hipLaunchKernelGGL(mlp3_fwd_kernel<KS>, grid, block, 0, s, *d, d->weights, Y, X_hist, H1, H2)
    if (t == 4) NIC_MLP3_FWD(4);
    else if (t == 16) NIC_MLP3_FWD(16);
    else if (t == 33) NIC_MLP3_FWD(33);
    else NIC_MLP3_FWD(48);
#undef NIC_MLP3_FWD
    return nic::check_launch("nic_mlp3_fwd");
}

int nic_mlp3_bwd(const NicMlp3Desc* d, const float* dY, const float* Y, const float* H1, const float* H2, float* dZ3, float* dZ2,
                 float* dZ1, float* dX, void* stream) {
    if (int e = validate(d, "nic_mlp3_bwd")) return e;
    NIC_REQUIRE(dY && Y && H1 && H2 && dZ3 && dZ2 && dZ1, "nic_mlp3_bwd: null buffer");
    const dim3 grid(nic::ceil_div(d->n_scenarios, 32), d->n_entities), block(64);
    hipStream_t s = nic::as_stream(stream);
    const int kg = (d->K + 31) / 32;
    nic::note_kernelf("mlp3_bwd_kernel<%d>", kg);
#define NIC_MLP3_BWD(KG) hipLaunchKernelGGL(mlp3_bwd_kernel<KG>, grid, block, 0, s, *d, d->weights, dY, Y, H1, H2, dZ3, dZ2, dZ1, dX)
    if (kg == 1) NIC_MLP3_BWD(1);
    else if (kg == 2) NIC_MLP3_BWD(2);
    else NIC_MLP3_BWD(3);
#undef NIC_MLP3_BWD
    return nic::check_launch("nic_mlp3_bwd");
}

int nic_segment_sum(float* dst, int64_t dst_row_stride, const float* src, int64_t src_row_stride, const int32_t* offsets,
                    const int32_t* items, const float* dst_scale, int32_t R, int32_t n_dst, int32_t n_scenarios, int32_t ldb,
                    int32_t accumulate, void* stream) {
    NIC_REQUIRE(dst && src && offsets && items, "nic_segment_sum: null buffer");
    NIC_REQUIRE(R > 0 && n_dst > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_segment_sum: bad sizes");
    const dim3 grid(nic::ceil_div(n_scenarios, 256), n_dst, R < 32 ? R : 32), block(256);
    nic::note_kernel("segment_sum_kernel");
    hipLaunchKernelGGL(segment_sum_kernel, grid, block, 0, nic::as_stream(stream), dst, dst_row_stride, src, src_row_stride,
                       offsets, items, dst_scale, R, n_scenarios, (int64_t)ldb, accumulate);
    return nic::check_launch("nic_segment_sum");
}
}
