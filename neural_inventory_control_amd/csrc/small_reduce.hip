// nic_small_rollout_reduce: what a training step of the whole-horizon small policies does AFTER its two kernels, in two launches
// instead of four torch reductions - the sum of the per-wavefront partial gradients (slab [n_rows][P] -> grad [P]) and the sum
// of the per-period costs (rewards [T][ldb] -> total, and the total of the periods >= ignore_periods).  Stage 1: every workgroup
// sums one (256-column chunk, row group) of the slab or one contiguous chunk of the rewards into `scratch`; stage 2: one
// workgroup per 64 columns (and one for the rewards) adds the partial sums in a fixed order.  No atomics, no semaphore pass:
// the result depends on the shapes only (bit-reproducible from call to call).  Replaces trainer.py:169-173's `sum` of the costs
// and autograd's accumulation of the weight gradients for these policies (csrc/small_rollout16.hip).
#include "nic_common.h"

namespace {
constexpr int kT = 256;        // threads per workgroup = columns per chunk
constexpr int kMaxGroups = 64;  // row groups of the slab
constexpr int kRewardBlocks = 256;

struct Plan {
    int cc, rg, rows_per_group, nb;
};
Plan plan_for(int n_rows, int P, int64_t n_reward_elems) {
    Plan p;
    p.cc = P > 0 && n_rows > 0 ? (P + kT - 1) / kT : 0;
    p.rg = p.cc ? (n_rows < kMaxGroups ? n_rows : kMaxGroups) : 0;
    p.rows_per_group = p.rg ? (n_rows + p.rg - 1) / p.rg : 0;
    if (p.rg) p.rg = (n_rows + p.rows_per_group - 1) / p.rows_per_group;   // (no empty groups)
    const int64_t n4 = n_reward_elems / 4;
    p.nb = n4 > 0 ? (int)((n4 + kT - 1) / kT < kRewardBlocks ? (n4 + kT - 1) / kT : kRewardBlocks) : 0;
    return p;
}

__device__ __forceinline__ float block_sum(float v, float* lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = kT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    const float out = lds[0];
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(kT) void sr_reduce_stage1(const float* __restrict__ slab, int n_rows, int64_t stride, int P, int cc, int rg,
                                                       int rows_per_group, const float* __restrict__ rewards, int64_t n4,
                                                       int64_t ignore4, int nb, float* __restrict__ scratch) {
    __shared__ float lds[kT];
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid < cc * rg) {
        const int c = bid % cc, r = bid / cc, col = c * kT + tid;
        float acc = 0.f;
        if (col < P) {
            const int r0 = r * rows_per_group, r1 = r0 + rows_per_group < n_rows ? r0 + rows_per_group : n_rows;
            const float* p = slab + (int64_t)r0 * stride + col;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = r0;
            for (; i + 4 <= r1; i += 4) {
                a0 += p[0];
                a1 += p[stride];
                a2 += p[2 * stride];
                a3 += p[3 * stride];
                p += 4 * stride;
            }
            for (; i < r1; ++i) {
                a0 += p[0];
                p += stride;
            }
            acc = (a0 + a1) + (a2 + a3);
        }
        scratch[(int64_t)bid * kT + tid] = acc;
        return;
    }
    // rewards: chunk b of the [T * ldb] costs, four at a time (ldb is a multiple of 4, so a float4 never straddles the first
    // reported period)
    const int b = bid - cc * rg;
    const int64_t per = (n4 + nb - 1) / nb, i0 = (int64_t)b * per, i1 = i0 + per < n4 ? i0 + per : n4;
    float tot = 0.f, rep = 0.f;
    for (int64_t i = i0 + tid; i < i1; i += kT) {
        const float4 v = reinterpret_cast<const float4*>(rewards)[i];
        const float s = (v.x + v.y) + (v.z + v.w);
        tot += s;
        rep += i >= ignore4 ? s : 0.f;
    }
    const float t_all = block_sum(tot, lds), r_all = block_sum(rep, lds);
    if (tid == 0) {
        float* out = scratch + (int64_t)cc * rg * kT + 2 * b;
        out[0] = t_all;
        out[1] = r_all;
    }
}

// Stage 2: workgroup b < ceil(P / 64) owns 64 columns; its four wavefronts each add every fourth row group (independent loads in
// flight), wavefront 0 adds the four partial sums in order.  The last workgroup adds the cost partials.
__global__ __launch_bounds__(kT) void sr_reduce_stage2(const float* __restrict__ scratch, int P, int cc, int rg, int nb, int col_blocks,
                                                       float* __restrict__ grad, float* __restrict__ totals) {
    __shared__ float lds[kT];
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid < col_blocks) {
        const int lane = tid & 63, w = tid >> 6, col = bid * 64 + lane;
        const int c = col / kT, t = col % kT;
        float a0 = 0.f, a1 = 0.f;
        if (col < cc * kT) {
            int r = w;
            for (; r + 4 < rg; r += 8) {
                a0 += scratch[((int64_t)r * cc + c) * kT + t];
                a1 += scratch[((int64_t)(r + 4) * cc + c) * kT + t];
            }
            if (r < rg) a0 += scratch[((int64_t)r * cc + c) * kT + t];
        }
        lds[tid] = a0 + a1;
        __syncthreads();
        if (w == 0 && col < P) grad[col] = (lds[lane] + lds[64 + lane]) + (lds[128 + lane] + lds[192 + lane]);
        return;
    }
    const float* part = scratch + (int64_t)cc * rg * kT;
    float tot = 0.f, rep = 0.f;
    for (int b = tid; b < nb; b += kT) {
        tot += part[2 * b];
        rep += part[2 * b + 1];
    }
    const float t_all = block_sum(tot, lds), r_all = block_sum(rep, lds);
    if (tid == 0) {
        totals[0] = t_all;
        totals[1] = r_all;
    }
}
}  // namespace

extern "C" {

int nic_small_rollout_reduce_scratch(int32_t n_rows, int32_t P, int64_t n_reward_elems) {
    const Plan p = plan_for(n_rows, P, n_reward_elems);
    return p.cc * p.rg * kT + 2 * p.nb + 4;   // (<= 64 row groups x the column chunks x 256 floats)
}

int nic_small_rollout_reduce(const float* slab, int32_t n_rows, int64_t slab_stride, int32_t P, float* grad, const float* rewards,
                             int64_t n_reward_elems, int64_t ignore_elems, float* totals, float* scratch, void* stream) {
    NIC_REQUIRE(scratch, "nic_small_rollout_reduce: null scratch buffer");
    NIC_REQUIRE((slab == nullptr) == (grad == nullptr) && (rewards == nullptr) == (totals == nullptr),
                "nic_small_rollout_reduce: slab / grad and rewards / totals go together");
    NIC_REQUIRE(slab || rewards, "nic_small_rollout_reduce: nothing to reduce");
    NIC_REQUIRE(!slab || (n_rows > 0 && P > 0 && slab_stride >= P), "nic_small_rollout_reduce: bad slab shape (%d rows, %d columns, stride %lld)",
                n_rows, P, (long long)slab_stride);
    NIC_REQUIRE(!rewards || (n_reward_elems > 0 && n_reward_elems % 4 == 0 && ignore_elems >= 0 && ignore_elems % 4 == 0 &&
                             (reinterpret_cast<uintptr_t>(rewards) & 15) == 0),
                "nic_small_rollout_reduce: the costs must be 16-byte aligned and a multiple of 4 floats long (%lld, first reported %lld)",
                (long long)n_reward_elems, (long long)ignore_elems);
    const Plan p = plan_for(slab ? n_rows : 0, slab ? P : 0, rewards ? n_reward_elems : 0);
    hipStream_t s = nic::as_stream(stream);
    nic::note_kernelf("sr_reduce_stage1+2<%d,%d,%d>", p.cc, p.rg, p.nb);
    hipLaunchKernelGGL(sr_reduce_stage1, dim3(p.cc * p.rg + p.nb), dim3(kT), 0, s, slab, n_rows, slab_stride, P, p.cc, p.rg,
                       p.rows_per_group, rewards, n_reward_elems / 4, ignore_elems / 4, p.nb, scratch);
    const int col_blocks = p.cc ? (P + 63) / 64 : 0;
    hipLaunchKernelGGL(sr_reduce_stage2, dim3(col_blocks + (p.nb ? 1 : 0)), dim3(kT), 0, s, scratch, P, p.cc, p.rg, p.nb, col_blocks, grad,
                       totals);
    return nic::check_launch("nic_small_rollout_reduce");
}

}  // extern "C"
