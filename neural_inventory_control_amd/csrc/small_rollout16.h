// Internal: launchers of the 16-scenarios-per-wavefront whole-horizon kernels (small_rollout16.hip), called by the C ABI entry
// points in small_rollout.hip (nic_small_rollout_fwd / nic_small_rollout_bwd_wgrad) when NicSmallRolloutDesc::lane_scenarios
// selects them.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nic_rollout.h"

namespace nic {
// rows ([rows][T][ldb] floats) of the state / logit history of the 16-wide kernels: a scenario's slots are padded to whole 16-byte
// accesses (n_out = 1: one float, no padding)
__host__ __device__ inline int sr16_state_rows(int F) { return (F + 3) & ~3; }
__host__ __device__ inline int sr16_logit_rows(int n_out) { return n_out == 1 ? 1 : (n_out + 3) & ~3; }
void small_rollout16_fwd(const NicSmallRolloutDesc& d, int shape, float* rewards, float* state_final, float* states_hist,
                         float* hidden_hist, float* logits_hist, hipStream_t s);
void small_rollout16_bwd_wgrad(const NicSmallRolloutDesc& d, int shape, const float* states_hist, const float* hidden_hist,
                               const float* logits_hist, NicTable2 g_reward, float* slab, int64_t slab_stride, hipStream_t s);
}  // namespace nic
