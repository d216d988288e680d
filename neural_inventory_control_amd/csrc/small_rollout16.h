// Internal: launchers of the 16-scenarios-per-wavefront whole-horizon kernels (small_rollout16.hip), called by the C ABI entry
// points in small_rollout.hip (nic_small_rollout_fwd / nic_small_rollout_bwd_wgrad) when NicSmallRolloutDesc::lane_scenarios
// selects them.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nic_rollout.h"

namespace nic {
void small_rollout16_fwd(const NicSmallRolloutDesc& d, int shape, float* rewards, float* state_final, float* states_hist,
                         float* hidden_hist, float* logits_hist, hipStream_t s);
void small_rollout16_bwd_wgrad(const NicSmallRolloutDesc& d, int shape, const float* states_hist, const float* hidden_hist,
                               const float* logits_hist, NicTable2 g_reward, float* slab, int64_t slab_stride, hipStream_t s);
}  // namespace nic
