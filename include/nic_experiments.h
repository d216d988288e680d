/* Entry points of routes that were built, verified and measured but do NOT win on any benchmark workload.
 * They are not part of libnic_hip.so's default build: `NIC_BUILD_EXPERIMENTS=1 python -m neural_inventory_control_amd.build` adds
 * tools/experiments/wide_rollout.hip to the library (and `FusedRollout.use_wide = True` / `bench.py --wide on` then take it).
 * A/B record: profiles/r05_wide_ab.json (8,192 scenarios: 29.2 ms against 28.2 ms on the tiled launches; 65,536: 231 against 180). */
#ifndef NIC_EXPERIMENTS_H
#define NIC_EXPERIMENTS_H
#include "nic_rollout.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- whole-horizon rollout of the WIDE vanilla_warehouse policy: one launch per direction for all T periods (round 5) -------
 * Trainer.simulate_batch's loop (trainer.py:190-213) for VanillaWarehouse (neural_networks.py:358-427) with 512-wide hidden
 * layers: a workgroup carries a block of 32 scenarios through every period - hidden layers on the FP32 matrix cores with the
 * weights streamed from L2 as pre-packed MFMA fragments and the activations in LDS, logits contracted straight from the last
 * layer's accumulators, softmax head + Simulator.step (environment.py:110-299) on LDS tiles, next period's first layer from the
 * new state tile - and leaves the histories the backward sweep and the weight-gradient contractions read.
 * Histories: element (t, row, b) at base + t * period stride + row * ldb + b.  Packed weights (built by the caller once per
 * optimizer step; neural_inventory_control_amd/wide_rollout.py):
 *   Wp_hidden[l]  layer l (1 <= l < n_hidden), [H/32][H/8][64 lanes][4]: lane (r, h) of (tile, group g) holds
 *                 W_l[32 tile + r][8 g + 2 j + h], j = 0..3
 *   Wq_out        logits layer, [H/32][16][64]: lane (n, h) of (tile, r) holds W_out[n][32 tile + (r & 3) + 8 (r >> 2) + 4 h]
 *                 (0 for n >= n_out)
 * Shapes: nic_wide_rollout_ok (H == 512, 2..4 hidden layers, <= 16 stores, (S + 1) Wn = n_out <= 32, S Ws + Wn Ww <= 51, pipelines
 * <= 4 slots, ldb % 64 == 0). */
typedef struct NicWideRollout {
    NicEnvStepIO io;            /* dims + static tables (underage, holding, lead_times, wh_*); state / demand / order members unused */
    const int32_t* adjacency;   /* [Wn][S] */
    float upper_bound;
    int32_t transshipment;
    int32_t T, H, n_hidden, n_out;
    const float* demand;        /* [T][S][ld_demand], period stride ps_demand (elements) */
    int64_t ps_demand, ld_demand;
    float* states;              /* [T + 1][F (+ 1)][ldb]: block 0 = the initial state (input); blocks 1..T written (rows < F) */
    float* orders;              /* [T][S Wn + Wn][ldb] */
    float* logits;              /* [T][n_out][ldb] */
    float* rewards;             /* [T][ldb] */
    float* hidden[4];           /* [T][H][ldb] post-ELU activation of hidden layer l, or NULL (evaluation: nothing kept) */
    int64_t ps_state, ps_orders, ps_logits, ps_hidden;
    const float* Wt_in;         /* first layer transposed [F + 1][ldwt_in], row F = its bias */
    int64_t ldwt_in;
    const float* Wp_hidden[4];  /* [l] for 1 <= l < n_hidden */
    const float* b_hidden[4];
    const float* Wq_out;
    const float* b_out;         /* [n_out] or NULL */
} NicWideRollout;
int nic_wide_rollout_ok(const NicEnvDims* dims, int32_t n_out, int32_t H, int32_t n_hidden);
int nic_wide_rollout_fwd(const NicWideRollout* w, void* stream);
/* Backward sweep over the histories nic_wide_rollout_fwd left (w->hidden[l] must be set for every hidden layer): for t = T-1 .. 0
 * the first layer's input gradient of period t+1 (contracted from the registers that hold its pre-activation gradient), the env /
 * head adjoints on LDS tiles, the logits layer's input gradient and the hidden layers' input gradients with ELU' from the
 * activation history.  Writes dZ_hidden[l] ([T][H][ldb], period stride ps_dz: pre-activation gradient of hidden layer l) and
 * dZ_out ([T][n_out][ldb], ps_dzout: logits gradient) - the operands of the weight-gradient contractions
 * (nic_linear_wgrad_periods).  g_reward as for nic_env_step_bwd.  Packed weights:
 *   WpT_hidden[l]  (1 <= l < n_hidden) the TRANSPOSE of layer l packed like Wp_hidden
 *   Wq_in          first layer, [2][H/32][16][64]: lane (f, h) of (mt, tile, r) holds W_in[32 tile + (r & 3) + 8 (r >> 2) + 4 h][32 mt + f]
 *                  (0 for state rows 32 mt + f >= F)
 *   Wo_t           logits layer, [H/32][NS][64], NS = 4 / 9 / 16 >= ceil(n_out / 2): lane (k, h) of (tile, s) holds
 *                  W_out[2 s + h][32 tile + k] (0 for rows >= n_out) */
int nic_wide_rollout_bwd(const NicWideRollout* w, NicTable2 g_reward, float* const* dZ_hidden, int64_t ps_dz, float* dZ_out,
                         int64_t ps_dzout, const float* const* WpT_hidden, const float* Wq_in, const float* Wo_t, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIC_EXPERIMENTS_H */
