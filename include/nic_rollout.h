/*
 * nic_rollout.h — C ABI of the MI355X-native differentiable inventory-rollout engine (libnic_hip.so).
 *
 * This is the drop-in boundary.  The upstream reference (MatiasAlvo/Neural_inventory_control) is pure
 * Python/PyTorch and has no FFI of its own; the entry points below are what its hot path would bind if it
 * called native code, one per group of aten ops it issues today.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference root).  The reference-side ctypes stubs a maintainer
 * would add are shown in INTEGRATION.md; neural_inventory_control_amd/_lib.py is the in-tree binding.
 *
 * Conventions
 *  - All buffers are DEVICE pointers owned by the caller (PyTorch's caching allocator in-tree). Kernels never
 *    allocate and never synchronise; every call enqueues work on `stream` (a hipStream_t passed as void*).
 *  - float32 everywhere (the reference casts every tensor to f32, data_handling.py:81); lead times are
 *    integer-valued floats and are converted with (int) exactly like `.long()` (environment.py:422).
 *  - "Scenario-minor" (SoA) layout: a per-scenario quantity with R rows is stored [R][ldb] with the scenario
 *    index contiguous (ldb >= n_scenarios, multiple of 64 recommended) so that the 64 lanes of a wavefront read
 *    64 consecutive scenarios.  Pipelines are [location][slot][ldb]; slot 0 = on hand.
 *  - Return value: 0 on success, nonzero on error; nic_last_error() returns a thread-local message.
 */
#ifndef NIC_ROLLOUT_H
#define NIC_ROLLOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIC_ABI_VERSION 1
#define NIC_MAX_SLOTS 16 /* longest supported pipeline (max lead time) */
#define NIC_MAX_WAREHOUSES 32

/* element (loc, b) of a per-location table lives at p[loc*loc_stride + b*scn_stride]; scn_stride == 0 broadcasts
 * one row to every scenario (the reference's `expand` views, data_handling.py:257-269). */
typedef struct NicTable2 {
    const float* p;
    int64_t loc_stride;
    int64_t scn_stride;
} NicTable2;

/* element (loc, supplier, b) at p[loc*loc_stride + sup*sup_stride + b*scn_stride] */
typedef struct NicTable3 {
    const float* p;
    int64_t loc_stride;
    int64_t sup_stride;
    int64_t scn_stride;
} NicTable3;

typedef struct NicEnvDims {
    int32_t n_scenarios;      /* B */
    int32_t ldb;              /* scenario stride between rows of SoA buffers */
    int32_t n_stores;         /* S  (problem_params['n_stores']) */
    int32_t n_warehouses;     /* Wn (0 => store orders have a single outside supplier column) */
    int32_t n_echelons;       /* E  (problem_params['n_extra_echelons']) */
    int32_t store_slots;      /* Ws = initial_inventories.shape[2]  (>= 2) */
    int32_t warehouse_slots;  /* Ww */
    int32_t echelon_slots;    /* We */
    int32_t lost_demand;      /* problem_params['lost_demand'] */
    int32_t maximize_profit;  /* problem_params['maximize_profit'] */
} NicEnvDims;

/* Everything one period of dynamics reads.  Replaces Simulator.step's inputs:
 * observation[...] + action[...] + _internal_data['demands'][:, :, t]  (environment.py:110-169). */
typedef struct NicEnvStepIO {
    NicEnvDims dims;
    const float* store_inv;   /* [S][Ws][ldb]   observation['store_inventories']      */
    const float* wh_inv;      /* [Wn][Ww][ldb]  observation['warehouse_inventories']  (NULL if Wn == 0) */
    const float* ech_inv;     /* [E][We][ldb]   observation['echelon_inventories']    (NULL if E == 0)  */
    NicTable2 demand;         /* (s, b) demand of this period (environment.py:171-177) */
    NicTable3 store_orders;   /* (s, w, b) action['stores']     (B,S,max(Wn,1)) */
    NicTable2 wh_orders;      /* (w, b)    action['warehouses'] (B,Wn,1) */
    NicTable2 ech_orders;     /* (e, b)    action['echelons']   (B,E,1)  */
    NicTable2 underage;       /* (s, b) observation['underage_costs'] */
    NicTable2 holding;        /* (s, b) observation['holding_costs']  */
    NicTable3 lead_times;     /* (s, w, b) observation['lead_times'] */
    NicTable2 wh_holding;     /* (w, b) */
    NicTable2 wh_lead_times;  /* (w, b) */
    NicTable2 wh_edge_costs;  /* (w, b); p == NULL when the setting has no edge costs (environment.py:254) */
    NicTable2 ech_holding;    /* (e, b) */
    NicTable2 ech_lead_times; /* (e, b) */
} NicEnvStepIO;

/* ---- library ------------------------------------------------------------------------------------------- */
int nic_abi_version(void);
/* Identity of the SOURCES the library was built from: the first 16 hex digits of a sha256 over every HIP source (with its
 * compiler flags) and header of the build (neural_inventory_control_amd/build.py::source_id).  A binding that ships next to the
 * sources compares the two at load time and refuses a stale binary ("unknown": built outside build.py). */
const char* nic_build_id(void);
const char* nic_last_error(void);
/* Name (with template arguments) of the kernel the calling thread's most recent nic_* call launched; "" before the first.
 * Measurement aid: bench.py labels its roofline object with it instead of keeping a shape -> kernel table by hand. */
const char* nic_last_kernel(void);
/* number of visible HIP devices (does not initialise a context beyond hipGetDeviceCount) */
int nic_device_count(void);

/* ---- one period of inventory dynamics -------------------------------------------------------------------
 * Forward.  Replaces Simulator.step (environment.py:110-169) = store / warehouse / echelon cost and pipeline
 * update (environment.py:179-299) incl. update_inventory_for_heterogeneous_lead_times (environment.py:391-434).
 * Writes the next state and reward[b] = per-scenario cost of the period (the `reward` Simulator.step returns). */
int nic_env_step_fwd(const NicEnvStepIO* io, float* store_inv_out, float* wh_inv_out, float* ech_inv_out,
                     float* reward /* [ldb] */, int32_t zero_lead_upstream, void* stream);
/* zero_lead_upstream (both directions): what happens to a non-zero store order whose lead time is 0.  0: it is dropped - it has
 * no slot to arrive in.  1: the reference's behaviour (environment.py:405-432 computes the flat index lead - 1 = -1 from the start
 * of the store's pipeline and torch's put wraps it): it is added to the LAST slot of the store in front, for store 0 the last
 * store of the scenario in front (scenario n_scenarios - 1 for scenario 0) - inside the same launch (round 6; before, the callers
 * patched the state with torch ops between launches).  Scenarios are coupled through it: one process, lead times that do not vary
 * over the scenarios. */

/* Backward of the same period (what autograd derives from environment.py:179-299,405-432, including the
 * `allocation != 0` filter at :426-429 and torch's clamp / minimum tie rules).
 * g_*_out: gradient w.r.t. the NEXT state (NULL = zeros); g_reward: (b) table, scn_stride 0 = one scalar.
 * Outputs (all SoA, fully overwritten for b < n_scenarios): gradient w.r.t. the CURRENT state and the orders. */
int nic_env_step_bwd(const NicEnvStepIO* io, const float* g_store_out, const float* g_wh_out,
                     const float* g_ech_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in,
                     float* g_ech_in, float* g_store_orders /* [S][max(Wn,1)][ldb] */,
                     float* g_wh_orders /* [Wn][ldb] */, float* g_ech_orders /* [E][ldb] */,
                     int32_t zero_lead_upstream, void* stream);

/* ---- policy MLP layers (feature-major activations) ------------------------------------------------------
 * Replaces nn.Linear + activation inside MyNeuralNetwork.create_sequential_net (neural_networks.py:80-106):
 *   Y[n][b] = act( sum_k W[n][k] * X[k][b] + bias[n] )          X: [K][ldb], Y: [N][ldb], W: [N][ldw]
 * on the FP32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate).
 * act: 0 = identity, 1 = ELU(alpha=1) (nn.ELU, neural_networks.py:38). */
#define NIC_ACT_NONE 0
#define NIC_ACT_ELU 1
int nic_linear_fwd(const float* W, int64_t ldw, const float* bias /* may be NULL */, const float* X, float* Y,
                   int32_t N, int32_t K, int32_t n_scenarios, int32_t ldb, int32_t act, void* stream);

/* The same forward for a layer with a SHORT contraction and many output rows (the policy's first layer, create_sequential_net's
 * LazyLinear at neural_networks.py:88-94: 512 x 51 at BASELINE cfg3), from the TRANSPOSED weights Wt[K][ldwt]: one pass that is
 * bound by the output write instead of a tiled GEMM's epilogue.  Same contraction order and ELU as nic_linear_fwd (same bits).
 * N: a multiple of 32, >= 128; K <= 52 (nic_linear_fwd_thin_in_ok); N * ldb * 4 < 2 GiB. */
int nic_linear_fwd_thin_in_ok(int32_t N, int32_t K);
int nic_linear_fwd_thin_in(const float* Wt, int64_t ldwt, const float* bias /* may be NULL */, const float* X, float* Y,
                           int32_t N, int32_t K, int32_t n_scenarios, int32_t ldb, int32_t act, void* stream);

/* Input gradient of the same layer (autograd of F.linear + activation of the PREVIOUS layer):
 *   dX[k][b] = (sum_n Wt[k][n] * dY[n][b]) * act'(Hprev[k][b])  (+ dX[k][b] if accumulate)
 * Wt = W transposed ([K][ldwt], refreshed by the caller once per optimizer step); Hprev = the previous layer's
 * POST-activation output (ELU' is recovered from the output: x>0 ? 1 : y+1), NULL when the input is not an
 * activation (first layer). */
int nic_linear_dgrad(const float* Wt, int64_t ldwt, const float* dY, const float* Hprev, float* dX, int32_t N,
                     int32_t K, int32_t n_scenarios, int32_t ldb, int32_t act_prev, int32_t accumulate,
                     void* stream);

/* Weight/bias gradient, accumulated over scenarios into per-split slabs that persist across the periods of a
 * rollout (deterministic: no atomics):
 *   slab[split][n][k] += sum_{b in split} dY[n][b] * X[k][b]   for k < K;   slab[split][n][K] += sum_b dY[n][b]
 * slab: [n_splits][N][lds] with lds >= K+1.  nic_wgrad_num_splits gives the split count the kernel expects. */
int nic_wgrad_num_splits(int32_t N, int32_t K, int32_t n_scenarios);
int nic_linear_wgrad(const float* dY, const float* X, float* slab, int64_t lds, int32_t N, int32_t K,
                     int32_t n_scenarios, int32_t ldb, int32_t n_splits, void* stream);
/* The same contraction over n_periods operand pairs (dY + t * period_stride_dy, X + t * period_stride_x; strides in
 * elements, multiples of 4) in ONE launch: the backward sweep of a rollout keeps every period's dY resident ([T][N][ldb],
 * HBM is sized for it) and contracts weight gradients once per training step instead of once per period — one slab
 * read-modify-write instead of T, no per-period pipeline fill/drain.  Periods are accumulated last first (autograd's
 * order; partial sums are added to the slab every ~8k terms).  Shapes the LDS-DMA kernel does not take are
 * served by one nic_linear_wgrad launch per period (same result).
 * The slab's n_splits slots are used as (scenario splits x period groups): scenario chunks go down to 128 scenarios and the
 * horizon is split into groups of periods for what is still missing to give every CU a workgroup (few scenarios x many
 * periods: the reference's shipped batch size 1,024, one_warehouse_lost_demand.yml:31-34).  nic_wgrad_periods_num_splits
 * gives the slot count that fills the chip; any n_splits >= 1 is accepted (unused slots are left untouched). */
int nic_wgrad_periods_num_splits(int32_t N, int32_t K, int32_t n_scenarios, int32_t n_periods);
int nic_linear_wgrad_periods(const float* dY, const float* X, float* slab, int64_t lds, int32_t N, int32_t K,
                             int32_t n_scenarios, int32_t ldb, int32_t n_splits, int32_t n_periods,
                             int64_t period_stride_dy, int64_t period_stride_x, void* stream);
/* x[r][b] = rint(x[r][b]) (round half to even = torch.round) over rows x n_scenarios of a scenario-minor block: the
 * reference's discrete allocation `action.round()` (trainer.py:201-202) between the policy head and the env step. */
int nic_round_orders(float* x, int32_t rows, int32_t n_scenarios, int32_t ldb, void* stream);

/* Fused backward of a thin layer (N <= NIC_THIN_MAX_ROWS output rows, K % 32 == 0: the logits layer of the policy
 * MLPs): one pass over X computes what nic_linear_dgrad (dX = act'(X) * W^T dY, no accumulate) and nic_linear_wgrad
 * (slab += dY X^T per split, bias column K) compute with two.  W is the layer's weight [N][ldw] (NOT transposed);
 * n_splits as for nic_linear_wgrad (any value >= 1 is accepted; the slab must have that many splits). */
#define NIC_THIN_MAX_ROWS 32
int nic_linear_bwd_thin(const float* W, int64_t ldw, const float* dY, const float* X, float* dX, float* slab,
                        int64_t lds, int32_t N, int32_t K, int32_t n_scenarios, int32_t ldb, int32_t act_prev,
                        int32_t n_splits, void* stream);
/* dW[n][k] = scale * sum_split slab[split][n][k] (k < K), db[n] = scale * sum_split slab[split][n][K] */
int nic_wgrad_reduce(const float* slab, int64_t lds, int32_t n_splits, float* dW, int64_t lddw, float* db,
                     int32_t N, int32_t K, float scale, void* stream);

/* ---- policy heads (logits -> feasible orders) ------------------------------------------------------------
 * vanilla_warehouse (neural_networks.py:369-427 + apply_softmax_feasibility_function :140-166):
 *   logits Z: [S*Wn + Wn][ldb] (row s*Wn + w = store s from warehouse w; then Wn warehouse rows)
 *   store_orders[s][w][b] = adjacency[w][s] ? softmax_{connected s' of w, plus a constant-1 'keep' logit}(Z)[s]
 *                                             * wh_inv[w][0][b] : 0
 *   wh_orders[w][b] = sigmoid(Z[S*Wn + w][b]) * upper_bound
 * adjacency: device int32 [Wn][S] (all ones for Wn == 1, neural_networks.py:384-385). */
int nic_head_warehouse_fwd(const float* Z, const float* wh_inv /* [Wn][Ww][ldb] */, const int32_t* adjacency,
                           float upper_bound, int32_t transshipment, float* store_orders /* [S][Wn][ldb] */,
                           float* wh_orders /* [Wn][ldb] */, int32_t S, int32_t Wn, int32_t Ww,
                           int32_t n_scenarios, int32_t ldb, void* stream);
/* backward: given gradients of the orders, produce dZ and ADD the head's contribution to the gradient of the
 * warehouse on-hand slot (g_wh_inv[w][0][b] += ...). */
int nic_head_warehouse_bwd(const float* Z, const float* wh_inv, const int32_t* adjacency, float upper_bound,
                           int32_t transshipment, const float* g_store_orders, const float* g_wh_orders,
                           float* dZ, float* g_wh_inv, int32_t S, int32_t Wn, int32_t Ww, int32_t n_scenarios,
                           int32_t ldb, void* stream);

/* vanilla_warehouse head + one period of dynamics in ONE launch (round 4): what VanillaWarehouse.forward does after its MLP
 * (neural_networks.py:393-426) followed by Simulator.step (environment.py:110-299) - nic_head_warehouse_fwd then
 * nic_env_step_fwd, bit for bit, without the second launch and without re-reading the orders from HBM.  `io` as for
 * nic_env_step_fwd with n_echelons == 0, n_stores <= 64; io->store_orders / io->wh_orders must describe dense
 * [S][Wn][ldb] / [Wn][ldb] blocks: the kernel WRITES the orders there (the backward sweep reads them) and consumes them.
 * The warehouse on-hand the head allocates is io->wh_inv[w][0][b] - the state the env step then advances. */
/* The adjoint pair in one launch (nic_head_env_bwd): nic_env_step_bwd (no echelons) then nic_head_warehouse_bwd.  g_store_orders /
 * g_wh_orders ([S][Wn][ldb] / [Wn][ldb]) are scratch outputs of the first half that the second half consumes; dZ receives the
 * logits' gradient, g_wh_in the warehouse state gradient INCLUDING the head's contribution to the on-hand slot.
 * COMPACT logits: on a sparse many-warehouse graph the logits of (store, warehouse) pairs without an edge are never read upstream
 * (`store_intermediate_outputs[:, connected_stores, w_idx]`, neural_networks.py:403-417), so the logits layer may compute the
 * connected pairs only.  logit_rows [Wn][S] (int32, device): row of Z / dZ of pair (w, s) - any valid row for a pair without an
 * edge (loaded, never used; no gradient row is written for it); first_wh_row: row of warehouse 0's own order logit (:422),
 * warehouse w's is first_wh_row + w.  logit_rows == NULL with first_wh_row < 0: dense logits, rows in (store, warehouse) order.
 * (Round 6: one pair of entry points - the former dense-only wrappers and their `_rows` twins are these.) */
int nic_head_env_fwd(const NicEnvStepIO* io, const float* Z, const int32_t* adjacency, const int32_t* logit_rows,
                     int32_t first_wh_row, float upper_bound, int32_t transshipment, float* store_inv_out, float* wh_inv_out,
                     float* reward, void* stream);
int nic_head_env_bwd(const NicEnvStepIO* io, const float* Z, const int32_t* adjacency, const int32_t* logit_rows,
                     int32_t first_wh_row, float upper_bound, int32_t transshipment, const float* g_store_out,
                     const float* g_wh_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_store_orders,
                     float* g_wh_orders, float* dZ, void* stream);

/* ---- the whole per-period "tail" of the vanilla_warehouse rollout in one launch per direction (round 5) -------------------
 * Between two periods' hidden-layer GEMMs the rollout of trainer.py:190-213 runs, forward, the logits layer
 * (neural_networks.py:88-106, last nn.Linear), the softmax head (:393-426, :140-166), Simulator.step (environment.py:110-299)
 * and the NEXT period's first layer + ELU (its input is the state the step just produced); backward the adjoints in reverse.
 * Every stage is column-local, so one workgroup carries a block of 32 scenarios through all of them on LDS tiles:
 *   nic_period_tail_fwd = nic_linear_fwd (logits) + nic_head_env_fwd + nic_linear_fwd_thin_in (ELU, bias as row F of Wt_in)
 *   nic_period_tail_bwd = nic_linear_dgrad (first layer of period t+1, accumulate) + nic_head_env_bwd + nic_linear_bwd_thin
 * with the same arithmetic and summation orders as those launches take at small batches (bit for bit: rewards, orders, states,
 * logits, hidden activations, state and activation gradients; the logits layer's WEIGHT gradient is summed per workgroup in a
 * different order: 1e-6).  Shapes: nic_period_tail_ok (<= 16 stores, (S + 1) Wn = n_out <= 32 logits, S Ws + Wn Ww + 1 <= 52
 * state rows, pipelines <= 8 slots, K <= 512, ldb % 64 == 0; weight rows padded to 16 bytes, padding finite). */
typedef struct NicPeriodTail {
    NicEnvStepIO io;           /* period t.  store_inv / wh_inv: ONE [S Ws + Wn Ww][ldb] block (the MLP's input rows);
                                * store_orders / wh_orders: ONE dense [S Wn + Wn][ldb] block (forward: written; backward: read);
                                * demand: a [S][ld] block (scn_stride 1) */
    const int32_t* adjacency;  /* [Wn][S] */
    float upper_bound;
    int32_t transshipment;
    const float* W_out;        /* logits layer [n_out][ldw_out] (rows padded to 16 bytes) */
    int64_t ldw_out;
    const float* b_out;        /* [n_out] or NULL */
    int32_t n_out, K;          /* K = width of the last hidden layer */
    const float* Wt_in;        /* first layer TRANSPOSED [F + 1][ldwt_in], row F = its bias (F = S Ws + Wn Ww) */
    int64_t ldwt_in;
    int32_t N1;                /* width of the first hidden layer */
} NicPeriodTail;
int nic_period_tail_ok(const NicEnvDims* dims, int32_t n_out, int32_t K, int32_t N1);
/* H_last [K][ldb]: last hidden activation of period t.  Writes Z [n_out][ldb] (logits), the orders (io), state_out [F][ldb]
 * (state of period t+1; row F of that block - the ones row - is the caller's), reward [ldb], and H_first_next [N1][ldb] =
 * ELU(first layer) of period t+1 (NULL: last period, stage skipped). */
int nic_period_tail_fwd(const NicPeriodTail* t, const float* H_last, float* Z, float* state_out, float* reward,
                        float* H_first_next, void* stream);
/* dZ_first_next [N1][ldb]: pre-activation gradient of period t+1's first layer; g_state_next [F][ldb]: what this call left in
 * g_state_out for period t+1 (both NULL for the last period: the state after it carries no gradient).  Writes g_state_out
 * [F][ldb] (the env + head part of d loss / d state(t); the first layer's part is added by the NEXT call), dH_last [K][ldb]
 * (ELU' applied) and adds the logits layer's weight / bias gradient to slab slot = workgroup ([n_slots][n_out][lds], lds >= K + 1,
 * n_slots = nic_period_tail_bwd_slots(n_scenarios); first != 0: the slot is overwritten - no memset per sweep). */
int nic_period_tail_bwd_slots(int32_t n_scenarios);
int nic_period_tail_bwd(const NicPeriodTail* t, const float* Z, const float* H_last, const float* dZ_first_next,
                        const float* g_state_next, NicTable2 g_reward, float* g_state_out, float* dH_last, float* slab,
                        int64_t lds, int32_t n_slots, int32_t first, void* stream);

/* vanilla_one_store (neural_networks.py:200-214): orders[s][b] = softplus(Z[s][b] + 1)  (threshold 20 like
 * nn.Softplus).  rows = number of output rows (1 for the shipped config). */
int nic_head_softplus_fwd(const float* Z, float* orders, int32_t rows, int32_t n_scenarios, int32_t ldb,
                          void* stream);
int nic_head_softplus_bwd(const float* Z, const float* g_orders, float* dZ, int32_t rows, int32_t n_scenarios,
                          int32_t ldb, void* stream);

/* data_driven (DataDrivenNet.forward, neural_networks.py:474-515): Z [Wn + S*Wn][ldb] = the last layer's output BEFORE its ReLU,
 * rows [Wn warehouse orders | S x Wn store orders, store-major]; mask [S][Wn] (device floats, 1 = edge) = the setting's
 * adjacency transposed.  wh_orders[w] = relu(Z[w]); store_orders[s][w] = relu(Z[Wn + s*Wn + w]) * mask[s][w] *
 * min(1, sum_k wh_inv[w][k] / (sum_s (...) + 1e-10)) - `apply_proportional_allocation` (:111-138) sums the pipeline it is handed.
 * Wn == 0 (one-store settings): store_orders[s] = relu(Z[s]); wh_inv / mask / wh_orders may be NULL.
 * Backward: dZ fully written, g_wh_inv[w][k][b] += d(scale)/... for every slot k (the clip passes where the ratio <= 1). */
int nic_head_data_driven_fwd(const float* Z, const float* wh_inv, const float* mask, float* store_orders, float* wh_orders,
                             int32_t S, int32_t Wn, int32_t Ww, int32_t n_scenarios, int32_t ldb, void* stream);
int nic_head_data_driven_bwd(const float* Z, const float* wh_inv, const float* mask, const float* g_store_orders,
                             const float* g_wh_orders, float* dZ, float* g_wh_inv, int32_t S, int32_t Wn, int32_t Ww,
                             int32_t n_scenarios, int32_t ldb, void* stream);

/* vanilla_serial (neural_networks.py:319-355): Z rows = [E echelons..., warehouse, store]; each row is
 * sigmoid(Z) * upstream on-hand, where upstream = [upper_bound, ech_inv[0..E-1][0], wh_inv[0][0]].
 * NOTE: the reference detaches the MLP input (torch.tensor(...) at :329); that is the caller's concern. */
int nic_head_serial_fwd(const float* Z, const float* wh_inv, const float* ech_inv, float upper_bound,
                        float* store_orders, float* wh_orders, float* ech_orders, int32_t E, int32_t Ww,
                        int32_t We, int32_t n_scenarios, int32_t ldb, void* stream);
int nic_head_serial_bwd(const float* Z, const float* wh_inv, const float* ech_inv, float upper_bound,
                        const float* g_store_orders, const float* g_wh_orders, const float* g_ech_orders,
                        float* dZ, float* g_wh_inv, float* g_ech_inv, int32_t E, int32_t Ww, int32_t We,
                        int32_t n_scenarios, int32_t ldb, void* stream);

/* ---- whole-horizon rollout of the small policies --------------------------------------------------------------
 * One kernel runs ALL T periods of Trainer.simulate_batch (trainer.py:190-213) for the one-store chain settings
 * (one_store_lost / one_store_backlogged / serial_system: S = 1, Wn <= 1, E <= 3) with a 32-wide MLP policy:
 * vanilla_one_store (neural_networks.py:195-214, head 0) or vanilla_serial (:314-355, head 1).  One lane = one scenario;
 * pipeline slots and activations stay in registers across the horizon.  The state block is [F][ldb] in the reference's
 * cat(flatten(...)) order: store slots, then warehouse slots, then echelon slots. */
#define NIC_SR_MAX_INPUTS 16  /* F  = Ws + Wn*Ww + E*We */
#define NIC_SR_HIDDEN 32      /* width of every hidden layer */
#define NIC_SR_MAX_OUTPUTS 8
#define NIC_SR16_STATE_ROWS(F) (((F) + 3) & ~3)
#define NIC_SR16_LOGIT_ROWS(n_out) ((n_out) == 1 ? 1 : (((n_out) + 3) & ~3))
typedef struct NicSmallRolloutDesc {
    int32_t n_scenarios, ldb, T, t0;       /* t0 = observation_params['demand']['period_shift'] */
    int32_t F, n_hidden, n_out, head;      /* n_hidden in 1..3; head 0: softplus(z+1), 1: sigmoid(z) * upstream on-hand */
    int32_t Ws, Wn, Ww, E, We;
    int32_t lost_demand, maximize_profit;
    int32_t detach_input;                  /* 1: no gradient through the MLP input (vanilla_serial, :329) */
    int32_t round_orders;                  /* 1: orders rounded half-to-even before the env step (discrete allocation,
                                              trainer.py:201-202); forward / evaluation only */
    float upper_bound;                     /* model.warehouse_upper_bound (head 1) */
    int32_t lane_scenarios;                /* scenarios per wavefront of the matrix-core kernels: 0 / 32 (v_mfma_f32_32x32x2_f32)
                                              or 16 (v_mfma_f32_16x16x4_f32: twice the wavefronts, half the chain per wavefront;
                                              for batches that leave SIMDs idle at 32).  With 16 the histories are in a
                                              wave-native order private to nic_small_rollout_fwd / nic_small_rollout_bwd_wgrad
                                              (pass the same value to both), and states_hist / logits_hist hold
                                              NIC_SR16_STATE_ROWS(F) / NIC_SR16_LOGIT_ROWS(n_out) rows of [T][ldb] floats */
    const float* weights;                  /* packed: [W1 32xF][b1 32] [W_l 32x32][b_l 32]... [Wout n_out x 32][bout n_out] */
    const float* demand;                   /* [T_total][ldb] */
    const float* state0;                   /* [F][ldb] */
    NicTable2 underage, holding, lead;     /* store tables (loc index 0) */
    NicTable2 wh_holding, wh_lead, wh_edge;
    NicTable2 ech_holding, ech_lead;       /* (e, b) */
} NicSmallRolloutDesc;

/* Forward: rewards [T][ldb] (per-period per-scenario cost), state_final [F][ldb].  When states_hist != NULL the
 * activations the backward needs are stored, every row contiguous over (t, b): states_hist [F][T][ldb],
 * hidden_hist [32*n_hidden][T][ldb] (post-ELU), logits_hist [n_out][T][ldb]. */
int nic_small_rollout_fwd(const NicSmallRolloutDesc* d, float* rewards, float* state_final, float* states_hist,
                          float* hidden_hist, float* logits_hist, void* stream);
/* Backward sweep over the stored activations: g_reward (b) = d loss / d reward[b, t] (same for every t).  Emits the
 * pre-activation gradients dz_hidden [32*n_hidden][T][ldb] and dz_out [n_out][T][ldb]; the weight gradients are then
 * nic_linear_wgrad contractions over n_scenarios = T*ldb with ldb = T*ldb (layer l: dY = its dz rows, X = its input rows). */
int nic_small_rollout_bwd(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                          const float* logits_hist, NicTable2 g_reward, float* dz_hidden, float* dz_out, void* stream);
/* The same backward sweep with the weight gradients contracted in the kernel (no dz history, no GEMM launches): every
 * wavefront (32 or 16 scenarios, see lane_scenarios) keeps dW / db of all layers in registers over the whole horizon and stores its partial gradient, in
 * the layout of the packed weights, to slab[wavefront * slab_stride ...]; the caller sums the
 * nic_small_rollout_bwd_wgrad_slots(n_scenarios) rows.  Replaces trainer.py:173 (`backward`) for the small policies. */
int nic_small_rollout_bwd_wgrad_slots(int32_t n_scenarios);
int nic_small_rollout_bwd_wgrad(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                                const float* logits_hist, NicTable2 g_reward, float* slab, int64_t slab_stride, void* stream);

/* What follows the two launches in a training step of these policies, as two small launches: grad [P] = sum over the n_rows
 * per-wavefront partial gradients of slab [n_rows][slab_stride >= P] (nic_small_rollout_bwd_wgrad), and totals[0] = sum of the
 * n_reward_elems = T * ldb per-period costs, totals[1] = sum of those from element ignore_elems = ignore_periods * ldb on
 * (trainer.py:207-210: total and reported reward; padding columns hold zeros).  Fixed summation order, no atomics.  Either pair
 * (slab, grad) / (rewards, totals) may be NULL.  scratch: nic_small_rollout_reduce_scratch(...) floats. */
int nic_small_rollout_reduce_scratch(int32_t n_rows, int32_t P, int64_t n_reward_elems);
int nic_small_rollout_reduce(const float* slab, int32_t n_rows, int64_t slab_stride, int32_t P, float* grad, const float* rewards,
                             int64_t n_reward_elems, int64_t ignore_elems, float* totals, float* scratch, void* stream);

/* ---- whole-horizon rollout of the closed-form policies ---------------------------------------------------------
 * base_stock (neural_networks.py:216-229), capped_base_stock (:296-311) and echelon_stock (:231-294) for T periods of
 * Trainer.simulate_batch (trainer.py:190-213) in ONE launch, forward AND gradient: one lane per store chain, pipelines in
 * registers, the demand trace is the only HBM stream.  The policies' few scalar parameters enter as `levels` (device
 * floats, the output of the reference's tiny `net` after its activation: base_stock [level]; capped [level, cap];
 * echelon_stock [level_0 .. level_{E+1}] = cumsum(softplus(net + 10)).flip(0), upstream first) and the kernel returns
 * d(total cost)/d(level_j) by forward-mode differentiation with torch's tie rules (csrc/closed_form_body.h); the chain
 * through `net` stays in torch autograd.
 * Settings: base / capped: Wn = 0, E = 0, any S (stores are independent chains: grid.y = store);
 *           echelon_stock: S = 1, Wn = 1, 1 <= E <= 3 (serial_system).  Every pipeline has 2..NIC_MAX_SLOTS slots. */
#define NIC_CF_MAX_LEVELS 5
#define NIC_CF_BASE_STOCK 0
#define NIC_CF_CAPPED 1
#define NIC_CF_ECHELON 2
typedef struct NicClosedFormDesc {
    int32_t n_scenarios, ldb, T, t0;       /* t0 = observation_params['demand']['period_shift'] */
    int32_t ignore_periods;                /* totals row 1 sums periods >= ignore_periods (trainer.py:209-210) */
    int32_t policy, n_levels;              /* NIC_CF_*; 1, 2 or E + 2 */
    int32_t S, Ws, Wn, Ww, E, We;
    int32_t lost_demand, maximize_profit;
    int32_t round_orders;                  /* discrete allocation (trainer.py:201-202); evaluation only */
    const float* levels;                   /* device [n_levels] */
    const float* demand;                   /* [T_total][S][ldb] */
    const float* state0;                   /* [S][F][ldb], F = Ws + Wn*Ww + E*We, slots in the reference's cat order */
    NicTable2 underage, holding, lead;     /* (s, b) */
    NicTable2 wh_holding, wh_lead, wh_edge;
    NicTable2 ech_holding, ech_lead;       /* (e, b) */
} NicClosedFormDesc;
/* Number of rows of the g_levels_partial buffer nic_closed_form_rollout fills (one per workgroup). */
int nic_closed_form_num_partials(int32_t n_scenarios, int32_t S);
/* Outputs, each may be NULL: reward_hist [T][S][ldb] (per-period cost per chain), totals [2][S][ldb] (sum over all periods;
 * over periods >= ignore_periods), state_final [S][F][ldb], g_levels_partial [num_partials][n_levels]: per-workgroup sums
 * over its chains of d(totals[0])/d(level_j) — the caller adds the rows (deterministic, no atomics).  Padding columns
 * (b >= n_scenarios) of the outputs are left untouched. */
/* (nic_closed_form_rollout_sums with with_grad = (partial != NULL), with_sums = 0, partial_stride = n_levels is the launch just
 * described; round 6 dropped the separate entry point.)
 * The launch with the per-chain sums added over each wavefront of chains as well: partial [num_partials][partial_stride],
 * row = [d(total)/d(level_j) for j < n_levels if with_grad][total, reported if with_sums] - the caller adds the rows and needs
 * neither `totals` (8 B per chain written, then reduced by a second kernel over all chains) nor a separate gradient buffer.
 * (Round 5: at 10^6 chains the reduction of `totals` took longer than the rollout.) */
int nic_closed_form_rollout_sums(const NicClosedFormDesc* d, float* reward_hist, float* totals, float* state_final, float* partial,
                                 int32_t partial_stride, int32_t with_grad, int32_t with_sums, void* stream);

/* ---- whole-horizon rollout of the data_driven policy for small batches (csrc/horizon_rollout.hip) --------------------
 * The reference trains DataDrivenNet (neural_networks.py:430-515, data_driven_net.yml: two 64-wide hidden layers) on batches of
 * 72 products x 21 stores x 3 warehouses x 95 weeks (many_warehouses_real_data_lost_demand.yml:44-47): per period that is a
 * handful of workgroups, and Trainer.simulate_batch's loop (trainer.py:190-213) becomes ~1,050 dependent launches per training
 * step.  These two entry points run ALL periods in one launch each: a workgroup owns 16 scenarios for the whole horizon, weights
 * stay in registers as MFMA fragments, state / logits / orders / static tables in LDS, head and env step are the per-period
 * kernels' own bodies.  The first layer is split: its contraction with the OBSERVATION rows of the input (past demands, costs,
 * days from christmas, lead times - independent of the rollout) is done by the caller for all periods with one nic_linear_fwd
 * over T * ldb columns (z1_obs, bias included); the kernel contracts the state rows per period.
 * Every history buffer holds element (row, t, b) at row * hist_stride + t * ldb + b, so that the weight gradients are plain
 * nic_linear_wgrad contractions over n_scenarios = T * ldb columns (padding columns of the dz histories are never written:
 * zero them once).  Supported: no extra echelons, <= 64 stores, pipelines of 2..8 slots, <= 256 state rows, hidden widths <= 64,
 * n_out = Wn + S*Wn (S when Wn == 0) <= 128 (nic_horizon_rollout_ok). */
typedef struct NicHorizonDesc {
    NicEnvStepIO io;            /* dims + static tables; store_inv / wh_inv / ech_inv / demand / *_orders members are ignored */
    int32_t T, t0;              /* periods; observation_params['demand']['period_shift'] */
    int32_t H1, H2, n_out;      /* hidden widths; logits rows [Wn warehouse orders | S x Wn store orders] */
    int32_t round_orders;       /* discrete allocation (trainer.py:201-202); forward / evaluation only */
    const float* W1;            /* [H1][ldw1]: first-layer weights; columns 0 .. S*Ws + Wn*Ww - 1 multiply the state rows */
    int64_t ldw1;
    const float* W2;            /* [H2][ldw2] */
    int64_t ldw2;
    const float* W3;            /* [n_out][ldw3] */
    int64_t ldw3;
    const float* b2;
    const float* b3;
    const float* mask;          /* [S][Wn] adjacency (1 = edge) as nic_head_data_driven_fwd takes it; NULL when Wn == 0 */
    const float* demand;        /* [>= t0 + T][S][ldb] */
    int64_t hist_stride;        /* elements between consecutive rows of z1_obs, the tape and every history (>= T * ldb) */
    /* Policies whose decisions do not need the MLP inside the loop ride on the same kernels (same env step, same histories):
     *   head_mode 1: `tape` [S*max(Wn,1) + Wn][T][ldb] holds every period's ORDERS (store orders, then warehouse orders) - policies
     *     that do not read the state, e.g. JustInTime (neural_networks.py:634-739); forward / evaluation only.
     *   head_mode 2: `tape` [S][T][ldb] holds order-up-to LEVELS: order = clip(level - sum of the store's pipeline, min = 0)
     *     (no clip with allow_negative) - QuantilePolicy.forecast_base_stock_allocation (neural_networks.py:560-575), whose levels
     *     depend on the demand trace and the policy's parameters but not on the state; one-supplier settings (Wn == 0).  The
     *     backward then writes d loss / d level into dz3_hist [S][T][ldb] (dz1_hist / dz2_hist and the h / logits histories
     *     are not touched and may be NULL).
     * W1..b3, H1, H2 are ignored in both; n_out stays the setting's order count. */
    int32_t head_mode;          /* 0: the MLP + data_driven head */
    int32_t allow_negative;     /* head_mode 2: no clip (ReturnsNV, neural_networks.py:613-622) */
    const float* tape;
} NicHorizonDesc;
/* 1 if the two kernels take this shape (sizes above, LDS budget), else 0 (nic_last_error says why). */
int nic_horizon_rollout_ok(const NicHorizonDesc* d);
/* Forward: z1_obs [H1][T][ldb] = first-layer pre-activations from the observation rows + bias; state0 [F_dyn][ldb] (store
 * pipelines, then warehouse pipelines: the reference's cat order).  rewards [T][ldb], state_final [F_dyn][ldb] (may be NULL).
 * With state_hist != NULL the histories the backward needs are written: state_hist [F_dyn][T][ldb] (state BEFORE period t),
 * h1_hist / h2_hist (post-ELU), logits_hist [n_out][T][ldb], orders_hist [S*max(Wn,1) + Wn + Wn][T][ldb] (the orders, then
 * what every warehouse shipped: the backward does not re-sum the orders). */
int nic_horizon_rollout_fwd(const NicHorizonDesc* d, const float* z1_obs, const float* state0, float* rewards, float* state_final,
                            float* state_hist, float* h1_hist, float* h2_hist, float* logits_hist, float* orders_hist,
                            void* stream);
/* Backward over the stored histories; g_reward (b) = d loss / d reward[b, t] (the same for every t).  Writes the pre-activation
 * gradients dz1_hist [H1][T][ldb], dz2_hist [H2][T][ldb], dz3_hist [n_out][T][ldb] (live columns only). */
int nic_horizon_rollout_bwd(const NicHorizonDesc* d, const float* state_hist, const float* h1_hist, const float* h2_hist,
                            const float* logits_hist, const float* orders_hist, NicTable2 g_reward, float* dz1_hist,
                            float* dz2_hist, float* dz3_hist, void* stream);

/* ---- fused three-layer 32-wide MLP over gathered inputs (graph policies) -----------------------------------------
 * The GNN policy (neural_networks.py:742-1492) applies five small MLPs (`gnn.yml`: K -> 32 -> 32 -> 32 or 1, ELU inside) to
 * every node / edge of the supply graph of every scenario, on inputs that are concatenations of gathered node / edge
 * features (:1105-1192, :1229-1269).  One launch evaluates one such MLP for all (entity, scenario) columns: a wavefront owns
 * 32 scenarios of one entity, GATHERS its K input rows straight from the source buffers through per-segment entity maps
 * (the static graph: no materialised concatenation), runs the three layers on the matrix cores with the activations chained
 * in registers (csrc/mlp3.hip), and writes the output plus what the backward needs.
 *   column (e, b), input row of segment s, row r:  seg[s].base[r*row_stride + ent*ent_stride + b*scn_stride],
 *   ent = seg[s].map ? seg[s].map[e] : e;  ent < 0 reads 0 (the virtual supplier / customer node).
 * Buffers are [rows][n_entities][ldb] (scenario-minor); weights packed [W1 32xK][b1 32][W2 32x32][b2 32][W3 n_out x 32][b3]. */
#define NIC_MLP3_MAX_K 96
#define NIC_MLP3_MAX_SEGS 4
#define NIC_MLP3_ACT_NONE 0
#define NIC_MLP3_ACT_ELU 1
#define NIC_MLP3_ACT_SOFTPLUS 2
typedef struct NicMlp3Seg {
    const float* base;
    const int32_t* map;      /* [n_entities] or NULL (identity) */
    int64_t row_stride;      /* elements between consecutive feature rows of the source buffer */
    int64_t ent_stride;      /* elements between entities (0 broadcasts one entity to all) */
    int64_t scn_stride;      /* 1, or 0 for a per-entity constant (e.g. an edge's lead time) */
    int32_t n_rows;
    int32_t reserved;
} NicMlp3Seg;
typedef struct NicMlp3Desc {
    int32_t n_entities, n_scenarios, ldb;
    int32_t K, n_out, out_act, n_segs;   /* K = sum of segment rows <= NIC_MLP3_MAX_K; n_out = 32 or 1..8 */
    int32_t hist_native;   /* 1: H1 / H2 are private to nic_mlp3_fwd / nic_mlp3_bwd_hist and kept as one contiguous [32 rows][32
                              scenarios] block per (entity, 32-scenario chunk): block (e * ldb / 32 + chunk) of the buffer, 4 KB
                              each - every history access of a wavefront is one DRAM page.  0: rows hist_row_stride apart */
    NicMlp3Seg seg[NIC_MLP3_MAX_SEGS];
    const float* weights;
    const float* weights_t;    /* the same weights transposed, read by the FORWARD kernel (lane i of contraction step k reads
                                  consecutive words): [W1^T K x 32][b1 32][W2^T 32 x 32][b2 32][W3^T 32 x 32, columns >= n_out
                                  zero][b3 padded to 32].  `weights` (natural layout) is what the backward kernel reads. */
    int64_t hist_row_stride;   /* elements between feature rows of X_hist / H1 / H2 / dZ*; 0 = n_entities * ldb.  A larger stride
                                  interleaves several periods in one row so that a weight gradient contracts over all of them */
    int64_t ent_row_stride;    /* elements between feature rows of Y / Ysum / residual / dY / dX; 0 = n_entities * ldb.  A larger
                                  stride evaluates the MLP for the FIRST n_entities entities of buffers that hold more (the GNN's
                                  edge update / output MLPs skip the demand edges, whose results nothing reads) */
} NicMlp3Desc;
/* Y [n_out][n_entities][ldb] = out_act(MLP(x)).  When X_hist != NULL the gathered inputs and the hidden activations are kept
 * for the backward / weight gradients: X_hist [K][n_entities][ldb], H1, H2 [32][n_entities][ldb] (post-ELU), rows
 * hist_row_stride apart. */
int nic_mlp3_fwd(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, void* stream);
/* ... with a residual connection folded in: additionally Ysum [n_out][n_entities][ldb] = residual + Y (the GNN's
 * nodes1 = nodes0 + node_update(...), edges1 = edges0 + edge_update(...); neural_networks.py:1311-1340). */
int nic_mlp3_fwd_residual(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, const float* residual, float* Ysum,
                          void* stream);
/* (Round 6: the round-2 backward that wrote the three pre-activation gradients of every column for later nic_linear_wgrad
 * contractions - `nic_mlp3_bwd` - is gone: it lost to the in-kernel weight gradients below by 10 % in round 2 and was kept only as a
 * test mode.)
 * Backward of the same MLP WITHOUT any stored history: the kernel gathers the inputs again, recomputes the two hidden layers, and
 * contracts the weight gradients itself — each wavefront keeps dW1 / dW2 / dW3 (+ bias columns) in registers across all the
 * (entity, 32-scenario) chunks it walks (operands transposed through LDS into the MFMA's row-owner layout) and adds them to
 * ITS slot of the slabs at the end:  slab_l[slot][n][k] += ..., column K_l = bias gradient, lds_l = slab row length.
 * nic_mlp3_bwd_fused_slots() slots per launch; slabs persist across the periods of a rollout and are reduced once with
 * nic_wgrad_reduce.  Needs d->weights AND d->weights_t.  HBM traffic per column: the K gathered input rows, dY / Y, dX. */
int nic_mlp3_bwd_fused_slots(void);
int nic_mlp3_bwd_fused(const NicMlp3Desc* d, const float* dY, const float* Y, float* dX, float* slab1, int64_t lds1,
                       float* slab2, int64_t lds2, float* slab3, int64_t lds3, void* stream);
/* The backward over the STORED activations with the weight gradients contracted in the kernel (no dZ history, no separate
 * contraction pass): reads dY / Y, X_hist / H1 / H2 (rows hist_row_stride apart, 16-byte aligned, K * hist_row_stride * 4 < 2^31),
 * writes dX, and adds dW1 / dW2 / dW3 (+ bias columns) to slab slot = workgroup; nic_mlp3_bwd_hist_slots() slots per launch,
 * reduced once with nic_wgrad_reduce.  X_hist may be NULL: the inputs are then read again from the segments' source buffers
 * (which must still hold what the forward read, be 16-byte aligned per row and zero in their padding columns), and the forward
 * need not keep an X history at all.  HBM traffic per column: K input rows + 64 history rows, dY / Y, dX. */
int nic_mlp3_bwd_hist_slots(void);
int nic_mlp3_bwd_hist(const NicMlp3Desc* d, const float* dY, const float* Y, const float* X_hist, const float* H1,
                      const float* H2, float* dX, float* slab1, int64_t lds1, float* slab2, int64_t lds2, float* slab3,
                      int64_t lds3, void* stream);
/* Several segment sums / plain addends into one destination in one launch:
 * dst = (accumulate ? dst : 0) + sum_j scale_j[n] * (offsets_j ? sum over items_j[offsets_j[n] .. offsets_j[n+1]) of src_j rows
 * : src_j row n), added in term order (the values a chain of nic_segment_sum(accumulate) launches and tensor adds produces). */
#define NIC_SEG_MAX_TERMS 4
typedef struct NicSegTerm {
    const float* src;
    int64_t src_row_stride;
    const int32_t* offsets;   /* [n_dst + 1] or NULL: the term is src's own row n */
    const int32_t* items;
    const float* scale;       /* [n_dst] or NULL */
} NicSegTerm;
int nic_segment_sum_terms(float* dst, int64_t dst_row_stride, const NicSegTerm* terms, int32_t n_terms, int32_t R, int32_t n_dst,
                          int32_t n_scenarios, int32_t ldb, int32_t accumulate, void* stream);
/* The GNN policy's proportional allocation of the warehouse's on-hand stock (neural_networks.py:111-138 via :1435-1492) and
 * its adjoint, one launch each: out [n_edges][ldb] = desired quantity per edge (internal edges 0..S-1 first), members = the
 * internal edges + the self loop e_self (-1: none); sum = sum of the members' rows, ratio = on_hand / (sum + 1e-10),
 * scale = cap_at_one ? min(ratio, 1) : ratio; orders [S+1][ldb]: rows s < S = out[s] * scale, row S = out[e_supplier].
 * Backward: d_out [n_edges][ldb] is fully written, g_on_hand[b] += d(scale)/... (clamp passes where ratio <= 1). */
int nic_gnn_alloc_fwd(const float* out, const float* on_hand, float* orders, float* sums, float* ratio, float* scale, int32_t S,
                      int32_t e_self, int32_t e_supplier, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream);
int nic_gnn_alloc_bwd(const float* out, const float* on_hand, const float* g_orders, const float* sums, const float* ratio,
                      const float* scale, float* d_out, float* g_on_hand, int32_t S, int32_t n_edges, int32_t e_self,
                      int32_t e_supplier, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream);
/* The one-warehouse allocation head FOLLOWED by Simulator.step (neural_networks.py:1435-1492 -> environment.py:110-299) in ONE
 * launch, and the adjoints in reverse order (csrc/gnn_alloc_env.hip): nic_gnn_alloc_fwd + nic_env_step_fwd /
 * nic_env_step_bwd + nic_gnn_alloc_bwd with the same arguments, bit-identical results.  `io` as for nic_env_step_fwd, with its
 * order tables pointing at the rows of `orders` [S + 1][ldb] (store orders, then the warehouse's own) and the warehouse's on-hand
 * slot read from io->wh_inv; one supplying warehouse, no extra echelons.  Backward: g_orders [S + 1][ldb] is scratch written by the
 * env adjoint and read by the head's; g_wh_in receives the env adjoint's warehouse gradient plus the allocation scale's. */
int nic_gnn_alloc_env_fwd(const NicEnvStepIO* io, const float* out, float* orders, float* sums, float* ratio, float* scale,
                          int32_t e_self, int32_t e_supplier, int32_t cap_at_one, float* store_inv_out, float* wh_inv_out, float* reward,
                          void* stream);
int nic_gnn_alloc_env_bwd(const NicEnvStepIO* io, const float* out, const float* sums, const float* ratio, const float* scale,
                          int32_t n_edges, int32_t e_self, int32_t e_supplier, int32_t cap_at_one, const float* g_store_out,
                          const float* g_wh_out, NicTable2 g_reward, float* g_store_in, float* g_wh_in, float* g_orders, float* d_out,
                          void* stream);

/* ---- the GNN policy's whole period in ONE launch (csrc/gnn_period.hip) ------------------------------------------
 * initial_node -> initial_edge -> message aggregation -> node_update -> edge_update -> output (neural_networks.py:1105-1392),
 * and on one-warehouse graphs the proportional allocation + Simulator.step behind them (what nic_gnn_alloc_env_fwd does): a
 * workgroup owns 16 scenarios and keeps their node / edge embeddings in LDS between the five MLPs, so no embedding is read
 * back from HBM inside a period.  What the launches above would have written for the backward (Y of every MLP, the residual
 * sums, the aggregation, the hidden histories in nic_mlp3's NATIVE layout, the pipeline rows of the node features) is written
 * when the pointers are given - nic_mlp3_bwd_hist then runs unchanged on it; an evaluation pass gives none and writes only the
 * desired quantities, the next state and the reward.
 * MLP order: 0 initial_node, 1 initial_edge, 2 node_update, 3 edge_update, 4 output.  Edges [0, n_live) are the ones the edge
 * update / output MLPs are evaluated for (GraphPlan keeps the demand edges last).
 * Packed weights of one MLP (first-layer rows in the order its inputs are contracted in, see ops.gnn_period_pack):
 *   [L1: 2 row blocks x s1q groups x 64 lanes x 4][b1 32][L2: 2 x 2 x 64 x 4][b2 32][L3: nrb3 x 2 x 64 x 4][b3 32 (zero padded)],
 *   fragment (rb, q, lane, j) of layer l = W_l[16 rb + (lane & 15)][k_l(4 q + j, lane >> 4)], 0 where k is past the layer's
 *   inputs; layers 2 / 3: k(s, g) = 16 (s >> 2) + 4 g + (s & 3); layer 1: the same map per 32-row embedding segment, rows
 *   4 s + g for the node features, row 64 at (s = 16, g = 0) for the edge MLP's lead time. */
typedef struct NicGnnPeriodMlp {
    const float* wpk;      /* packed weights, nic_gnn_period_pack_size(s1q, n_out) floats, 16-byte aligned */
    float* Y;              /* [n_out][entities][ldb] (rows row_stride apart) or NULL; the output MLP's [1][n_edges][ldb] is required */
    float* Ysum;           /* node_update: nodes1 = nodes0 + Y; edge_update: edges1 = edges0 + Y; else NULL */
    float* H1;             /* hidden activations in nic_mlp3's native layout (NicMlp3Desc.hist_native), or NULL */
    float* H2;
    int64_t row_stride;    /* elements between feature rows of Y / Ysum */
} NicGnnPeriodMlp;
typedef struct NicGnnPeriod {
    int32_t n_nodes, n_edges, n_live, n_scenarios, ldb;
    int32_t Dn, max_inv;          /* node feature rows; the first max_inv are pipeline slots (zero past a node's own length) */
    int32_t store_feat;           /* 1: write the pipeline rows of `feat` (the backward of initial_node reads them there) */
    int32_t fuse_env;             /* 1: allocation head + env step behind the policy (one warehouse) */
    int32_t e_self, e_supplier, cap_at_one;   /* as nic_gnn_alloc_env_fwd */
    int32_t n_agg_items;                      /* length of agg_items */
    int32_t wb0_floats, wb1_floats, tab_words;   /* set by the library */
    const int32_t* src;           /* [n_edges] source / target node of an edge, -1: the virtual (all-zero) node */
    const int32_t* tgt;
    const int32_t* agg_off;       /* [2 n_nodes + 1]: CSR lists of a node's incoming edges, then of its outgoing edges */
    const int32_t* agg_items;
    const float* agg_scale;       /* [2 n_nodes] 1 / sqrt(degree) */
    const float* lead;            /* [n_edges] per-edge lead-time input */
    const int32_t* node_row0;     /* [n_nodes] first row of the node's pipeline in `state` */
    const int32_t* node_slots;    /* [n_nodes] its length */
    const float* state;           /* [rows][ldb] inventory state before the period */
    float* feat;                  /* [Dn][n_nodes][ldb] node features: static rows (>= max_inv) read, pipeline rows written */
    float* agg;                   /* [32][2 n_nodes][ldb] or NULL */
    NicGnnPeriodMlp mlp[5];
    NicEnvStepIO io;              /* fuse_env: as for nic_gnn_alloc_env_fwd */
    float* orders;                /* [S + 1][ldb] */
    float* sums;
    float* ratio;
    float* scale;
    float* store_out;
    float* wh_out;
    float* reward;
    float* edge_scratch;          /* nic_gnn_period_ok() == 2: ceil(n_scenarios / 16) * n_edges * 512 floats for the edge tiles; else NULL */
} NicGnnPeriod;
int nic_gnn_period_pack_size(int32_t s1q, int32_t n_out);
/* 1 if a graph's embeddings (n_nodes + n_edges tiles of 2 KB) and the staged weights fit in a workgroup's LDS; 2 if only the node
 * tiles do - the edge tiles then live in `edge_scratch` (global memory, L2-resident: ceil(n_scenarios / 16) * n_edges * 512 floats);
 * 0 if neither */
int nic_gnn_period_ok(int32_t n_nodes, int32_t n_edges, int32_t Dn);
int nic_gnn_period_fwd(const NicGnnPeriod* p, void* stream);

/* ---- ... and its BACKWARD in one launch (csrc/gnn_period_bwd.hip) -------------------------------------------------
 * The adjoint of nic_gnn_period_fwd's five MLPs (neural_networks.py:1105-1392 under autograd), behind the env-step / allocation
 * adjoint (nic_gnn_alloc_env_bwd, or nic_env_step_bwd + nic_gnn_alloc_groups_bwd) that leaves d_out [n_edges][ldb] (gradient of
 * the output MLP's desired quantities) and the state gradient g_state [rows][ldb] of the period: replaces five
 * nic_mlp3_bwd_hist launches, three nic_segment_sum_terms launches and the row adds into g_state.  Reads what the forward wrote
 * (Y of every MLP, nodes1 / edges1, the aggregation, the node features, the hidden histories in the native layout), adds the
 * weight gradients of the period to the slab slot of each workgroup (same slabs as nic_mlp3_bwd_hist: [slots][N][lds], bias
 * in column K; slots >= nic_gnn_period_bwd_max_grid()) and the pipeline rows of d features to g_state.  Any graph size: tiles
 * that cross a stage go through `scratch` (nic_gnn_period_bwd_scratch_floats).  A workgroup owns n_sub blocks of 16 scenarios.
 * MLP order as above.  list k: node -> edges it is the source of (k = 0, live edges only; k = 2, all edges) / the target of
 * (k = 1 live, k = 3 all), CSR offsets [n_nodes + 1] + items, in the order the sums are to be taken.
 * Transposed packs (ops.GnnPeriodBwdPack): [L3^T][L2^T][L1^T per 32-row input segment], a 32 x 32 block = fragments
 * (rb, q, lane, j) = M[16 rb + (lane & 15)][16 (s >> 2) + 4 (lane >> 4) + (s & 3)], s = 4 q + j, M = the transposed weights
 * (rows past the inputs zero); the output MLP's 32 x 1 last layer is one contraction step (512 floats). */
typedef struct NicGnnPeriodBwdMlp {
    const float* wpk_t;    /* nic_gnn_period_bwd_pack_size(n_out, segments) floats, 16-byte aligned */
    const float* Y;        /* forward output [n_out][entities][ldb], rows row_stride apart */
    const float* H1;       /* hidden activations, native layout */
    const float* H2;
    int64_t row_stride;
    float* slab1;          /* [slots][32][lds1]: dW1 (columns < K), bias gradient in column K */
    int64_t lds1;
    float* slab2;          /* [slots][32][lds2] */
    int64_t lds2;
    float* slab3;          /* [slots][n_out][lds3] */
    int64_t lds3;
} NicGnnPeriodBwdMlp;
typedef struct NicGnnPeriodBwd {
    int32_t n_nodes, n_edges, n_live, n_scenarios, ldb, Dn;
    int32_t n_sub;                /* 16-scenario blocks per workgroup (1 .. 4) */
    int32_t reserved;
    int32_t n_items[4];           /* lengths of list_items[k] */
    const int32_t* src;           /* [n_edges], -1: the virtual node */
    const int32_t* tgt;
    const float* lead;            /* [n_edges] */
    const int32_t* node_row0;     /* [n_nodes] first row of the node's pipeline in g_state */
    const int32_t* node_slots;
    const float* agg_scale;       /* [2 n_nodes] */
    const int32_t* list_off[4];
    const int32_t* list_items[4];
    const float* feat;            /* [Dn][n_nodes][ldb] */
    const float* nodes0;          /* [32][n_nodes][ldb] = Y of initial_node */
    const float* nodes1;          /* nodes0 + node update */
    const float* edges0;          /* [32][n_edges][ldb] = Y of initial_edge */
    const float* edges1;
    const float* agg;             /* [32][2 n_nodes][ldb] */
    int64_t node_row_stride;      /* elements between rows of feat / nodes0 / nodes1 (agg: twice that) */
    int64_t edge_row_stride;      /* ... of edges0 / edges1 */
    float* d_out;                 /* [n_edges][ldb]: read; written first when fuse_env */
    float* g_state;               /* [rows][ldb], pipeline rows of d features are ADDED */
    float* scratch;
    NicGnnPeriodBwdMlp mlp[5];
    /* fuse_env = 1 (one supplying warehouse): the env-step adjoint + allocation adjoint run in FRONT of the MLP adjoints on this
     * launch's first wavefronts - what nic_gnn_alloc_env_bwd does, same bodies, same arguments - so that d_out and the env part of
     * g_state need no launch of their own: io as for nic_gnn_alloc_env_bwd (its order tables = the rows of the orders the forward
     * wrote), g_store_out / g_wh_out = the state gradient of the NEXT period, g_store_in / g_wh_in = the store / warehouse rows of
     * g_state, g_orders [S + 1][ldb] scratch, sums / ratio / scale as the forward left them, the desired quantities = mlp[4].Y */
    int32_t fuse_env;
    int32_t e_self, e_supplier, cap_at_one;
    NicEnvStepIO io;
    const float* sums;
    const float* ratio;
    const float* scale;
    const float* g_store_out;
    const float* g_wh_out;
    NicTable2 g_reward;
    float* g_store_in;
    float* g_wh_in;
    float* g_orders;
} NicGnnPeriodBwd;
/* pack size: (n_out == 1 ? 512 : 1024) + 1024 (1 + segments) floats; workgroups per launch: at most nic_mlp3_bwd_hist_slots(), the
 * slot count of the slabs both backwards share */
int64_t nic_gnn_period_bwd_scratch_floats(int32_t n_nodes, int32_t n_edges, int32_t n_live, int32_t n_scenarios, int32_t n_sub);
int nic_gnn_period_bwd(const NicGnnPeriodBwd* p, void* stream);

/* The same allocation for SEVERAL supplying nodes (many-warehouse graphs: `_apply_proportional_allocation_to_graph`,
 * neural_networks.py:1435-1492, loops over every node with outgoing edges).  groups [n_groups][4] (device) = {first member edge,
 * member count, self-loop edge or -1, supplier edge} per warehouse - its internal edges are contiguous rows of `out`;
 * on_hand + g * on_hand_group_stride = on-hand row of group g; sums / ratio / scale are [n_groups][ldb]; order_row [n_edges]
 * (device) = row of `orders` (resp. `g_orders`) an edge's quantity is written to (read from), -1 for edges that order nothing.
 * The column an internal edge lands in is the caller's choice - upstream uses "j-th connected warehouse of the store"
 * (:1423-1428), not the warehouse's own index; the host side reproduces that and says so.  Backward writes d_out for every
 * member / self-loop / supplier row and clears rows [zero_first, zero_first + zero_count) (the demand edges). */
int nic_gnn_alloc_groups_fwd(const float* out, const float* on_hand, int64_t on_hand_group_stride, float* orders, float* sums,
                             float* ratio, float* scale, const int32_t* groups, const int32_t* order_row, int32_t n_groups,
                             int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream);
int nic_gnn_alloc_groups_bwd(const float* out, const float* on_hand, int64_t on_hand_group_stride, const float* g_orders,
                             const float* sums, const float* ratio, const float* scale, float* d_out, float* g_on_hand,
                             const int32_t* groups, const int32_t* order_row, int32_t n_groups, int32_t zero_first,
                             int32_t zero_count, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream);
/* dst[r][n][b] (+)= dst_scale[n] * sum_{p in [offsets[n], offsets[n+1])} src[r][items[p]][b] for r < R, in item order
 * (deterministic: no atomics).  Forward: message aggregation over a node's incident edges (:1229-1269, with the
 * 1/sqrt(degree) normalisation :1275-1296 as dst_scale); backward: the adjoint of every gather above. */
int nic_segment_sum(float* dst, int64_t dst_row_stride, const float* src, int64_t src_row_stride, const int32_t* offsets,
                    const int32_t* items, const float* dst_scale /* [n_dst] or NULL */, int32_t R, int32_t n_dst,
                    int32_t n_scenarios, int32_t ldb, int32_t accumulate, void* stream);

/* ---- batched demand sampler -------------------------------------------------------------------------------
 * Replaces Scenario.generate_normal_demand / generate_poisson_demand (data_handling.py:178-211) for synthetic
 * throughput runs: counter-based Philox4x32-10 keyed by (seed, global scenario index, period), so results do not
 * depend on how scenarios are sharded over GPUs.  Output layout [T][S][ldb] (the layout nic_env_step reads).
 *   kind 0: normal   d = mean[s] + sum_j chol[s][j] * z[j]   (chol = lower Cholesky factor of the covariance,
 *                    row-major [S][S]; identity*std for independent stores), clipped at 0 if clip != 0
 *   kind 1: poisson  d ~ Poisson(mean[s]) (inversion by sequential search; mean <= ~30)
 * Statistical (not bitwise) parity with numpy's MT19937 stream — see DESIGN.md. */
int nic_sample_demand(float* out, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb, int64_t scenario_offset,
                      uint64_t seed, int32_t kind, const float* mean /* [S] */, const float* chol /* [S][S] */,
                      int32_t clip, void* stream);
/* The covariance the reference actually builds (data_handling.py:194-201): cov_ij = rho * std_i * std_j (i != j), std_i^2 on
 * the diagonal, 0 <= rho <= 1.  d_s = mean_s + std_s * (sqrt(rho) * z_common + sqrt(1 - rho) * z_s) has exactly that
 * covariance with S + 1 normals per (scenario, period) and no factor matrix (S = 1 or rho = 0: independent stores). */
int nic_sample_demand_equicorrelated(float* out, int32_t T, int32_t S, int32_t n_scenarios, int32_t ldb,
                                     int64_t scenario_offset, uint64_t seed, const float* mean /* [S] */,
                                     const float* std /* [S] */, float rho, int32_t clip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIC_ROLLOUT_H */
