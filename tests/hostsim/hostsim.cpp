// TEST INFRASTRUCTURE: runs the NIC_HD per-scenario bodies of the HIP kernels on the host (plain g++), looping
// over scenarios, so their arithmetic can be checked against the oracle without a GPU.  Never linked into the
// product library; the product has no CPU path.
#include "../../neural_inventory_control_amd/csrc/env_step_body.h"
#include "../../neural_inventory_control_amd/csrc/policy_heads_body.h"
#include "../../neural_inventory_control_amd/csrc/small_rollout_body.h"
#include "../../neural_inventory_control_amd/csrc/closed_form_body.h"

template <int MAXW>
static void fwd_all(const NicEnvStepIO& io, float* so, float* wo, float* eo, float* r) {
    for (int64_t b = 0; b < io.dims.n_scenarios; ++b) nic::env_step_fwd_scenario<MAXW>(io, so, wo, eo, r, b);
}
template <int MAXW>
static void bwd_all(const NicEnvStepIO& io, const float* gso, const float* gwo, const float* geo, NicTable2 gr,
                    float* gsi, float* gwi, float* gei, float* gas, float* gaw, float* gae) {
    for (int64_t b = 0; b < io.dims.n_scenarios; ++b)
        nic::env_step_bwd_scenario<MAXW>(io, gso, gwo, geo, gr, gsi, gwi, gei, gas, gaw, gae, b);
}

template <int NP, int MF, bool CHAIN>
static void closed_form_all(const NicClosedFormDesc& d, float* reward_hist, float* totals, float* state_final, double* g_levels) {
    for (int s = 0; s < d.S; ++s)
        for (int64_t b = 0; b < d.n_scenarios; ++b) {
            float g[NP > 0 ? NP : 1] = {0.f};
            nic::closed_form_chain<NP, MF, CHAIN>(d, reward_hist, totals, state_final, s, b, g);
            for (int j = 0; j < NP; ++j) g_levels[j] += g[j];
        }
}

// One period composed from the ONE-STORE bodies the way csrc/horizon_rollout.hip composes it (a lane per (scenario, store), the
// warehouse lanes summing their shipments themselves, every store lane recomputing its suppliers' on-hand gradient): must equal
// the quad composition above bit for bit.  Settings without extra echelons.
template <int MAXW>
static void fwd_per_store(const NicEnvStepIO& io, float* so, float* wo, float* r) {
    const NicEnvDims& d = io.dims;
    for (int64_t b = 0; b < d.n_scenarios; ++b) {
        const nic::IoAccess a{io, b, so, wo, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        float cst[256], rq[nic::kQuad] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < d.n_stores; ++s) cst[s] = nic::env_fwd_store_t<MAXW>(a, s);
        for (int q = 0; q < nic::kQuad; ++q)
            for (int s = q; s < d.n_stores; s += nic::kQuad) rq[q] += cst[s];
        float total = nic::combine4(rq[0], rq[1], rq[2], rq[3]);
        if (d.n_warehouses > 0) {
            float r_wh = 0.f;
            for (int w = 0; w < d.n_warehouses; ++w) r_wh += nic::env_fwd_warehouse_t<MAXW>(a, w, nic::env_shipped_t(a, w));
            total += r_wh;
        }
        r[b] = total;
    }
}
template <int MAXW>
static void bwd_per_store(const NicEnvStepIO& io, const float* gso, const float* gwo, NicTable2 gr_t, float* gsi, float* gwi,
                          float* gas, float* gaw) {
    const NicEnvDims& d = io.dims;
    for (int64_t b = 0; b < d.n_scenarios; ++b) {
        const nic::IoAccess a{io, b, nullptr, nullptr, gso, gwo, gsi, gwi, gas, gaw};
        const float gr = gr_t.p[b * gr_t.scn_stride];
        float shipped[NIC_MAX_WAREHOUSES];
        for (int w = 0; w < d.n_warehouses; ++w) {
            shipped[w] = nic::env_shipped_t(a, w);
            (void)nic::env_bwd_warehouse_t<MAXW>(a, gr, w, shipped[w]);
        }
        for (int s = 0; s < d.n_stores; ++s)
            nic::env_bwd_store_t<MAXW>(a, gr, [&](int w) { return nic::env_bwd_wh_g_after_t(a, gr, w, shipped[w]); }, s);
    }
}

extern "C" {
int hostsim_env_step_fwd_per_store(const NicEnvStepIO* io, float* so, float* wo, float* r) {
    if (io->dims.n_echelons != 0 || io->dims.n_stores > 256) return 1;
    fwd_per_store<NIC_MAX_SLOTS>(*io, so, wo, r);
    return 0;
}
int hostsim_env_step_bwd_per_store(const NicEnvStepIO* io, const float* gso, const float* gwo, NicTable2 gr, float* gsi, float* gwi,
                                   float* gas, float* gaw) {
    if (io->dims.n_echelons != 0) return 1;
    bwd_per_store<NIC_MAX_SLOTS>(*io, gso, gwo, gr, gsi, gwi, gas, gaw);
    return 0;
}
int hostsim_head_data_driven(const float* Z, const float* wh, const float* mask, const float* g_so, const float* g_wo, float* so,
                             float* wo, float* dZ, float* g_wh, int32_t S, int32_t Wn, int32_t Ww, int32_t B, int32_t ldb) {
    for (int64_t b = 0; b < B; ++b)
        for (int w = 0; w < Wn; ++w) {
            nic::head_data_driven_fwd_one(Z, wh, mask, so, wo, S, Wn, Ww, ldb, b, w);
            nic::head_data_driven_bwd_one(Z, wh, mask, g_so, g_wo, dZ, g_wh, S, Wn, Ww, ldb, b, w);
        }
    return 0;
}
int hostsim_env_step_fwd(const NicEnvStepIO* io, float* so, float* wo, float* eo, float* r) {
    fwd_all<NIC_MAX_SLOTS>(*io, so, wo, eo, r);
    return 0;
}
int hostsim_env_step_bwd(const NicEnvStepIO* io, const float* gso, const float* gwo, const float* geo, NicTable2 gr,
                         float* gsi, float* gwi, float* gei, float* gas, float* gaw, float* gae) {
    bwd_all<NIC_MAX_SLOTS>(*io, gso, gwo, geo, gr, gsi, gwi, gei, gas, gaw, gae);
    return 0;
}
int hostsim_head_warehouse_fwd(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int32_t trans,
                               float* so, float* wo, int32_t S, int32_t Wn, int32_t Ww, int32_t B, int32_t ldb) {
    // up to 64 stores: the quad pieces the device kernel runs (four lanes in a loop, same exchange order); else the one-lane form
    for (int64_t b = 0; b < B; ++b) {
        if (S <= 64) nic::head_warehouse_fwd_quad_scenario<16>(Z, wh_inv, adj, ub, trans, so, wo, S, Wn, Ww, ldb, b);
        else nic::head_warehouse_fwd_scenario(Z, wh_inv, adj, ub, trans, so, wo, S, Wn, Ww, ldb, b);
    }
    return 0;
}
int hostsim_head_warehouse_bwd(const float* Z, const float* wh_inv, const int32_t* adj, float ub, int32_t trans,
                               const float* gso, const float* gwo, float* dZ, float* gwi, int32_t S, int32_t Wn,
                               int32_t Ww, int32_t B, int32_t ldb) {
    for (int64_t b = 0; b < B; ++b) {
        if (S <= 64) nic::head_warehouse_bwd_quad_scenario<16>(Z, wh_inv, adj, ub, trans, gso, gwo, dZ, gwi, S, Wn, Ww, ldb, b);
        else nic::head_warehouse_bwd_scenario(Z, wh_inv, adj, ub, trans, gso, gwo, dZ, gwi, S, Wn, Ww, ldb, b);
    }
    return 0;
}
int hostsim_head_softplus_fwd(const float* Z, float* o, int32_t rows, int32_t B, int32_t ldb) {
    for (int r = 0; r < rows; ++r)
        for (int64_t b = 0; b < B; ++b) o[(int64_t)r * ldb + b] = nic::softplus1_fwd(Z[(int64_t)r * ldb + b]);
    return 0;
}
int hostsim_head_softplus_bwd(const float* Z, const float* g, float* dZ, int32_t rows, int32_t B, int32_t ldb) {
    for (int r = 0; r < rows; ++r)
        for (int64_t b = 0; b < B; ++b)
            dZ[(int64_t)r * ldb + b] = g[(int64_t)r * ldb + b] * nic::softplus1_grad(Z[(int64_t)r * ldb + b]);
    return 0;
}
int hostsim_head_serial_fwd(const float* Z, const float* wh_inv, const float* ech_inv, float ub, float* so, float* wo,
                            float* eo, int32_t E, int32_t Ww, int32_t We, int32_t B, int32_t ldb) {
    for (int64_t b = 0; b < B; ++b) nic::head_serial_fwd_scenario(Z, wh_inv, ech_inv, ub, so, wo, eo, E, Ww, We, ldb, b);
    return 0;
}
int hostsim_head_serial_bwd(const float* Z, const float* wh_inv, const float* ech_inv, float ub, const float* gso,
                            const float* gwo, const float* geo, float* dZ, float* gwi, float* gei, int32_t E,
                            int32_t Ww, int32_t We, int32_t B, int32_t ldb) {
    for (int64_t b = 0; b < B; ++b)
        nic::head_serial_bwd_scenario(Z, wh_inv, ech_inv, ub, gso, gwo, geo, dZ, gwi, gei, E, Ww, We, ldb, b);
    return 0;
}
int hostsim_small_rollout_fwd(const NicSmallRolloutDesc* d, float* rewards, float* state_final, float* states_hist,
                              float* hidden_hist, float* logits_hist) {
    for (int64_t b = 0; b < d->n_scenarios; ++b) {
        if (d->n_hidden == 1) nic::small_rollout_fwd_scenario<1>(*d, rewards, state_final, states_hist, hidden_hist, logits_hist, b);
        else if (d->n_hidden == 2) nic::small_rollout_fwd_scenario<2>(*d, rewards, state_final, states_hist, hidden_hist, logits_hist, b);
        else nic::small_rollout_fwd_scenario<3>(*d, rewards, state_final, states_hist, hidden_hist, logits_hist, b);
    }
    return 0;
}
int hostsim_small_rollout_bwd(const NicSmallRolloutDesc* d, const float* states_hist, const float* hidden_hist,
                              const float* logits_hist, NicTable2 g_reward, float* dz_hidden, float* dz_out) {
    for (int64_t b = 0; b < d->n_scenarios; ++b) {
        if (d->n_hidden == 1) nic::small_rollout_bwd_scenario<1>(*d, states_hist, hidden_hist, logits_hist, g_reward, dz_hidden, dz_out, b);
        else if (d->n_hidden == 2) nic::small_rollout_bwd_scenario<2>(*d, states_hist, hidden_hist, logits_hist, g_reward, dz_hidden, dz_out, b);
        else nic::small_rollout_bwd_scenario<3>(*d, states_hist, hidden_hist, logits_hist, g_reward, dz_hidden, dz_out, b);
    }
    return 0;
}

// g_levels: [n_levels] doubles (sum over all chains of d total / d level_j), or NULL for a forward-only run
int hostsim_closed_form_rollout(const NicClosedFormDesc* d, float* reward_hist, float* totals, float* state_final,
                                double* g_levels) {
    const int np = g_levels ? d->n_levels : 0;
    if (d->policy == NIC_CF_ECHELON) {  // (a host build only needs one slot count per form; the device picks 4 / 8 / 16)
        switch (np) {
            case 0: closed_form_all<0, NIC_MAX_SLOTS, true>(*d, reward_hist, totals, state_final, g_levels); break;
            case 3: closed_form_all<3, NIC_MAX_SLOTS, true>(*d, reward_hist, totals, state_final, g_levels); break;
            case 4: closed_form_all<4, NIC_MAX_SLOTS, true>(*d, reward_hist, totals, state_final, g_levels); break;
            default: closed_form_all<5, NIC_MAX_SLOTS, true>(*d, reward_hist, totals, state_final, g_levels); break;
        }
    } else if (d->Ws <= 4) {
        if (np == 0) closed_form_all<0, 4, false>(*d, reward_hist, totals, state_final, g_levels);
        else if (np == 1) closed_form_all<1, 4, false>(*d, reward_hist, totals, state_final, g_levels);
        else closed_form_all<2, 4, false>(*d, reward_hist, totals, state_final, g_levels);
    } else {
        if (np == 0) closed_form_all<0, NIC_MAX_SLOTS, false>(*d, reward_hist, totals, state_final, g_levels);
        else if (np == 1) closed_form_all<1, NIC_MAX_SLOTS, false>(*d, reward_hist, totals, state_final, g_levels);
        else closed_form_all<2, NIC_MAX_SLOTS, false>(*d, reward_hist, totals, state_final, g_levels);
    }
    return 0;
}
}
