// Fused three-layer 32-wide MLP over gathered inputs and the graph glue around it (include/nic_rollout.h):
//   nic_mlp3_fwd / nic_mlp3_fwd_residual   forward (+ optional residual output)           mlp3_fwd_kernel<KS, CH, ADDR>
//   nic_mlp3_bwd_hist                      backward, weight gradients contracted in-kernel  mlp3_bwd_hist_kernel<KG, GATHER, TAIL>   (what the GNN engine runs)
//   nic_mlp3_bwd / nic_mlp3_bwd_fused      backward with a dz history / without any history (kept as references and as the fallback)
//   nic_segment_sum / nic_segment_sum_terms  message aggregation and the adjoints of every gather / residual connection
//   nic_gnn_alloc_fwd / nic_gnn_alloc_bwd  proportional allocation head
//
// One wavefront = 32-scenario chunks of ONE entity (node or edge), so every gather index is wave-uniform.  The layers run on the
// matrix cores exactly like csrc/small_rollout.hip: A = weights (lane l holds W[i = l & 31][kk = l >> 5]), B = activations
// (lane = scenario), and MFMA step s of layers 2 / 3 is defined to contract over k = crow(s, h) - the row the previous layer's
// accumulator register s holds - so activations never move between layers.  Both halves of the wave carry the same 32 scenarios
// and own different feature rows; every history row is written once.  FP32 (v_mfma_f32_32x32x2_f32: exact products, f32
// accumulate); ELU as in the other kernels (small_rollout_body.h).
// What the compiled code taught (DESIGN.md §4 "reading the ISA"): resolve the K input-row pointers across the LANES (the scalar
// unit was the bottleneck), load in unconditional batches (a conditional load in a loop is a dependent round trip per iteration),
// keep loop-invariant scalar offsets / lane masks from being hoisted into spilled SGPRs, address history buffers through buffer
// descriptors with one 32-bit lane offset each.
#include "gnn_alloc_body.h"
#include "nic_common.h"
#include "small_rollout_body.h"

namespace {
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// The segment table of one entity, resolved ONCE per wavefront into scalars (base pointer of the entity's rows per segment,
// first input row of each segment): finding the pointer of input row k afterwards is a few scalar compares and selects, no
// memory access.  (Resolving the table per row cost three dependent scalar loads per row: the kernel was bound by them.)
struct SegRegs {
    const float* base[NIC_MLP3_MAX_SEGS];   // nullptr: the virtual (all-zero) node
    int64_t row_stride[NIC_MLP3_MAX_SEGS];
    int scn[NIC_MLP3_MAX_SEGS];
    int start[NIC_MLP3_MAX_SEGS + 1];
};
__device__ __forceinline__ SegRegs resolve_segments(const NicMlp3Desc& d, int e) {
    SegRegs R;
    R.start[0] = 0;
#pragma unroll
    for (int q = 0; q < NIC_MLP3_MAX_SEGS; ++q) {
        R.base[q] = nullptr;
        R.row_stride[q] = 0;
        R.scn[q] = 0;
        R.start[q + 1] = R.start[q];
        if (q < d.n_segs) {
            const NicMlp3Seg& S = d.seg[q];
            const int ent = S.map ? S.map[e] : e;
            R.base[q] = ent >= 0 ? S.base + (int64_t)ent * S.ent_stride : nullptr;
            R.row_stride[q] = S.row_stride;
            R.scn[q] = (int)S.scn_stride;
            R.start[q + 1] = R.start[q] + S.n_rows;
        }
    }
    return R;
}
struct RowRef {
    const float* p;
    int64_t scn;
};
__device__ __forceinline__ RowRef input_row(const SegRegs& R, int k) {
    RowRef out{nullptr, 0};
#pragma unroll
    for (int q = 0; q < NIC_MLP3_MAX_SEGS; ++q)
        if (k >= R.start[q] && k < R.start[q + 1]) {
            out.p = R.base[q] ? R.base[q] + (int64_t)(k - R.start[q]) * R.row_stride[q] : nullptr;
            out.scn = R.scn[q];
        }
    return out;
}

// Row pointers computed ACROSS THE LANES: lane l evaluates the segment table for input row `row0 + l` on the vector unit (a
// few dozen instructions for 64 rows at once) and every contraction step fetches the pointer it needs from lane 2s + h with one
// lane permute per 32-bit half.  Resolving each row on the scalar unit (input_row) is ~80 SALU instructions per row; with 96 rows
// per wavefront the forward kernel was bound by scalar issue (one slot per SIMD every four cycles), not by memory or the matrix
// pipe.  Encoding: a 4-byte aligned address with bit 0 = the row is indexed by the scenario (scn_stride 1); rows that do not
// exist (past K, or of the virtual all-zero node) point at a zero constant with bit 0 clear.
__device__ const float kZeroRow = 0.f;
__device__ __forceinline__ uint64_t encoded_row_pointer(const NicMlp3Desc& d, int e, int k) {
    uint64_t enc = reinterpret_cast<uint64_t>(&kZeroRow);
    int start = 0;
#pragma unroll
    for (int q = 0; q < NIC_MLP3_MAX_SEGS; ++q) {
        if (q < d.n_segs) {
            const NicMlp3Seg& S = d.seg[q];
            const int ent = S.map ? S.map[e] : e;
            const bool in = k >= start && k < start + S.n_rows && ent >= 0;
            const uint64_t cand = reinterpret_cast<uint64_t>(S.base + (int64_t)ent * S.ent_stride + (int64_t)(k - start) * S.row_stride) |
                                  (S.scn_stride ? 1ull : 0ull);
            enc = in ? cand : enc;
            start += S.n_rows;
        }
    }
    return enc;
}
__device__ __forceinline__ uint64_t lane_fetch64(uint64_t v, int src_lane) {
    const unsigned lo = __shfl((unsigned)v, src_lane), hi = __shfl((unsigned)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
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

constexpr int kWaves = 4;   // wavefronts per workgroup (they share one staged copy of the weights)
constexpr int kChunks = 1;  // 32-scenario chunks each wavefront walks (1: nothing is loop-invariant, so the weight fragments are read from LDS as they are used instead of being hoisted into ~130 VGPRs, which left one wavefront per SIMD)

// KS = MFMA steps of the first layer (input rows 2s + h): K <= 2 * KS.
// Independent wavefronts (four per workgroup, adjacent chunks; no LDS, no barriers): the weight fragments are read from the PRE-TRANSPOSED copy (`weights_t`: lane i of step
// k reads consecutive words - one 128-B line per half; the natural layout would touch 32 lines per read) right where they are
// used, so the kernel stays at 112 VGPRs and four wavefronts per SIMD hide the gather latency.  (Variants that staged
// the weights in LDS for four wavefronts let the compiler hoist every fragment into registers - 300+ VGPRs or scratch - and
// were 2-4x slower; measured.)
#ifndef NIC_FWD_WAVES
#define NIC_FWD_WAVES 4
#endif
constexpr int kFwdWaves = NIC_FWD_WAVES;
#ifndef NIC_FWD_PAIR_MIN_WAVES
#define NIC_FWD_PAIR_MIN_WAVES 2048
#endif
constexpr int kFwdPairMinWaves = NIC_FWD_PAIR_MIN_WAVES;   // (below: not enough work to care; one chunk keeps more CUs busy)
constexpr int kFwdSlots = 256 * 4 * 5;
// CH = adjacent 32-scenario chunks one wavefront owns (1 or 2).  With two, every weight fragment, bias and row pointer is fetched
// once for both, each row is touched as 2 x 128 B back to back, and a wavefront keeps twice the bytes in flight: the memory
// pattern alone (tools/layout_probe.hip, "pair") moves the same bytes 20-28 % faster than one chunk per wavefront.
// ADDR = how the outputs are addressed.  The SQ counters (tools/pmc_sq_probe.sh) show this kernel bound by vector-ALU issue, not
// by memory: 2,067 VALU instructions per wavefront for 80 MFMAs, a quarter of them 64-bit address arithmetic of the stores.
//   kAddrBuf / kAddrBufX: every output buffer is written through a raw buffer descriptor based at the wavefront's first column -
//     ONE 32-bit lane offset per buffer, the row as a scalar offset (needs rows that span < 2 GiB; the launcher checks) - without
//     (kAddrBuf: no per-step branch either) or with (kAddrBufX) the stored copy of the inputs;
//   kAddrFlat: 64-bit addresses and a run-time X history test, for buffers too large for 32-bit offsets.
constexpr int kAddrFlat = 0, kAddrBuf = 1, kAddrBufX = 2;
template <int KS, int CH, int ADDR>
__global__ __launch_bounds__(64 * kFwdWaves) void mlp3_fwd_kernel(NicMlp3Desc d, const float* __restrict__ wt, float* __restrict__ Y,
                                                      float* __restrict__ Xh, float* __restrict__ H1, float* __restrict__ H2,
                                                      const float* __restrict__ Rsd, float* __restrict__ Ysum) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, i = j;
    const int e = blockIdx.y, K = d.K;
    const int chunk = (blockIdx.x * kFwdWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * CH;
    if ((int64_t)chunk * 32 >= d.n_scenarios) return;
    bool live[CH];
    int64_t b[CH], col[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int64_t b_raw = (int64_t)(chunk + c) * 32 + j;
        live[c] = b_raw < d.n_scenarios;
        b[c] = live[c] ? b_raw : 0;
        col[c] = (int64_t)e * d.ldb + b[c];
    }
    const bool full = (int64_t)(chunk + CH) * 32 <= d.n_scenarios;   // every lane of every chunk is a scenario
    const int64_t ent_ld = d.ent_row_stride ? d.ent_row_stride : (int64_t)d.n_entities * d.ldb;   // elements between feature rows of the [rows][E][ldb] buffers
    // ... and of the history buffers.  hist_native: H1 / H2 are private to the forward / backward pair of kernels and kept per
    // (entity, 32-scenario chunk) as one contiguous [32 rows][32 scenarios] block (4 KB) instead of 32 pieces of 128 B that lie
    // n_entities * ldb floats apart - same bytes, one DRAM page per wavefront access instead of 32
    const bool nat = d.hist_native != 0;
    const int64_t hs = nat ? 32 : (d.hist_row_stride ? d.hist_row_stride : ent_ld);
    // buffer addressing: descriptor base = column 0 of the wavefront's first chunk in row 0 of the entity; a lane adds
    // (4 h rows + its column) once, a store adds the row as a scalar
    constexpr bool BUF = ADDR != kAddrFlat;
    const int64_t col0 = (int64_t)e * d.ldb + (int64_t)chunk * 32;
    const int64_t hcol0 = nat ? ((int64_t)e * (d.ldb / 32) + chunk) * 1024 : col0;
    auto rsrc_of = [](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rH1 = rsrc_of(BUF && H1 ? H1 + hcol0 : wt), rH2 = rsrc_of(BUF && H2 ? H2 + hcol0 : wt),
                                 rY = rsrc_of(BUF ? Y + col0 : wt), rYs = rsrc_of(BUF && Ysum ? Ysum + col0 : wt),
                                 rR = rsrc_of(BUF && Rsd ? Rsd + col0 : wt), rXh = rsrc_of(ADDR == kAddrBufX ? Xh + col0 : wt);
    const int hs4 = (int)hs * 4, el4 = (int)ent_ld * 4;
    const int vo_h = 4 * h * hs4, vo_e = 4 * h * el4, vo_x = h * hs4;
    int vc[CH];   // the lane's column inside the wavefront's chunks, in bytes (lanes past the last scenario: column 0, never stored)
    int vch[CH];  // ... inside the history buffers (native: the next chunk is the next 4-KB block)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        vc[c] = live[c] ? c * 128 + j * 4 : 0;
        vch[c] = nat ? (live[c] ? c * 4096 + j * 4 : 0) : vc[c];
    }
    auto put = [](__amdgpu_buffer_rsrc_t r, float v, int voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
    };
    const float* w1t = wt;                 // [K][32]
    const float* b1 = w1t + K * 32;
    const float* w2t = b1 + 32;            // [32][32]
    const float* b2 = w2t + 32 * 32;
    const float* w3t = b2 + 32;            // [32][32], columns >= n_out zero
    const float* b3 = w3t + 32 * 32;       // padded to 32
    const uint64_t pv0 = encoded_row_pointer(d, e, lane), pv1 = (2 * KS > 64) ? encoded_row_pointer(d, e, 64 + lane) : 0ull;
    f32x16 acc[CH];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = b1[crow(r, h)];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c][r] = bv;
    }
    // First layer in groups of kG contraction steps: the gathers and weight fragments of a group are loaded UNCONDITIONALLY
    // (a missing row reads a valid dummy address and is zeroed by a select) and all together, the next group's while this
    // group's MFMAs run.  Written as one step at a time - `x = p ? p[..] : 0; a = k < K ? w[..] : 0; mfma` - every step was
    // load, s_waitcnt vmcnt(0), MFMA behind two branches: 48 dependent memory round trips per wavefront.
    constexpr int kG = CH == 1 ? 16 : 8, NG = (KS + kG - 1) / kG;
    float xg[2][CH][kG], ag[2][kG];
    auto load_group = [&](int g, float (&x)[CH][kG], float (&a)[kG]) {
#pragma unroll
        for (int u = 0; u < kG; ++u) {
            const int s = g * kG + u;
            if (s < KS) {
                const int k = 2 * s + h;
                const uint64_t enc = lane_fetch64(2 * s < 64 ? pv0 : pv1, (2 * s + h) & 63);
                const float* q = reinterpret_cast<const float*>(enc & ~3ull);
                const float av = w1t[(k < K ? k : 0) * 32 + i];
#pragma unroll
                for (int c = 0; c < CH; ++c) x[c][u] = q[(enc & 1ull) ? b[c] : 0];
                a[u] = k < K ? av : 0.f;
            }
        }
    };
    load_group(0, xg[0], ag[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load_group(g + 1, xg[(g + 1) & 1], ag[(g + 1) & 1]);
#pragma unroll
        for (int u = 0; u < kG; ++u) {
            const int s = g * kG + u;
            if (s < KS) {
                const int k = 2 * s + h;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[g & 1][u], xg[g & 1][c][u], acc[c], 0, 0, 0);
                    if (ADDR == kAddrFlat) {
                        if (Xh && live[c] && k < K) Xh[(int64_t)k * hs + col[c]] = xg[g & 1][c][u];
                    } else if (ADDR == kAddrBufX) {
                        if (live[c] && k < K) put(rXh, xg[g & 1][c][u], vo_x + vc[c], 2 * s * hs4);
                    }
                }
            }
        }
    }
    float hcur[CH][16];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hcur[c][r] = nic::elu1(acc[c][r]);
    }
    // the two hidden activations the backward needs.  (`full` is wave-uniform: the stores of the two chunks alternate, no
    // per-store lane masks)
    auto put_hidden = [&](float* H, __amdgpu_buffer_rsrc_t rH) {
        if (!H) return;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (BUF) put(rH, hcur[c][r], vo_h + vch[c], ((r & 3) + 8 * (r >> 2)) * hs4);
                    else H[(int64_t)crow(r, h) * hs + (nat ? hcol0 + c * 1024 + j : col[c])] = hcur[c][r];
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (live[c]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (BUF) put(rH, hcur[c][r], vo_h + vch[c], ((r & 3) + 8 * (r >> 2)) * hs4);
                        else H[(int64_t)crow(r, h) * hs + (nat ? hcol0 + c * 1024 + j : col[c])] = hcur[c][r];
                    }
                }
            }
        }
    };
    put_hidden(H1, rH1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = b2[crow(r, h)];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c][r] = bv;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float wv = w2t[crow(s, h) * 32 + i];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, hcur[c][s], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hcur[c][r] = nic::elu1(acc[c][r]);
    }
    put_hidden(H2, rH2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = b3[crow(r, h)];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c][r] = bv;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float wv = w3t[crow(s, h) * 32 + i];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, hcur[c][s], acc[c], 0, 0, 0);
    }
    const int n_out = d.n_out, out_act = d.out_act;
    if (Ysum) {   // residual connection folded in: Ysum = out + R (the same rounding as the separate tensor add it replaces)
        float rs[CH][16];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_s = (r & 3) + 8 * (r >> 2);
                if (BUF)   // rows past n_out: the lane's column in row 0 (read, never used)
                    rs[c][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rR, (row_s + 4 * h < n_out ? vo_e + row_s * el4 : 0) + vc[c], 0, 0));
                else
                    rs[c][r] = Rsd[(int64_t)(crow(r, h) < n_out ? crow(r, h) : 0) * ent_ld + col[c]];
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (live[c]) {
                float y[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = out_act_fwd(out_act, acc[c][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (crow(r, h) < n_out) {
                        if (BUF) {
                            put(rY, y[r], vo_e + vc[c], ((r & 3) + 8 * (r >> 2)) * el4);
                            put(rYs, rs[c][r] + y[r], vo_e + vc[c], ((r & 3) + 8 * (r >> 2)) * el4);
                        } else {
                            Y[(int64_t)crow(r, h) * ent_ld + col[c]] = y[r];
                            Ysum[(int64_t)crow(r, h) * ent_ld + col[c]] = rs[c][r] + y[r];
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (live[c]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (crow(r, h) < n_out) {
                        const float y = out_act_fwd(out_act, acc[c][r]);
                        if (BUF) put(rY, y, vo_e + vc[c], ((r & 3) + 8 * (r >> 2)) * el4);
                        else Y[(int64_t)crow(r, h) * ent_ld + col[c]] = y;
                    }
                }
            }
        }
    }
}

// ---- history-free backward: gather again, recompute the hidden layers, contract the weight gradients in the kernel -------------
// Layouts: the MLP chain runs "column-owner" (lane (h, c) holds rows crow(r, h) of column c: MFMA C layout / B operand); a weight
// gradient dW[i][k] = sum_c dz[i][c] act[k][c] contracts over the COLUMNS, so both of its operands must be "row-owner" (lane
// (hh, row) holds columns 2s + hh).  Each 32 x 32 tile changes layout through a wave-private LDS tile with 33-word rows
// (conflict-free both ways); 5 + KG tiles per chunk.  The dW accumulators (16 registers per 32 x 32 block) stay in registers over
// all chunks a wavefront walks.
constexpr int kFusedWaves = 4;     // wavefronts per workgroup
constexpr int kFusedBlocks = 512;  // persistent workgroups (two per CU): one slab slot per wavefront
constexpr int kTile = 32 * 33;

__device__ __forceinline__ void to_row_owner(float* tile, const float (&v)[16], float (&out)[16], int h, int j) {
    // in: lane (h, c = j) holds rows crow(r, h) of column c;  out: lane (hh = h, row = j) holds columns 2s + hh of its row
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[crow(r, h) * 33 + j] = v[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) out[s2] = tile[j * 33 + 2 * s2 + h];
    __builtin_amdgcn_wave_barrier();
}

template <int KG>
__global__ __launch_bounds__(64 * kFusedWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp3_bwd_fused_kernel(
    NicMlp3Desc d, const float* __restrict__ wt, const float* __restrict__ w, const float* __restrict__ dY,
    const float* __restrict__ Yo, float* __restrict__ dX, float* __restrict__ slab1, int64_t lds1, float* __restrict__ slab2,
    int64_t lds2, float* __restrict__ slab3, int64_t lds3) {
    __shared__ float tiles[kFusedWaves * kTile];
    // every weight fragment both directions need, staged once per workgroup (with one wavefront per SIMD reading them from L2
    // next to each MFMA, every MFMA waited ~0.5 us for its operand: 1 ms per launch)
    __shared__ float sw1t[32 * KG * 32], sw2t[32 * 32], sW1[32 * 32 * KG], sW2[32 * 32], sW3[32 * 32], sb[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5, i = j;
    float* tile = tiles + wv * kTile;
    const int K = d.K;
    const int64_t ent_ld = d.ent_row_stride ? d.ent_row_stride : (int64_t)d.n_entities * d.ldb;
    {
        const float* T = wt;                     // [W1^T K x 32][b1][W2^T][b2]...
        const float* T2 = T + K * 32 + 32;
        const float* N1 = w;                     // natural [W1 32 x K][b1][W2][b2][W3 n_out x 32]
        const float* N2 = N1 + 32 * K + 32;
        const float* N3 = N2 + 32 * 32 + 32;
        for (int idx = threadIdx.x; idx < 32 * KG * 32; idx += 64 * kFusedWaves) {
            sw1t[idx] = idx < K * 32 ? T[idx] : 0.f;
            const int n = idx / (32 * KG), k = idx % (32 * KG);
            sW1[idx] = k < K ? N1[n * K + k] : 0.f;
        }
        for (int idx = threadIdx.x; idx < 32 * 32; idx += 64 * kFusedWaves) {
            sw2t[idx] = T2[idx];
            sW2[idx] = N2[idx];
            sW3[idx] = (idx >> 5) < d.n_out ? N3[idx] : 0.f;
        }
        if (threadIdx.x < 32) {
            sb[threadIdx.x] = T[K * 32 + threadIdx.x];
            sb[32 + threadIdx.x] = T2[32 * 32 + threadIdx.x];
        }
    }
    __syncthreads();
    const float* w1t_ = sw1t;
    const float* b1_ = sb;
    const float* w2t_ = sw2t;
    const float* b2_ = sb + 32;
    const float* W1_ = sW1;
    const float* W2_ = sW2;
    const float* W3_ = sW3;
    constexpr int KS = 16 * KG;                  // first-layer MFMA steps (rows 2s + h), K <= 32 * KG

    f32x16 g1[KG], g2, g3;                       // dW1 (KG column blocks), dW2, dW3: rows crow(r, h), column = lane j
    float gb1 = 0.f, gb2 = 0.f, gb3 = 0.f;       // bias gradients of row i = j (row-owner sums)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        g2[r] = 0.f;
        g3[r] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) g1[g][r] = 0.f;
    }
    const int chunks = (d.n_scenarios + 31) / 32;
    const int n_items = d.n_entities * chunks;
    const int wave_id = blockIdx.x * kFusedWaves + wv, n_waves = gridDim.x * kFusedWaves;
#pragma unroll 1
    for (int item = wave_id; item < n_items; item += n_waves) {
        const int e = item / chunks, ch = item - e * chunks;
        const int64_t b_raw = (int64_t)ch * 32 + j;
        const bool live = b_raw < d.n_scenarios;
        const int64_t b = live ? b_raw : 0;
        const int64_t col = (int64_t)e * d.ldb + b;
        const SegRegs segs = resolve_segments(d, e);
        // The weights are the same for every item, so the compiler would hoist all ~180 fragment loads out of this loop into
        // registers (512 VGPRs + scratch, one wavefront per SIMD).  An offset it cannot see through keeps each load next to
        // its MFMA (they hit L1 / L2), and the kernel inside 256 VGPRs.
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* w1t = w1t_ + opaque;
        const float* b1 = b1_ + opaque;
        const float* w2t = w2t_ + opaque;
        const float* b2 = b2_ + opaque;
        const float* W1 = W1_ + opaque;
        const float* W2 = W2_ + opaque;
        const float* W3 = W3_ + opaque;
        // gather the inputs (B-operand layout: lane (h, c) holds row 2s + h) and recompute the hidden layers
        // (the inputs are NOT kept in registers until the dW1 contraction at the end of the item - 48 VGPRs that pushed the
        // kernel into scratch; they are gathered a second time there, from L2)
        auto gather = [&](int s) -> float {
            const RowRef r0 = input_row(segs, 2 * s), r1 = input_row(segs, 2 * s + 1);
            const float* p = h ? r1.p : r0.p;
            const int64_t scn = h ? r1.scn : r0.scn;
            return p ? p[b * scn] : 0.f;
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b1[crow(r, h)];
#pragma unroll 2
        for (int s = 0; s < KS; ++s) {  // (rolled: unrolled, all 48 row pointers and loads are hoisted to the top)
            const int k = 2 * s + h;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1t[k * 32 + i], gather(s), acc, 0, 0, 0);
        }
        float h1[16], h2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[r] = nic::elu1(acc[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b2[crow(r, h)];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w2t[crow(s, h) * 32 + i], h1[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[r] = nic::elu1(acc[r]);
        // output layer: dz3 = dY * act'(y)
        float dz[16], at[16], bt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = crow(r, h), rr = row < d.n_out ? row : 0;
            const float gy = dY[(int64_t)rr * ent_ld + col], yo = Yo[(int64_t)rr * ent_ld + col];
            dz[r] = (row < d.n_out && live) ? gy * out_act_grad(d.out_act, yo) : 0.f;
        }
        to_row_owner(tile, dz, at, h, j);
        to_row_owner(tile, h2, bt, h, j);
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                g3 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bt[s], g3, 0, 0, 0);
                sb += at[s];
            }
            gb3 += sb + __shfl_xor(sb, 32);
        }
        // dH2 = W3^T dz3 -> dz2
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W3[crow(s, h) * 32 + i], dz[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = acc[r] * nic::elu1_grad_from_out(h2[r]);
        to_row_owner(tile, dz, at, h, j);
        to_row_owner(tile, h1, bt, h, j);
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                g2 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bt[s], g2, 0, 0, 0);
                sb += at[s];
            }
            gb2 += sb + __shfl_xor(sb, 32);
        }
        // dH1 = W2^T dz2 -> dz1
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W2[crow(s, h) * 32 + i], dz[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = acc[r] * nic::elu1_grad_from_out(h1[r]);
        to_row_owner(tile, dz, at, h, j);
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sb += at[s];
            gb1 += sb + __shfl_xor(sb, 32);
        }
        // dW1 block g: the gathered inputs of rows 32g .. 32g+31, from the B-operand layout (row 2s + h) to row-owner
#pragma unroll
        for (int g = 0; g < KG; ++g) {
#pragma unroll 4
            for (int u = 0; u < 16; ++u) tile[(2 * u + h) * 33 + j] = gather(16 * g + u);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bt[s2] = tile[j * 33 + 2 * s2 + h];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) g1[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bt[s], g1[g], 0, 0, 0);
        }
        // dX = W1^T dz1
        if (dX) {
#pragma unroll
            for (int g = 0; g < KG; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[crow(s, h) * 32 * KG + 32 * g + i], dz[s], acc, 0, 0, 0);
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
    // this wavefront's slab slot: dW[n = crow(r, h)][k = 32g + j] (+)=, bias column K (row-owner sums: lane (0, i) has row i)
    float* s1 = slab1 + (int64_t)wave_id * 32 * lds1;
    float* s2 = slab2 + (int64_t)wave_id * 32 * lds2;
    float* s3 = slab3 + (int64_t)wave_id * d.n_out * lds3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = crow(r, h);
#pragma unroll
        for (int g = 0; g < KG; ++g)
            if (32 * g + j < K) s1[(int64_t)n * lds1 + 32 * g + j] += g1[g][r];
        s2[(int64_t)n * lds2 + j] += g2[r];
        if (n < d.n_out) s3[(int64_t)n * lds3 + j] += g3[r];
    }
    if (h == 0) {
        s1[(int64_t)i * lds1 + K] += gb1;
        s2[(int64_t)i * lds2 + 32] += gb2;
        if (i < d.n_out) s3[(int64_t)i * lds3 + 32] += gb3;
    }
}

// ---- backward with in-kernel weight gradients over the STORED activations ------------------------------------------------------
// The history path above writes the three pre-activation gradients of every column (64 + n_out rows) for a later contraction
// kernel that reads them back together with X / H1 / H2.  Here the contraction happens where dz is produced: dz changes to the
// row-owner layout through a wave-private LDS tile (A operand), the stored activations are read a second time straight from
// global memory in row-owner form (lane (hh, row) takes the 16 consecutive columns hh*16 .. hh*16+15 of its row: four 16-B
// loads, the lines were just fetched by the column-owner reads), and the dW accumulators stay in registers over all the
// (entity, chunk) items a wavefront walks.  No recomputation, no re-gather (what made mlp3_bwd_fused_kernel slow), no dz
// history.  A workgroup's four wavefronts add their accumulators through LDS in a fixed order into ONE slab slot.
constexpr int kHistWaves = 4;
constexpr int kHistBlocks = 512;   // persistent workgroups (two per CU)

// TAIL: input rows beyond the KG full 32-row blocks (K = 32 KG + TAIL, TAIL <= 4; e.g. the edge MLP's [source | target | lead
// time] = 65 rows).  A 32-row MFMA block for one row costs 32 MFMAs per item (a fifth of the kernel); the tail rows' weight-
// gradient column and input gradient are ~40 vector instructions each instead.
constexpr int kMaxTail = 4;
template <int KG, bool GATHER, int TAIL>
__global__ __launch_bounds__(64 * kHistWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp3_bwd_hist_kernel(
    NicMlp3Desc d, const float* __restrict__ weights, const float* __restrict__ dY, const float* __restrict__ Yo,
    const float* __restrict__ Xh, const float* __restrict__ H1, const float* __restrict__ H2, float* __restrict__ dX,
    float* __restrict__ slab1, int64_t lds1, float* __restrict__ slab2, int64_t lds2, float* __restrict__ slab3, int64_t lds3) {
    __shared__ float sW[32 * 32 * (KG + 2)];            // W1 [32][32*KG] | W2 [32][32] | W3 [32][32]; reused for the reduction
    __shared__ float tiles[kHistWaves * kTile];
    __shared__ float sbias[3 * 32];
    __shared__ float sWt[32 * kMaxTail], stail[32 * kMaxTail];   // tail columns of W1 [n][tail]; their reduced weight gradients
    float* const sW1 = sW;
    float* const sW2 = sW + 32 * 32 * KG;
    float* const sW3 = sW2 + 32 * 32;
    const int K = d.K;
    {
        const float* W1 = weights;
        const float* W2 = W1 + 32 * K + 32;
        const float* W3 = W2 + 32 * 32 + 32;
        // (all loads of a thread first, then its LDS stores: as `s[idx] = cond ? W[..] : 0` in a loop each load was waited for
        // before the next was issued - 4 (KG + 2) dependent round trips at the head of every launch)
        constexpr int NT = 64 * kHistWaves, IT1 = 32 * 32 * KG / NT, IT2 = 32 * 32 / NT;
        float v1[IT1], v2[IT2], v3[IT2];
#pragma unroll
        for (int u = 0; u < IT1; ++u) {
            const int idx = threadIdx.x + u * NT, n = idx / (32 * KG), k = idx % (32 * KG);
            v1[u] = W1[k < K ? n * K + k : 0];
        }
#pragma unroll
        for (int u = 0; u < IT2; ++u) {
            const int idx = threadIdx.x + u * NT;
            v2[u] = W2[idx];
            v3[u] = W3[(idx >> 5) < d.n_out ? idx : 0];
        }
#pragma unroll
        for (int u = 0; u < IT1; ++u) {
            const int idx = threadIdx.x + u * NT, k = idx % (32 * KG);
            sW1[idx] = k < K ? v1[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < IT2; ++u) {
            const int idx = threadIdx.x + u * NT;
            sW2[idx] = v2[u];
            sW3[idx] = (idx >> 5) < d.n_out ? v3[u] : 0.f;
        }
        if (TAIL > 0 && threadIdx.x < 32 * TAIL) sWt[threadIdx.x] = W1[(threadIdx.x / TAIL) * K + 32 * KG + threadIdx.x % TAIL];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, h = lane >> 5, i = j;
    float* tile = tiles + wv * kTile;
    const int64_t ent_ld = d.ent_row_stride ? d.ent_row_stride : (int64_t)d.n_entities * d.ldb;
    const bool nat = d.hist_native != 0;   // H1 / H2 as one [32][32] block per (entity, chunk): see mlp3_fwd_kernel
    const int64_t hs = nat ? 32 : (d.hist_row_stride ? d.hist_row_stride : ent_ld);

    f32x16 g1[KG], g2, g3;
    float gb1 = 0.f, gb2 = 0.f, gb3 = 0.f;
    float gtail[TAIL > 0 ? TAIL : 1];   // lane (h, n): sum over its 16 columns of dz1[n][c] x_tail[c]
#pragma unroll
    for (int tt = 0; tt < (TAIL > 0 ? TAIL : 1); ++tt) gtail[tt] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        g2[r] = 0.f;
        g3[r] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) g1[g][r] = 0.f;
    }
    const int chunks = (d.n_scenarios + 31) / 32;
    const int n_items = d.n_entities * chunks;
    const int wave_id = blockIdx.x * kHistWaves + wv, n_waves = gridDim.x * kHistWaves;
    // Addressing: every buffer is read through a raw buffer descriptor whose base is the item's first column (wave-uniform);
    // a lane contributes ONE 32-bit byte offset per buffer and the row of each load is a scalar offset - 64-bit per-load
    // addresses (two VGPRs each, ~100 loads per item) were what pushed the first version into scratch.
    auto rsrc_of = [](const float* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
    };
    auto ld1 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    // row-owner operand of a weight-gradient contraction: the 16 consecutive columns h*16 .. h*16+15 of the lane's row
    auto row_owner = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&out)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16 * q, soff, 0));
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
    };
    // column-owner registers (rows crow(r, h) of column j) -> row-owner A operand (row j, columns h*16 + s)
    auto transpose = [&](const float (&v)[16], float (&out)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[crow(r, h) * 33 + j] = v[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) out[s2] = tile[j * 33 + h * 16 + s2];
        __builtin_amdgcn_wave_barrier();
    };
    // Without an X history (X_hist == NULL) the inputs are read again from the SOURCE buffers, directly in row-owner form: lane
    // (h, j) owns input row k and resolves that row's pointer from the segment table itself (entity index wave-uniform, a few
    // selects per segment), then takes the same 16 consecutive columns.  The sources are per-period buffers, so they still hold
    // what the forward read; the K-row copy the forward would write and this kernel would read back is 27 % of the GNN's traffic.
    auto gather_row_owner = [&](int k, int e, int ch, float (&out)[16]) {
        const float* rp = nullptr;
        int scn = 1, start = 0;
#pragma unroll
        for (int q = 0; q < NIC_MLP3_MAX_SEGS; ++q) {
            if (q < d.n_segs) {
                const NicMlp3Seg& S = d.seg[q];
                const int ent = S.map ? S.map[e] : e;
                const bool in = k >= start && k < start + S.n_rows;
                const float* cand = S.base + (int64_t)ent * S.ent_stride + (int64_t)(k - start) * S.row_stride;
                if (in) {
                    rp = ent >= 0 ? cand : nullptr;
                    scn = (int)S.scn_stride;
                }
                start += S.n_rows;
            }
        }
        if (rp == nullptr) {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) out[s2] = 0.f;
        } else if (scn) {
            const float4* p4 = reinterpret_cast<const float4*>(rp + (int64_t)ch * 32 + h * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = p4[q];
                out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
            }
        } else {  // per-entity constant (an edge's lead time)
            const float v = *rp;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) out[s2] = v;
        }
    };
    const int hs4_ = (int)hs * 4, el4_ = (int)ent_ld * 4;    // row strides in bytes (checked by the launcher to fit)
    const int out_act = d.out_act, n_out_ = d.n_out;
#pragma unroll 1
    for (int item = wave_id; item < n_items; item += n_waves) {
        const int e = item / chunks, ch = item - e * chunks;
        const int64_t colbase = (int64_t)e * d.ldb + (int64_t)ch * 32;
        const bool live = (int64_t)ch * 32 + j < d.n_scenarios;
        const int jl = live ? j : 0;
        const int64_t hbase = nat ? ((int64_t)e * (d.ldb / 32) + ch) * 1024 : colbase;
        const __amdgpu_buffer_rsrc_t rY = rsrc_of(Yo + colbase), rG = rsrc_of(dY + colbase), rH2 = rsrc_of(H2 + hbase),
                                     rH1 = rsrc_of(H1 + hbase), rX = rsrc_of(GATHER ? H1 : Xh + colbase);
        // column-owner offsets: row crow(r, h) = (r & 3) + 8 (r >> 2) [scalar] + 4 h [lane]
        const int vo_h = (4 * h) * hs4_ + jl * 4, vo_e = (4 * h) * el4_ + jl * 4;
        const int vo_row = j * hs4_ + h * 64;                 // row-owner: row j, columns h*16 ..
        // (see mlp3_bwd_fused_kernel: an offset the compiler cannot see through keeps the weight fragments out of registers)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* W1 = sW1 + opaque;
        const float* W2 = sW2 + opaque;
        const float* W3 = sW3 + opaque;
        // (the scalar row offsets r * stride are loop invariants too: hoisted, there were ~100 of them, spilled to VGPR lanes and
        // read back with v_readlane + s_nop inside the item loop; recomputed per item they are one s_mul each on the idle SALU)
        const int hs4 = hs4_ + opaque, el4 = el4_ + opaque;
        const int n_out = n_out_ + opaque, Kq = K + opaque;   // (and the lane masks built from them: 64 SGPR pairs)
        // Software-pipelined by hand: each phase issues the loads of the NEXT phase before its own MFMA loop, and scheduling
        // barriers keep hipcc from hoisting every load of the item to the top.
        float gy[16], yo[16], h2[16], h1[16], dz[16], at[16], bts[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row_s = (r & 3) + 8 * (r >> 2);
            // dY / Y have n_out rows: lanes whose row does not exist read row 0 of their column and are masked below
            const int vo_y = (row_s + 4 * h < n_out) ? vo_e + row_s * el4 : jl * 4;
            gy[r] = ld1(rG, vo_y, 0);
            yo[r] = ld1(rY, vo_y, 0);
            h2[r] = ld1(rH2, vo_h, row_s * hs4);
        }
        row_owner(rH2, vo_row, 0, bts[0]);
        if (out_act == NIC_MLP3_ACT_ELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = (crow(r, h) < n_out && live) ? gy[r] * nic::elu1_grad_from_out(yo[r]) : 0.f;
        } else if (out_act == NIC_MLP3_ACT_SOFTPLUS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = (crow(r, h) < n_out && live) ? gy[r] * (1.f - expf(-yo[r])) : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = (crow(r, h) < n_out && live) ? gy[r] : 0.f;
        }
        transpose(dz, at);
        __builtin_amdgcn_sched_barrier(0);
        // layer 3: dW3 += dz3 H2^T, dH2 = W3^T dz3; layer 2's operands on their way
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[r] = ld1(rH1, vo_h, ((r & 3) + 8 * (r >> 2)) * hs4);
        row_owner(rH1, vo_row, 0, bts[1]);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                g3 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bts[0][s], g3, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W3[crow(s, h) * 32 + i], dz[s], acc, 0, 0, 0);
                sb += at[s];
            }
            gb3 += sb + __shfl_xor(sb, 32);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = acc[r] * nic::elu1_grad_from_out(h2[r]);
        transpose(dz, at);
        __builtin_amdgcn_sched_barrier(0);
        // layer 2; the first block of X on its way (rows >= K of the last block: past the descriptor's reach -> zeros below)
        if (j >= K) {
#pragma unroll
            for (int s = 0; s < 16; ++s) bts[0][s] = 0.f;
        } else if (!GATHER) {
            row_owner(rX, vo_row, 0, bts[0]);
        } else {
            gather_row_owner(j, e, ch, bts[0]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                g2 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bts[1][s], g2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W2[crow(s, h) * 32 + i], dz[s], acc, 0, 0, 0);
                sb += at[s];
            }
            gb2 += sb + __shfl_xor(sb, 32);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = acc[r] * nic::elu1_grad_from_out(h1[r]);
        transpose(dz, at);
        {
            float sb = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sb += at[s];
            gb1 += sb + __shfl_xor(sb, 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        // layer 1: dW1 block g += dz1 X[32g .. 32g+31]^T, dX block g = W1[:, 32g ..]^T dz1; block g + 1 of X on its way
        const __amdgpu_buffer_rsrc_t rD = rsrc_of(dX ? dX + colbase : H1);
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (g + 1 < KG) {
                if (32 * (g + 1) + j >= K) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) bts[(g + 1) & 1][s] = 0.f;
                } else if (!GATHER) {
                    row_owner(rX, vo_row, 32 * (g + 1) * hs4, bts[(g + 1) & 1]);
                } else {
                    gather_row_owner(32 * (g + 1) + j, e, ch, bts[(g + 1) & 1]);
                }
            }
            if (dX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    g1[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bts[g & 1][s], g1[g], 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[crow(s, h) * 32 * KG + 32 * g + i], dz[s], acc, 0, 0, 0);
                }
                if (live && (K == 32 * KG || g + 1 < KG)) {   // a full block: no per-row guard (an exec-mask branch per store)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k_s = 32 * g + (r & 3) + 8 * (r >> 2);
                        const float v = acc[r];   // (bit_cast straight from the vector element stored element 0 sixteen times)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rD, vo_e, k_s * el4, 0);
                    }
                } else if (live) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k_s = 32 * g + (r & 3) + 8 * (r >> 2);
                        const float v = acc[r];
                        if (k_s + 4 * h < Kq) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rD, vo_e, k_s * el4, 0);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) g1[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], bts[g & 1][s], g1[g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAIL > 0) {   // rows 32 KG .. 32 KG + TAIL - 1 on the vector unit
#pragma unroll
            for (int tt = 0; tt < TAIL; ++tt) {
                const int k = 32 * KG + tt;
                float xt[16];   // the row's columns h*16 .. h*16+15 (the same for every lane of a half)
                if (!GATHER) row_owner(rX, h * 64, k * hs4, xt);
                else gather_row_owner(k, e, ch, xt);
                float part = 0.f, gx = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) part = fmaf(at[s2], xt[s2], part);
                gtail[tt] += part;
#pragma unroll
                for (int r = 0; r < 16; ++r) gx = fmaf(sWt[crow(r, h) * TAIL + tt], dz[r], gx);
                gx += __shfl_xor(gx, 32);
                if (dX && live && h == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gx), rD, jl * 4, k * el4, 0);
            }
        }
    }
    // the four wavefronts add their accumulators in wave order through LDS (over the weight staging area), then the workgroup
    // adds the sum to its slab slot
    __syncthreads();
    float* const r1 = sW;
    float* const r2 = sW + 32 * 32 * KG;
    float* const r3 = r2 + 32 * 32;
#pragma unroll 1
    for (int turn = 0; turn < kHistWaves; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = crow(r, h);
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    float* q = r1 + n * 32 * KG + 32 * g + j;
                    *q = turn ? *q + g1[g][r] : g1[g][r];
                }
                float* q2 = r2 + n * 32 + j;
                *q2 = turn ? *q2 + g2[r] : g2[r];
                float* q3 = r3 + n * 32 + j;
                *q3 = turn ? *q3 + g3[r] : g3[r];
            }
            if (h == 0) {
                sbias[i] = turn ? sbias[i] + gb1 : gb1;
                sbias[32 + i] = turn ? sbias[32 + i] + gb2 : gb2;
                sbias[64 + i] = turn ? sbias[64 + i] + gb3 : gb3;
            }
            if (TAIL > 0) {
#pragma unroll
                for (int tt = 0; tt < TAIL; ++tt) {
                    const float tot = gtail[tt] + __shfl_xor(gtail[tt], 32);   // both column halves of row i
                    if (h == 0) stail[tt * 32 + i] = turn ? stail[tt * 32 + i] + tot : tot;
                }
            }
        }
        __syncthreads();
    }
    float* s1 = slab1 + (int64_t)blockIdx.x * 32 * lds1;
    float* s2 = slab2 + (int64_t)blockIdx.x * 32 * lds2;
    float* s3 = slab3 + (int64_t)blockIdx.x * d.n_out * lds3;
    {   // slab slot += block sum: the slot's old values are all fetched before the first add / store
        constexpr int NT = 64 * kHistWaves, IT1 = 32 * 32 * KG / NT, IT2 = 32 * 32 / NT;
        float o1[IT1], o2[IT2], o3[IT2];
#pragma unroll
        for (int u = 0; u < IT1; ++u) {
            const int idx = threadIdx.x + u * NT, n = idx / (32 * KG), k = idx % (32 * KG);
            o1[u] = s1[(int64_t)n * lds1 + (k < K ? k : 0)];
        }
#pragma unroll
        for (int u = 0; u < IT2; ++u) {
            const int idx = threadIdx.x + u * NT, n = idx >> 5, k = idx & 31;
            o2[u] = s2[(int64_t)n * lds2 + k];
            o3[u] = s3[(int64_t)(n < d.n_out ? n : 0) * lds3 + k];
        }
#pragma unroll
        for (int u = 0; u < IT1; ++u) {
            const int idx = threadIdx.x + u * NT, n = idx / (32 * KG), k = idx % (32 * KG);
            if (k < K) s1[(int64_t)n * lds1 + k] = o1[u] + r1[idx];
        }
#pragma unroll
        for (int u = 0; u < IT2; ++u) {
            const int idx = threadIdx.x + u * NT, n = idx >> 5, k = idx & 31;
            s2[(int64_t)n * lds2 + k] = o2[u] + r2[idx];
            if (n < d.n_out) s3[(int64_t)n * lds3 + k] = o3[u] + r3[idx];
        }
    }
    if (threadIdx.x < 32) {
        s1[(int64_t)threadIdx.x * lds1 + K] += sbias[threadIdx.x];
        if (TAIL > 0) {
#pragma unroll
            for (int tt = 0; tt < TAIL; ++tt) s1[(int64_t)threadIdx.x * lds1 + 32 * KG + tt] += stail[tt * 32 + threadIdx.x];
        }
        s2[(int64_t)threadIdx.x * lds2 + 32] += sbias[32 + threadIdx.x];
        if ((int)threadIdx.x < d.n_out) s3[(int64_t)threadIdx.x * lds3 + 32] += sbias[64 + threadIdx.x];
    }
}

// One thread = V consecutive scenarios of one (row, destination) pair; V = 4 (16-byte accesses) when the rows allow it.
template <int V>
__global__ void segment_sum_kernel(float* __restrict__ dst, int64_t dst_rs, const float* __restrict__ src, int64_t src_rs,
                                   const int32_t* __restrict__ offsets, const int32_t* __restrict__ items,
                                   const float* __restrict__ dst_scale, int R, int B, int64_t ldb, int accumulate) {
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int n = blockIdx.y;
    if (b >= B) return;
    const int lo = offsets[n], hi = offsets[n + 1];
    const float sc = dst_scale ? dst_scale[n] : 1.f;
    for (int r = blockIdx.z; r < R; r += gridDim.z) {
        float s[V];
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = 0.f;
        // in item order (deterministic; the association upstream's Python loop has); the loads of up to four items are issued
        // together (one dependent round trip per item otherwise: 17 for the warehouse node)
        for (int p0 = lo; p0 < hi; p0 += 4) {
            float x[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u < hi ? p0 + u : p0;
                const float* q = src + (int64_t)r * src_rs + (int64_t)items[p] * ldb + b;
                if (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(q);
                    x[u][0] = t.x; x[u][1 % V] = t.y; x[u][2 % V] = t.z; x[u][3 % V] = t.w;
                } else {
                    x[u][0] = q[0];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u < hi) {
#pragma unroll
                    for (int v = 0; v < V; ++v) s[v] += x[u][v];
                }
            }
        }
        float* o = dst + (int64_t)r * dst_rs + (int64_t)n * ldb + b;
        if (V == 4) {
            float4 y = accumulate ? *reinterpret_cast<const float4*>(o) : make_float4(0.f, 0.f, 0.f, 0.f);
            y.x = accumulate ? y.x + sc * s[0] : sc * s[0];
            y.y = accumulate ? y.y + sc * s[1] : sc * s[1];
            y.z = accumulate ? y.z + sc * s[2] : sc * s[2];
            y.w = accumulate ? y.w + sc * s[3] : sc * s[3];
            *reinterpret_cast<float4*>(o) = y;
        } else {
            *o = accumulate ? *o + sc * s[0] : sc * s[0];
        }
    }
}

// Several segment sums (and plain addends: offsets == NULL takes row n itself) into ONE destination in one launch:
// dst = (accumulate ? dst : 0) + scale_0 * S_0 + scale_1 * S_1 + ..., added in term order - exactly the values a chain of
// nic_segment_sum(..., accumulate) launches and tensor adds produces, without the intermediate round trips through HBM.
struct SegTerms {
    NicSegTerm t[NIC_SEG_MAX_TERMS];
    int n;
};
template <int V>
__global__ void segment_sum_terms_kernel(float* __restrict__ dst, int64_t dst_rs, SegTerms T, int R, int B, int64_t ldb,
                                         int accumulate) {
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int n = blockIdx.y;
    if (b >= B) return;
    for (int r = blockIdx.z; r < R; r += gridDim.z) {
        float* o = dst + (int64_t)r * dst_rs + (int64_t)n * ldb + b;
        float y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = 0.f;
        if (accumulate) {
#pragma unroll
            for (int v = 0; v < V; ++v) y[v] = o[v];
        }
#pragma unroll
        for (int j = 0; j < NIC_SEG_MAX_TERMS; ++j) {
            if (j < T.n) {
                const NicSegTerm& t = T.t[j];
                const float* base = t.src + (int64_t)r * t.src_row_stride + b;
                const float sc = t.scale ? t.scale[n] : 1.f;
                float s[V];
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] = 0.f;
                if (t.offsets == nullptr) {
#pragma unroll
                    for (int v = 0; v < V; ++v) s[v] = base[(int64_t)n * ldb + v];
                } else {
                    const int lo = t.offsets[n], hi = t.offsets[n + 1];
                    for (int p0 = lo; p0 < hi; p0 += 4) {
                        float x[4][V];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int p = p0 + u < hi ? p0 + u : p0;
                            const float* q = base + (int64_t)t.items[p] * ldb;
#pragma unroll
                            for (int v = 0; v < V; ++v) x[u][v] = q[v];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (p0 + u < hi) {
#pragma unroll
                                for (int v = 0; v < V; ++v) s[v] += x[u][v];
                            }
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < V; ++v) y[v] = (j == 0 && !accumulate) ? sc * s[v] : y[v] + sc * s[v];
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = y[v];
    }
}

// ---- proportional allocation of the warehouse's on-hand stock over its outgoing edges (+ self loop) ------------------------------
// (neural_networks.py:111-138 via :1435-1492).  One lane = one scenario; the ~25 small tensor ops per period this replaces were
// 10 % of the GNN step.  out [E][ldb] = desired quantity per edge; members = internal edges 0..S-1 (+ e_self if >= 0);
// orders [S+1][ldb] = out[s] * scale for the stores, out[e_supplier] for the warehouse's own order.
constexpr int kAllocThreads = 64;   // (64-thread workgroups: 8,192 scenarios spread over 128 CUs instead of 32)
using nic::kAllocBatch;
// (the one-warehouse bodies live in gnn_alloc_body.h: csrc/gnn_alloc_env.hip runs them in one launch with the env step)
__global__ void gnn_alloc_fwd_kernel(const float* __restrict__ out, const float* __restrict__ on_hand, float* __restrict__ orders,
                                     float* __restrict__ sums, float* __restrict__ ratio, float* __restrict__ scale, int S,
                                     int e_self, int e_sup, int cap_at_one, int B, int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    nic::gnn_alloc_fwd_one(out, on_hand, orders, sums, ratio, scale, S, e_self, e_sup, cap_at_one, b, ldb);
}

__global__ void gnn_alloc_bwd_kernel(const float* __restrict__ out, const float* __restrict__ on_hand,
                                     const float* __restrict__ g_orders, const float* __restrict__ sums,
                                     const float* __restrict__ ratio, const float* __restrict__ scale, float* __restrict__ d_out,
                                     float* __restrict__ g_on_hand, int S, int E, int e_self, int e_sup, int cap_at_one, int B,
                                     int64_t ldb) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    nic::gnn_alloc_bwd_one(out, on_hand, g_orders, sums, ratio, scale, d_out, g_on_hand, S, E, e_self, e_sup, cap_at_one, b, ldb);
}

// ---- the same allocation for SEVERAL supplying nodes (many-warehouse graphs) ----------------------------------------------------
// groups [G][4] = {first member edge, member count, self-loop edge or -1, supplier edge}: the internal edges of one warehouse are
// contiguous (adjacency.nonzero() is warehouse-major); order_row [n_edges] = row of the orders buffer an edge's quantity goes
// to (-1: none).  One lane = (scenario, group): blockIdx.y = group.  Sums run in edge order like the one-warehouse kernel.
__global__ void gnn_alloc_groups_fwd_kernel(const float* __restrict__ out, const float* __restrict__ on_hand, int64_t oh_stride,
                                            float* __restrict__ orders, float* __restrict__ sums, float* __restrict__ ratio,
                                            float* __restrict__ scale, const int32_t* __restrict__ groups,
                                            const int32_t* __restrict__ order_row, int cap_at_one, int B, int64_t ldb) {
#pragma clang fp contract(off)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int g = blockIdx.y;
    const int first = groups[4 * g], count = groups[4 * g + 1], e_self = groups[4 * g + 2], e_sup = groups[4 * g + 3];
    const float sup_v = out[(int64_t)e_sup * ldb + b];
    orders[(int64_t)order_row[e_sup] * ldb + b] = sup_v;
    if (count == 0 && e_self < 0) return;   // a warehouse that supplies nobody: its own order only
    const float oh = on_hand[(int64_t)g * oh_stride + b];
    float sum = 0.f;
    for (int i = 0; i < count; ++i) sum += out[(int64_t)(first + i) * ldb + b];
    if (e_self >= 0) sum += out[(int64_t)e_self * ldb + b];
    const float r = oh / (sum + 1e-10f);
    const float sc = cap_at_one ? fminf(r, 1.f) : r;
    sums[(int64_t)g * ldb + b] = sum;
    ratio[(int64_t)g * ldb + b] = r;
    scale[(int64_t)g * ldb + b] = sc;
    for (int i = 0; i < count; ++i) orders[(int64_t)order_row[first + i] * ldb + b] = out[(int64_t)(first + i) * ldb + b] * sc;
}

// adjoint; rows [zero_first, zero_first + zero_count) of d_out (the demand edges: they feed nothing) are cleared by group 0
__global__ void gnn_alloc_groups_bwd_kernel(const float* __restrict__ out, const float* __restrict__ on_hand, int64_t oh_stride,
                                            const float* __restrict__ g_orders, const float* __restrict__ sums,
                                            const float* __restrict__ ratio, const float* __restrict__ scale,
                                            float* __restrict__ d_out, float* __restrict__ g_on_hand,
                                            const int32_t* __restrict__ groups, const int32_t* __restrict__ order_row,
                                            int zero_first, int zero_count, int cap_at_one, int B, int64_t ldb) {
#pragma clang fp contract(off)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int g = blockIdx.y;
    const int first = groups[4 * g], count = groups[4 * g + 1], e_self = groups[4 * g + 2], e_sup = groups[4 * g + 3];
    if (g == 0)
        for (int i = 0; i < zero_count; ++i) d_out[(int64_t)(zero_first + i) * ldb + b] = 0.f;
    d_out[(int64_t)e_sup * ldb + b] = g_orders[(int64_t)order_row[e_sup] * ldb + b];
    if (count == 0 && e_self < 0) return;
    const int64_t gb = (int64_t)g * ldb + b;
    const float rt = ratio[gb], sm = sums[gb], sc = scale[gb], oh = on_hand[(int64_t)g * oh_stride + b];
    float dot = 0.f;
    for (int i = 0; i < count; ++i)
        dot += g_orders[(int64_t)order_row[first + i] * ldb + b] * out[(int64_t)(first + i) * ldb + b];
    const float passes = cap_at_one ? (rt <= 1.f ? 1.f : 0.f) : 1.f;
    const float d_scale = dot * passes;
    const float den = sm + 1e-10f;
    const float common = -(d_scale * oh / (den * den));
    for (int i = 0; i < count; ++i)
        d_out[(int64_t)(first + i) * ldb + b] = common + g_orders[(int64_t)order_row[first + i] * ldb + b] * sc;
    if (e_self >= 0) d_out[(int64_t)e_self * ldb + b] = common;
    g_on_hand[(int64_t)g * oh_stride + b] += d_scale / den;
}

int validate(const NicMlp3Desc* d, const char* who) {
    NIC_REQUIRE(d && d->weights, "%s: null descriptor / weights", who);
    NIC_REQUIRE(d->n_entities > 0 && d->n_scenarios > 0 && d->ldb >= d->n_scenarios && d->ldb % 32 == 0,
                "%s: bad sizes (ldb must be a multiple of 32 and >= n_scenarios)", who);
    NIC_REQUIRE(d->ent_row_stride == 0 || d->ent_row_stride >= (int64_t)d->n_entities * d->ldb,
                "%s: ent_row_stride shorter than n_entities * ldb", who);
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

int nic_mlp3_fwd_residual(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, const float* residual, float* Ysum,
                          void* stream);
int nic_mlp3_fwd(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, void* stream) {
    return nic_mlp3_fwd_residual(d, Y, X_hist, H1, H2, nullptr, nullptr, stream);
}

int nic_mlp3_fwd_residual(const NicMlp3Desc* d, float* Y, float* X_hist, float* H1, float* H2, const float* residual, float* Ysum,
                          void* stream) {
    if (int e = validate(d, "nic_mlp3_fwd")) return e;
    NIC_REQUIRE((residual == nullptr) == (Ysum == nullptr), "nic_mlp3_fwd_residual: residual and Ysum go together");
    NIC_REQUIRE(Y, "nic_mlp3_fwd: null output");
    NIC_REQUIRE((!X_hist && !H1 && !H2) || (H1 && H2), "nic_mlp3_fwd: incomplete history buffers (H1 and H2 go together)");
    NIC_REQUIRE(!(d->hist_native && X_hist), "nic_mlp3_fwd: the native history layout keeps H1 / H2 only (no X_hist)");
    NIC_REQUIRE(d->weights_t, "nic_mlp3_fwd: weights_t (the pre-transposed weight copy) is required");
    // Two adjacent chunks per wavefront when one chunk per wavefront is about one round of the chip's wavefront slots or less
    // (256 CUs x 4 SIMDs x 4-5 wavefronts of ~100 VGPRs): then the launch lasts as long as one wavefront's dependency chain, and
    // a wavefront that keeps twice the bytes in flight shortens it (node update, 17 nodes x 256 chunks: 52.6 -> 46-50 us).
    // Launches with clearly more wavefronts than slots lose more from the lower occupancy of the 164-VGPR variant than they gain
    // (edge MLPs, 34 x 256 chunks: +3 ... +17 %).
    const int64_t waves1 = (int64_t)nic::ceil_div(d->n_scenarios, 32) * d->n_entities;
    const int ch = (waves1 <= kFwdSlots && waves1 >= kFwdPairMinWaves) ? 2 : 1;
    const dim3 grid(nic::ceil_div(d->n_scenarios, 32 * ch * kFwdWaves), d->n_entities), block(64 * kFwdWaves);
    hipStream_t s = nic::as_stream(stream);
    const int ks = (d->K + 1) / 2;
    const int t = ks <= 4 ? 4 : (ks <= 16 ? 16 : (ks <= 33 ? 33 : 48));
    // 32-bit row offsets whenever every buffer's rows span < 2 GiB (always at the sizes this engine allocates)
    const int64_t hs = d->hist_row_stride ? d->hist_row_stride : (d->ent_row_stride ? d->ent_row_stride : (int64_t)d->n_entities * d->ldb);
    const int64_t top = 2 * (int64_t)((d->K + 1) / 2) + 32;
    const bool fits = top * hs * 4 < (1ll << 31) && top * (d->ent_row_stride ? d->ent_row_stride : (int64_t)d->n_entities * d->ldb) * 4 < (1ll << 31);
    const int addr = !fits ? kAddrFlat : (X_hist ? kAddrBufX : kAddrBuf);
    nic::note_kernelf("mlp3_fwd_kernel<%d,%d,%d>", t, ch, addr);
#define NIC_MLP3_FWD2(KS, CH_)                                                                                                   \
    do {                                                                                                                         \
        if (addr == kAddrBuf) hipLaunchKernelGGL((mlp3_fwd_kernel<KS, CH_, kAddrBuf>), grid, block, 0, s, *d, d->weights_t, Y, X_hist, H1, H2, residual, Ysum); \
        else if (addr == kAddrBufX) hipLaunchKernelGGL((mlp3_fwd_kernel<KS, CH_, kAddrBufX>), grid, block, 0, s, *d, d->weights_t, Y, X_hist, H1, H2, residual, Ysum); \
        else hipLaunchKernelGGL((mlp3_fwd_kernel<KS, CH_, kAddrFlat>), grid, block, 0, s, *d, d->weights_t, Y, X_hist, H1, H2, residual, Ysum); \
    } while (0)
#define NIC_MLP3_FWD(KS)                                                                                                         \
    do {                                                                                                                         \
        if (ch == 2) NIC_MLP3_FWD2(KS, 2);                                                                                       \
        else NIC_MLP3_FWD2(KS, 1);                                                                                               \
    } while (0)
    if (t == 4) NIC_MLP3_FWD(4);
    else if (t == 16) NIC_MLP3_FWD(16);
    else if (t == 33) NIC_MLP3_FWD(33);
    else NIC_MLP3_FWD(48);
#undef NIC_MLP3_FWD2
#undef NIC_MLP3_FWD
    return nic::check_launch("nic_mlp3_fwd");
}

int nic_mlp3_bwd_fused_slots(void) { return kFusedBlocks * kFusedWaves; }

int nic_mlp3_bwd_fused(const NicMlp3Desc* d, const float* dY, const float* Y, float* dX, float* slab1, int64_t lds1, float* slab2,
                       int64_t lds2, float* slab3, int64_t lds3, void* stream) {
    if (int e = validate(d, "nic_mlp3_bwd_fused")) return e;
    NIC_REQUIRE(dY && Y && slab1 && slab2 && slab3, "nic_mlp3_bwd_fused: null buffer");
    NIC_REQUIRE(d->weights_t, "nic_mlp3_bwd_fused: weights_t (the pre-transposed weight copy) is required");
    NIC_REQUIRE(lds1 >= d->K + 1 && lds2 >= 33 && lds3 >= 33, "nic_mlp3_bwd_fused: slab rows too short");
    const dim3 grid(kFusedBlocks), block(64 * kFusedWaves);
    hipStream_t s = nic::as_stream(stream);
    const int kg = (d->K + 31) / 32;
    nic::note_kernelf("mlp3_bwd_fused_kernel<%d>", kg);
#define NIC_MLP3_BF(KG)                                                                                                      \
    hipLaunchKernelGGL(mlp3_bwd_fused_kernel<KG>, grid, block, 0, s, *d, d->weights_t, d->weights, dY, Y, dX, slab1, lds1, slab2, \
                       lds2, slab3, lds3)
    if (kg == 1) NIC_MLP3_BF(1);
    else if (kg == 2) NIC_MLP3_BF(2);
    else NIC_MLP3_BF(3);
#undef NIC_MLP3_BF
    return nic::check_launch("nic_mlp3_bwd_fused");
}

int nic_mlp3_bwd_hist_slots(void) { return kHistBlocks; }

int nic_mlp3_bwd_hist(const NicMlp3Desc* d, const float* dY, const float* Y, const float* X_hist, const float* H1, const float* H2,
                      float* dX, float* slab1, int64_t lds1, float* slab2, int64_t lds2, float* slab3, int64_t lds3, void* stream) {
    if (int e = validate(d, "nic_mlp3_bwd_hist")) return e;
    NIC_REQUIRE(dY && Y && H1 && H2 && slab1 && slab2 && slab3, "nic_mlp3_bwd_hist: null buffer");
    NIC_REQUIRE(!(d->hist_native && X_hist), "nic_mlp3_bwd_hist: the native history layout re-gathers the inputs (no X_hist)");
    NIC_REQUIRE(lds1 >= d->K + 1 && lds2 >= 33 && lds3 >= 33, "nic_mlp3_bwd_hist: slab rows too short");
    const int64_t hs = d->hist_row_stride ? d->hist_row_stride : (d->ent_row_stride ? d->ent_row_stride : (int64_t)d->n_entities * d->ldb);
    NIC_REQUIRE(hs % 4 == 0 && (reinterpret_cast<uintptr_t>(X_hist) & 15) == 0 && (reinterpret_cast<uintptr_t>(H1) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(H2) & 15) == 0,
                "nic_mlp3_bwd_hist: history rows must be 16-byte aligned");
    NIC_REQUIRE((int64_t)(d->K + 31) / 32 * 32 * hs * 4 < (1ll << 31) && (int64_t)(d->K + 32) * d->n_entities * d->ldb * 4 < (1ll << 31),
                "nic_mlp3_bwd_hist: a buffer's rows must span less than 2 GiB (32-bit row offsets)");
    const dim3 grid(kHistBlocks), block(64 * kHistWaves);
    hipStream_t s = nic::as_stream(stream);
    // K = 32 kg + tail: up to four rows past the last full block are handled on the vector unit instead of a 32-row MFMA block
    int kg = (d->K + 31) / 32, tail = 0;
    if (d->K >= 32 && d->K % 32 >= 1 && d->K % 32 <= 1) kg = d->K / 32, tail = d->K % 32;
    nic::note_kernelf("mlp3_bwd_hist_kernel<%d,%s%s>", kg, X_hist ? "stored" : "gather", tail ? ",tail" : "");
#define NIC_MLP3_BH(KG, G, TL)                                                                                                \
    hipLaunchKernelGGL((mlp3_bwd_hist_kernel<KG, G, TL>), grid, block, 0, s, *d, d->weights, dY, Y, X_hist, H1, H2, dX, slab1,     \
                       lds1, slab2, lds2, slab3, lds3)
    if (X_hist) {
        if (tail == 1 && kg == 1) NIC_MLP3_BH(1, false, 1);
        else if (tail == 1 && kg == 2) NIC_MLP3_BH(2, false, 1);
        else if (kg == 1) NIC_MLP3_BH(1, false, 0);
        else if (kg == 2) NIC_MLP3_BH(2, false, 0);
        else NIC_MLP3_BH(3, false, 0);
    } else {
        if (tail == 1 && kg == 1) NIC_MLP3_BH(1, true, 1);
        else if (tail == 1 && kg == 2) NIC_MLP3_BH(2, true, 1);
        else if (kg == 1) NIC_MLP3_BH(1, true, 0);
        else if (kg == 2) NIC_MLP3_BH(2, true, 0);
        else NIC_MLP3_BH(3, true, 0);
    }
#undef NIC_MLP3_BH
    return nic::check_launch("nic_mlp3_bwd_hist");
}

int nic_segment_sum_terms(float* dst, int64_t dst_row_stride, const NicSegTerm* terms, int32_t n_terms, int32_t R, int32_t n_dst,
                          int32_t n_scenarios, int32_t ldb, int32_t accumulate, void* stream) {
    NIC_REQUIRE(dst && terms && n_terms >= 1 && n_terms <= NIC_SEG_MAX_TERMS, "nic_segment_sum_terms: 1..%d terms", NIC_SEG_MAX_TERMS);
    NIC_REQUIRE(R > 0 && n_dst > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_segment_sum_terms: bad sizes");
    SegTerms T;
    T.n = n_terms;
    bool vec = ldb % 4 == 0 && dst_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (n_scenarios + 3) / 4 * 4 <= ldb;
    for (int j = 0; j < NIC_SEG_MAX_TERMS; ++j) {
        T.t[j] = terms[j < n_terms ? j : 0];
        if (j < n_terms) {
            NIC_REQUIRE(terms[j].src && (terms[j].offsets == nullptr || terms[j].items), "nic_segment_sum_terms: null term buffer");
            vec = vec && terms[j].src_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(terms[j].src) & 15) == 0;
        }
    }
    if (vec) {
        const dim3 grid(nic::ceil_div(nic::ceil_div(n_scenarios, 4), 256), n_dst, R < 32 ? R : 32), block(256);
        nic::note_kernel("segment_sum_terms_kernel<4>");
        hipLaunchKernelGGL(segment_sum_terms_kernel<4>, grid, block, 0, nic::as_stream(stream), dst, dst_row_stride, T, R, n_scenarios,
                           (int64_t)ldb, accumulate);
    } else {
        const dim3 grid(nic::ceil_div(n_scenarios, 256), n_dst, R < 32 ? R : 32), block(256);
        nic::note_kernel("segment_sum_terms_kernel<1>");
        hipLaunchKernelGGL(segment_sum_terms_kernel<1>, grid, block, 0, nic::as_stream(stream), dst, dst_row_stride, T, R, n_scenarios,
                           (int64_t)ldb, accumulate);
    }
    return nic::check_launch("nic_segment_sum_terms");
}

int nic_gnn_alloc_fwd(const float* out, const float* on_hand, float* orders, float* sums, float* ratio, float* scale, int32_t S,
                      int32_t e_self, int32_t e_supplier, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(out && on_hand && orders && sums && ratio && scale, "nic_gnn_alloc_fwd: null buffer");
    NIC_REQUIRE(S > 0 && e_supplier >= 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_gnn_alloc_fwd: bad sizes");
    nic::note_kernel("gnn_alloc_fwd_kernel");
    hipLaunchKernelGGL(gnn_alloc_fwd_kernel, dim3(nic::ceil_div(n_scenarios, kAllocThreads)), dim3(kAllocThreads), 0, nic::as_stream(stream), out, on_hand,
                       orders, sums, ratio, scale, S, e_self, e_supplier, cap_at_one, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_gnn_alloc_fwd");
}

int nic_gnn_alloc_bwd(const float* out, const float* on_hand, const float* g_orders, const float* sums, const float* ratio,
                      const float* scale, float* d_out, float* g_on_hand, int32_t S, int32_t n_edges, int32_t e_self,
                      int32_t e_supplier, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(out && on_hand && g_orders && sums && ratio && scale && d_out && g_on_hand, "nic_gnn_alloc_bwd: null buffer");
    NIC_REQUIRE(S > 0 && n_edges > S && e_supplier >= 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_gnn_alloc_bwd: bad sizes");
    nic::note_kernel("gnn_alloc_bwd_kernel");
    hipLaunchKernelGGL(gnn_alloc_bwd_kernel, dim3(nic::ceil_div(n_scenarios, kAllocThreads)), dim3(kAllocThreads), 0, nic::as_stream(stream), out, on_hand,
                       g_orders, sums, ratio, scale, d_out, g_on_hand, S, n_edges, e_self, e_supplier, cap_at_one, n_scenarios,
                       (int64_t)ldb);
    return nic::check_launch("nic_gnn_alloc_bwd");
}

int nic_gnn_alloc_groups_fwd(const float* out, const float* on_hand, int64_t on_hand_group_stride, float* orders, float* sums,
                             float* ratio, float* scale, const int32_t* groups, const int32_t* order_row, int32_t n_groups,
                             int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(out && on_hand && orders && sums && ratio && scale && groups && order_row, "nic_gnn_alloc_groups_fwd: null buffer");
    NIC_REQUIRE(n_groups > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_gnn_alloc_groups_fwd: bad sizes");
    nic::note_kernel("gnn_alloc_groups_fwd_kernel");
    hipLaunchKernelGGL(gnn_alloc_groups_fwd_kernel, dim3(nic::ceil_div(n_scenarios, kAllocThreads), n_groups), dim3(kAllocThreads), 0,
                       nic::as_stream(stream), out, on_hand, on_hand_group_stride, orders, sums, ratio, scale, groups, order_row,
                       cap_at_one, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_gnn_alloc_groups_fwd");
}

int nic_gnn_alloc_groups_bwd(const float* out, const float* on_hand, int64_t on_hand_group_stride, const float* g_orders,
                             const float* sums, const float* ratio, const float* scale, float* d_out, float* g_on_hand,
                             const int32_t* groups, const int32_t* order_row, int32_t n_groups, int32_t zero_first,
                             int32_t zero_count, int32_t cap_at_one, int32_t n_scenarios, int32_t ldb, void* stream) {
    NIC_REQUIRE(out && on_hand && g_orders && sums && ratio && scale && d_out && g_on_hand && groups && order_row,
                "nic_gnn_alloc_groups_bwd: null buffer");
    NIC_REQUIRE(n_groups > 0 && zero_first >= 0 && zero_count >= 0 && n_scenarios > 0 && ldb >= n_scenarios,
                "nic_gnn_alloc_groups_bwd: bad sizes");
    nic::note_kernel("gnn_alloc_groups_bwd_kernel");
    hipLaunchKernelGGL(gnn_alloc_groups_bwd_kernel, dim3(nic::ceil_div(n_scenarios, kAllocThreads), n_groups), dim3(kAllocThreads), 0,
                       nic::as_stream(stream), out, on_hand, on_hand_group_stride, g_orders, sums, ratio, scale, d_out, g_on_hand,
                       groups, order_row, zero_first, zero_count, cap_at_one, n_scenarios, (int64_t)ldb);
    return nic::check_launch("nic_gnn_alloc_groups_bwd");
}

int nic_segment_sum(float* dst, int64_t dst_row_stride, const float* src, int64_t src_row_stride, const int32_t* offsets,
                    const int32_t* items, const float* dst_scale, int32_t R, int32_t n_dst, int32_t n_scenarios, int32_t ldb,
                    int32_t accumulate, void* stream) {
    NIC_REQUIRE(dst && src && offsets && items, "nic_segment_sum: null buffer");
    NIC_REQUIRE(R > 0 && n_dst > 0 && n_scenarios > 0 && ldb >= n_scenarios, "nic_segment_sum: bad sizes");
    // 16-byte path: rows 16-byte aligned and the scenario count padded to a multiple of 4 inside ldb (padding columns are
    // summed too: they hold zeros / are never read back)
    const bool vec = ldb % 4 == 0 && dst_row_stride % 4 == 0 && src_row_stride % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                     (n_scenarios + 3) / 4 * 4 <= ldb;
    if (vec) {
        const dim3 grid(nic::ceil_div(nic::ceil_div(n_scenarios, 4), 256), n_dst, R < 32 ? R : 32), block(256);
        nic::note_kernel("segment_sum_kernel<4>");
        hipLaunchKernelGGL(segment_sum_kernel<4>, grid, block, 0, nic::as_stream(stream), dst, dst_row_stride, src, src_row_stride,
                           offsets, items, dst_scale, R, n_scenarios, (int64_t)ldb, accumulate);
    } else {
        const dim3 grid(nic::ceil_div(n_scenarios, 256), n_dst, R < 32 ? R : 32), block(256);
        nic::note_kernel("segment_sum_kernel<1>");
        hipLaunchKernelGGL(segment_sum_kernel<1>, grid, block, 0, nic::as_stream(stream), dst, dst_row_stride, src, src_row_stride,
                           offsets, items, dst_scale, R, n_scenarios, (int64_t)ldb, accumulate);
    }
    return nic::check_launch("nic_segment_sum");
}
}
