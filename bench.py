#!/usr/bin/env python
"""Benchmark of the hot path: one TRAINING STEP of the differentiable inventory rollout
(reset + T x (policy forward + env step) + cost reduction + backward through the horizon [+ gradient all-reduce] + Adam)
on synthetic demand, BASELINE.json's metric: scenario-steps/s = scenarios x n_stores x T / wall-time.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3]

N = 1: this process.  N > 1 with no RANK in the environment: this process starts N fresh ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, one rank per GPU) BEFORE touching the GPU,
relays rank 0's JSON line and exits with the job's code; it refuses (non-zero exit, no JSON line) when fewer than N GPUs are
visible.  Under torchrun (RANK set) each rank runs its shard: scenarios are sharded (weak scaling: 65,536 scenarios PER GPU
for cfg3), one RCCL all-reduce of the flat gradient per step.  Rank 0 prints ONE JSON line.  Inputs (demand traces from the
HIP Philox sampler, initial state, weights) are resident in HBM before the timed region.

The JSON line carries `roofline` for the kernel class with the largest total time in the step (its name is what the C ABI
reported launching, `nic_last_kernel()`; `bound` is "mfma" for the 512-wide policy GEMMs and "hbm" for everything else,
with the algorithmic bytes / flops per launch defined in `algorithmic_work`), `kernels` with the same figures for every
class, `roofline_env_step` for the env-step kernel, and `cpu_baseline` (the oracle timed on this host's cores).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X dense FP32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0         # HBM3E (MI355X_MICROARCH.md)
MACHINE_BALANCE = MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)   # 19.7 flop per byte


# ---- N > 1 without a launcher: start the ranks ourselves ---------------------------------------------------------------
def spawn_ranks(n_gpus, argv):
    """Parent of a multi-GPU run.  Decided before any HIP call (device_count() does not initialise the GPU on this image);
    the parent never touches the GPU, only waits and relays."""
    visible = torch.cuda.device_count()
    if visible < n_gpus:
        print(f"bench.py: --gpus {n_gpus} requested but only {visible} GPU(s) visible; refusing to report a "
              f"{visible or 1}-rank number as a {n_gpus}-GPU result", file=sys.stderr)
        return 2
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout[-4000:])
        print(f"bench.py: the {n_gpus}-rank job failed (rc {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(lines[-1])
    return 0


_REAL_STDOUT = None


def _quiet_stdout():
    """From here on file descriptor 1 is stderr: RCCL writes version banners and warnings to stdout (at communicator set-up and
    tear-down, some without a trailing newline), and the contract is ONE JSON line there.  The line itself goes out through a
    duplicate of the original descriptor (`_emit`)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(out):
    f = _REAL_STDOUT or sys.stdout
    f.write(json.dumps(out) + "\n")
    f.flush()


def _lib_id():
    try:
        from neural_inventory_control_amd import _lib
        return (_lib.lib().nic_build_id() or b"").decode()
    except Exception:
        return None


def build_case(workload, device, rank, world, scenarios=None, periods=None, generic_route=False):
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from neural_inventory_control_amd.data_handling import Scenario
    from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
    from neural_inventory_control_amd.closed_form import ClosedFormRollout
    from neural_inventory_control_amd.gnn_rollout import GnnRollout
    from neural_inventory_control_amd.rollout import FusedRollout
    setting, policy, n, T, desc = workloads.get(workload)
    if scenarios or periods:
        desc += f" [overridden: {scenarios or n} scenarios/GPU x T={periods or T}]"
    n, T = scenarios or n, periods or T
    obs = defaultdict(lambda: None, setting["observation_params"])
    # each rank generates rows [rank*n, (rank+1)*n) of the (world*n)-scenario job; initial inventories use the GLOBAL demand mean
    real = setting["store_params"]["demand"]["distribution"] == "real"
    if real:   # file-backed demand: the train split of the period axis (past-demand window + T periods), reference-style
        shift = setting["observation_params"]["demand"]["period_shift"]
        sc = Scenario(shift + T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                      setting["echelon_params"], n, obs, setting["seeds"], device=device, scenario_offset=rank * n,
                      num_total=world * n)
    else:
        sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                      setting["echelon_params"], n, obs, setting["seeds"], sampler="hip", device=device,
                      scenario_offset=rank * n, num_total=world * n)
    if real:
        from neural_inventory_control_amd.data_handling import DatasetCreator
        data = {k: v.to(device) for k, v in DatasetCreator().split_by_period(sc, [f"(0, {shift + T})"])[0].items()}
    else:
        data = {k: v.to(device) for k, v in sc.get_data().items()}
    torch.manual_seed(1234)  # identical initial weights on every rank (and broadcast below when world > 1)
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=device)
    eng = None
    if generic_route:   # A/B: Simulator.step (one HIP kernel per period) + HipLinear layers + autograd, no fused engine
        pass
    elif FusedRollout.supports(model):
        eng = FusedRollout(model, setting["problem_params"], device)
    elif ClosedFormRollout.supports(model):
        eng = ClosedFormRollout(model, setting["problem_params"], device)
        if not eng.shapes_ok(data):
            eng = None
    elif GnnRollout.supports(model, setting["problem_params"]) and "mean" in data:
        eng = GnnRollout(model, setting["problem_params"], device)
    return setting, policy, sc, data, model, eng, n, T, desc


_TRAFFIC_WORKLOAD = None   # set by main(): counter passes of another workload's launches of the same kernel do not apply


def _pmc_traffic(kernel_name, n_scenarios, label=None):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*traffic*.json): FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc passes and corrected as the files' notes say.  None if no pass has this kernel
    at this scenario count.  Entries that carry a `label` (kernel class: two MLPs may share one template) must match it."""
    import glob
    # rocprofv3 prints template arguments as numbers, nic_last_kernel() with the enumerator names
    squash = lambda s: s.replace(" ", "").replace("EPI_BIAS_ACT", "0").replace("EPI_DGRAD", "1")  # noqa: E731
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            doc = json.load(open(f))
            if not isinstance(doc, dict) or doc.get("n_scenarios") != n_scenarios:
                continue
            if _TRAFFIC_WORKLOAD and doc.get("workload") and doc["workload"] != _TRAFFIC_WORKLOAD:
                continue
            for e in doc.get("kernels", []):
                if squash(e["kernel"]) == squash(kernel_name) and e.get("label", label) == label:
                    return {"bytes_per_launch": e["hbm_bytes_per_launch"], "source": os.path.basename(f),
                            "periods": doc.get("periods")}
        except Exception:
            pass
    return None


def _pick_threads(avail, probe):
    """Thread count for the CPU baseline, calibrated on what is timed: `probe()` is a short training step of the oracle
    (a slice of the sample, a few periods).  Eager PyTorch on this path collapses when oversubscribed (a 256-thread host
    ran it 777 x slower with every thread than with 32), so the candidates stop at 64 and the fastest one is used for the
    whole baseline and reported as `cores`; the per-candidate timings go into the bench line (`thread_probe`)."""
    best, best_t, seen = 1, None, {}
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        seen[n] = round(dt, 4)
        if best_t is None or dt < 0.97 * best_t:   # (ties go to the smaller count: less run-to-run variance)
            best, best_t = n, dt
    return best, seen


def _certified_ratio():
    """oracle / reference wall time on identical inputs, timed in the build container (tools/oracle_vs_reference_timing.py;
    the reference cannot travel to the GPU box): what relates `cpu_baseline` to the reference's own CPU speed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*oracle_vs_reference_timing.json")), reverse=True):
        try:
            d = json.load(open(f))
            return {"ratio": round(d["ratio_oracle_over_reference"], 3), "threads": d["threads"], "shape": d["shape"],
                    "source": os.path.basename(f)}
        except Exception:
            pass
    return None


def _full_batch_record(workload):
    """The committed one-off measurement of the CPU baseline at the GPU run's full batch (tools/cpu_baseline_full.py on the GPU
    box's host cores; profiles/*cpu_baseline_full.json), quoted beside the bounded sample a default run times."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_baseline_full.json")), reverse=True):
        try:
            rec = json.load(open(f))["workloads"].get(workload)
            if rec and rec.get("value"):
                return dict(rec, source=os.path.basename(f))
        except Exception:
            pass
    return None


def cpu_baseline(workload, sample_scenarios, periods, reps=3, model=None, setting_policy=None, strict_reps=False):
    """The oracle (CPU restatement of the reference path, PyTorch eager) timed on this host's cores: training steps
    (rollout + backward) of the same workload on a bounded sample of scenarios; 1 warm-up + `reps` timed repetitions,
    median."""
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from oracle import inventory_oracle as orc
    avail = os.cpu_count() or 1
    setting, policy = setting_policy if setting_policy is not None else workloads.get(workload)[:2]
    obs = defaultdict(lambda: None, setting["observation_params"])
    real = setting["store_params"]["demand"]["distribution"] == "real"
    shift = setting["observation_params"]["demand"]["period_shift"]
    data = orc.generate_scenario_data(periods + (shift if real else 0), setting["problem_params"], setting["store_params"],
                                      setting["warehouse_params"], setting["echelon_params"], sample_scenarios, obs,
                                      setting["seeds"])
    if real:
        data = orc.split_data_by_period(data, [f"(0, {shift + periods})"], obs, setting["problem_params"])[0]
    S = setting["problem_params"]["n_stores"]
    F = data["initial_inventories"].shape[1] * data["initial_inventories"].shape[2]
    if policy["name"] != "vanilla_one_store":
        F += sum(data[k].shape[1] * data[k].shape[2] for k in ("initial_warehouse_inventories", "initial_echelon_inventories")
                 if k in data)
    if policy["name"] in ("base_stock", "capped_base_stock", "echelon_stock"):
        F = 1  # closed-form policies: one Linear fed the constant 0 (neural_networks.py:228)
    quantile = policy["name"] in ("transformed_nv", "fixed_quantile", "quantile_nv", "returns_nv")
    if policy["name"] in ("gnn", "data_driven") or quantile:  # the oracle takes the (already materialised) weights of the device model
        state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        fstate = ({k: v.detach().cpu() for k, v in model.fixed_nets["quantile_forecaster"].state_dict().items()}
                  if quantile else None)   # (the frozen forecaster the policy inverts)
        pol = orc.policy_from_state_dict(policy, state, setting["problem_params"], forecaster_state=fstate)
    else:
        pol = orc.init_policy(policy, setting["problem_params"], F, 1234, setting["store_params"])
    warm = {k: v[:max(8, sample_scenarios // 16)] for k, v in data.items()}
    orc.train_step_gradients(pol, min(periods, 10), setting["problem_params"], warm, obs)  # warm-up
    # thread count calibrated on a short ROLLOUT of this workload (a quarter of the sample, 10 periods), fixed for the run
    part = {k: v[:max(8, sample_scenarios // 4)] for k, v in data.items()}
    cores, probe = _pick_threads(avail, lambda: orc.train_step_gradients(pol, min(periods, 10), setting["problem_params"],
                                                                         part, obs))
    torch.set_num_threads(cores)
    times = []
    while len(times) < reps or (sum(times) < 10.0 and len(times) < 25):  # >= 3 repetitions and ~10 s of CPU work
        t0 = time.perf_counter()
        orc.train_step_gradients(pol, periods, setting["problem_params"], data, obs)
        times.append(time.perf_counter() - t0)
        if strict_reps and len(times) >= reps:   # (the one-off full-batch record: exactly `reps` repetitions, however long)
            break
        if not strict_reps and sum(times) > 45.0:  # keep the default run within minutes on a slow host
            break
    dt = statistics.median(times)
    full = _full_batch_record(workload)
    return {"value": sample_scenarios * S * periods / dt, "unit": "scenario-steps/s", "cores": cores,
            **({"full_batch_value": full["value"], "full_batch": {k: full.get(k) for k in ("scenarios", "periods", "cores", "is_full_batch", "source")}}
               if full else {}),
            "thread_probe_s": probe, "oracle_over_reference_wall_time": _certified_ratio(),
            "host_cores": avail, "kind": "port",
            "sample": f"oracle (PyTorch-CPU eager restatement of the reference path), training step fwd+bwd on "
                      f"{sample_scenarios} scenarios x {S} stores x T={periods}: 1 warm-up + {len(times)} timed repetitions, "
                      f"median {dt:.3f} s (min {min(times):.3f}, max {max(times):.3f}), {cores} of {avail} host threads "
                      f"(fastest of 8/16/32/64 on a short rollout of the same workload; more threads than that oversubscribe "
                      f"this eager path)"}


def algorithmic_work(tag, kernel, shape):
    """(bound, amount per launch, unit) of one launch of a kernel class.  `shape`: n (scenarios), T, S, Wn, E, Ws, Ww, We, F (MLP
    input rows), nh (hidden layers of the small route), n_out, train.  DESIGN.md §4 states the same formulas."""
    n, T = shape["n"], shape["T"]
    kind, _, dims = tag.partition("_")
    if kind in ("fwd", "fwdT", "dgrad", "wgrad", "wgradT") and "x" in dims and dims.replace("x", "").isdigit():
        N, K = (int(v) for v in dims.split("x"))
        cols = n * (T if kind in ("wgradT", "fwdT") else 1)   # (..T: one launch over all (period x scenario) columns)
        if kernel.startswith("wgrad_small_kernel"):   # small route: one launch contracts over all T * ldb columns
            return "hbm", 4.0 * (N + K) * n * T, "B"  # operands read once (dZ [N] + X [K] rows per column)
        # A layer's GEMM streams its two activation operands once (N + K rows per column; the dgrad epilogue also reads the
        # stored activation: + K rows) and does 2 N K flop per column.  Which roofline binds is decided by the arithmetic
        # intensity against the machine balance (157.3 TFLOP/s / 8 TB/s = 19.7 flop/B), not by the layer's width: 17 x 512 is
        # 8 flop/B (HBM), 512 x 51 is 23 (just MFMA), 512 x 512 is 128 (MFMA).  Both figures are reported (`other`).
        flops = 2.0 * N * K * cols
        nbytes = 4.0 * (N + K + (K if kind == "dgrad" and min(N, K) > 64 else 0)) * cols
        if max(N, K) <= 64 or flops / nbytes < MACHINE_BALANCE:
            return "hbm", nbytes, "B", ("mfma", flops, "FLOP")
        return "mfma", flops, "FLOP", ("hbm", nbytes, "B")
    if tag.startswith("mlp3_fwd_") or tag.startswith("mlp3_bwd_"):  # fused 3-layer MLP: K -> 32 -> 32 -> n_out per column
        # HBM-bound: 0.4 flop per byte of activations.  Forward (training): K gathered input rows in, the two hidden
        # activations the backward needs and the output out.  Backward: dY / Y, the hidden activations and the inputs in,
        # the input gradient out (weight gradients stay in registers).  The optional X history is a design choice, not counted.
        # (a forward launch that folds a residual connection also writes the sum: `fold` rows)
        K, n_out, n_ent, fold = shape["gnn"][tag[len("mlp3_fwd_"):]]
        rows = (K + 64 + n_out + fold) if tag.startswith("mlp3_fwd_") else (2 * K + 64 + 2 * n_out)
        return "hbm", 4.0 * rows * n_ent * n, "B"
    S, Wn, E = shape["S"], shape["Wn"], shape["E"]
    f_state = S * shape["Ws"] + Wn * shape["Ww"] + E * shape["We"]
    n_ord = S * max(Wn, 1) + Wn + E
    if tag.startswith("bwd_thin_"):
        N, K = (int(v) for v in tag[len("bwd_thin_"):].split("x"))
        return "hbm", 4.0 * (2 * K + N) * n, "B"      # layer input read once, input gradient written once, dY read
    if tag == "env_fwd":  # state read + write, demand, orders, reward (static tables amortised over T)
        return "hbm", 4.0 * (2 * f_state + S + n_ord + 1) * n, "B"
    if tag == "env_bwd":  # state + orders + demand read, incoming state gradient read, state / order gradients written
        return "hbm", 4.0 * (3 * f_state + S + 2 * n_ord) * n, "B"
    if tag == "head_env_fwd":  # fused head + env step: state read + write, demand, logits in, orders written (the sweep reads them), reward
        return "hbm", 4.0 * (2 * f_state + S + shape["n_out"] + n_ord + 1) * n, "B"
    if tag == "head_env_bwd":  # state, orders, demand, logits, incoming state gradient in; state gradient, order gradients, dZ out
        return "hbm", 4.0 * (3 * f_state + S + 2 * n_ord + 2 * shape["n_out"]) * n, "B"
    if tag in ("wide_fwd", "wide_bwd") and shape.get("hidden"):
        # whole-horizon kernels of the wide policy (csrc/wide_rollout.hip): every layer of every period of every scenario on the
        # matrix cores (forward: first layer, hidden layers, logits; backward: their input gradients); the activation histories
        # (forward: hidden rows out; backward: hidden rows in, their gradients out) are the HBM side, reported as `other`
        hid, no = shape["hidden"], shape["n_out"]
        F = S * shape["Ws"] + Wn * shape["Ww"]
        per_col = 2.0 * ((F + 1) * hid[0] + sum(a_ * b_ for a_, b_ in zip(hid[:-1], hid[1:])) + no * hid[-1])
        hist = 4.0 * sum(hid) * (1 if tag == "wide_fwd" else 2)
        return "mfma", per_col * n * T, "FLOP", ("hbm", (hist + 4.0 * (2 * F + S + no + n_ord + 1)) * n * T, "B")
    if tag in ("tail_fwd", "tail_bwd") and shape.get("hidden"):
        # fused per-period tail (csrc/period_tail.hip).  Forward: last hidden activation in (K rows), next period's first hidden
        # activation out (N1 rows) + the fused head / env step's bytes.  Backward: the first layer's pre-activation gradient of the
        # next period in (N1), last hidden activation in and its gradient out (2 K) + the adjoints' bytes; the slab slot is
        # read-modify-written once per workgroup (not per column: not counted).  MFMA work (padded rows excluded) as `other`.
        K, N1, no, F = shape["hidden"][-1], shape["hidden"][0], shape["n_out"], f_state
        if tag == "tail_fwd":
            return ("hbm", 4.0 * (K + N1 + 2 * f_state + S + no + n_ord + 1) * n, "B",
                    ("mfma", 2.0 * (no * K + N1 * (F + 1)) * n, "FLOP"))
        return ("hbm", 4.0 * (N1 + 2 * K + 3 * f_state + S + n_ord + no) * n, "B",
                ("mfma", 2.0 * (F * N1 + 2 * no * K) * n, "FLOP"))
    if tag == "gnn_period_fwd" and shape.get("gnn"):
        # the GNN policy's period in one launch (csrc/gnn_period.hip): embeddings stay in LDS, so the HBM side is what the backward
        # will read - per evaluated (entity, scenario) column the two hidden activations + the output (+ the residual sum), the
        # aggregation rows (32 x 2 nodes), the node features' pipeline rows - plus the state in / out, demand, orders, reward of the
        # allocation + env step; an evaluation pass keeps the last line only.  MFMA work of the five MLPs as `other`.
        g = shape["gnn"]
        n_nodes = g["initial_node"][2]
        hist = sum((64 + n_out + (n_out if name in ("node_update", "edge_update") else 0)) * n_ent for name, (K, n_out, n_ent, _) in g.items())
        hist += 32 * 2 * n_nodes + f_state
        env = 2 * f_state + S + n_ord + 3 + 1 + g["output"][2]
        flops = sum(2.0 * (K * 32 + 32 * 32 + 32 * n_out) * n_ent for K, n_out, n_ent, _ in g.values())
        return "hbm", 4.0 * ((hist if shape["train"] else 0) + env) * n, "B", ("mfma", flops * n, "FLOP")
    if tag == "gnn_period_bwd" and shape.get("gnn"):
        # the GNN policy's backward of a period in one launch (csrc/gnn_period_bwd.hip).  Matrix-core work: input gradients
        # (W^T dz per layer) + weight gradients (dz x^T per layer) = twice the forward's contractions.  HBM side (`other`): what the
        # forward stored, read once - hidden activations + output per evaluated column, the residual sums, the aggregation, the
        # node features - plus d_out in and the state gradient read-modify-written; the tiles between stages are per-workgroup
        # scratch (L2) and the weight-gradient slabs are per workgroup, not per column: not counted.
        g = shape["gnn"]
        n_nodes, n_live = g["initial_node"][2], g["output"][2]
        flops = 2 * sum(2.0 * (K * 32 + 32 * 32 + 32 * n_out) * n_ent for K, n_out, n_ent, _ in g.values())
        rows = sum((64 + n_out) * n_ent for K, n_out, n_ent, _ in g.values()) + 32 * (n_nodes + n_live) + 64 * n_nodes + \
            g["initial_node"][0] * n_nodes + n_live + 2 * f_state
        return "mfma", flops * n, "FLOP", ("hbm", 4.0 * rows * n, "B")
    if tag in ("alloc_env_fwd", "alloc_env_bwd"):   # GNN: allocation head + env step in one launch (one warehouse)
        n_edges = shape["gnn"]["output"][2] if shape.get("gnn") else S + 2
        if tag == "alloc_env_fwd":  # state read + write, demand, desired quantities of the member / self / supplier edges, orders, sums / ratio / scale, reward
            return "hbm", 4.0 * (2 * f_state + S + (S + 2) + n_ord + 3 + 1) * n, "B"
        # state, orders, demand, desired quantities, incoming state gradient in; state gradient, order gradients (scratch) and d_out out
        return "hbm", 4.0 * (3 * f_state + S + 2 * n_ord + (S + 2) + 3 + n_edges) * n, "B"
    if tag == "head_fwd":  # logits + the warehouse / echelon on-hand row in, orders out
        return "hbm", 4.0 * (shape["n_out"] + Wn + E + n_ord) * n, "B"
    if tag == "head_bwd":  # logits, on-hand rows and order gradients in; logit gradients (+ on-hand gradients) out
        return "hbm", 4.0 * (2 * shape["n_out"] + 2 * (Wn + E) + n_ord) * n, "B"
    hist = shape["F"] + 32 * shape["nh"] + shape["n_out"]
    if tag == "small_rollout_fwd":  # demand in, reward out (+ the activation history when training)
        return "hbm", 4.0 * (2 + (hist if shape["train"] else 0)) * n * T, "B"
    if tag == "small_rollout_bwd":  # history + demand in (weight gradients stay in registers; the first version also wrote dZ)
        dz = 0 if "wgrad" in kernel else 32 * shape["nh"] + shape["n_out"]
        return "hbm", 4.0 * (1 + hist + dz) * n * T, "B"
    if tag in ("horizon_fwd", "horizon_bwd"):
        # whole-horizon data_driven kernels (csrc/horizon_rollout.hip).  Forward: the observation rows' share of the first layer
        # and the demand in; reward and the histories out (state rows, two hidden activations, logits, orders).  Backward: the
        # histories and the demand in, the three pre-activation gradients out.  (A latency-bound chain on n / 16 CUs: the
        # fraction of the HBM roofline is reported for completeness, microseconds per period is the figure of merit.)
        h1, h2 = (list(shape["hidden"]) + [0, 0])[:2]   # (tape modes: no layers; the tape row(s) stand in for z1_obs / the logits)
        no = shape["n_out"] or n_ord
        hist_rows = f_state + h1 + h2 + no + n_ord
        if tag == "horizon_fwd":
            return "hbm", 4.0 * (h1 + S + 1 + (hist_rows if shape["train"] else 0)) * n * T, "B"
        return "hbm", 4.0 * (hist_rows + S + h1 + h2 + no) * n * T, "B"
    if tag == "closed_form_fwd":  # whole-horizon closed-form policy: the demand trace + one state load / store (sums leave per wavefront)
        return "hbm", 4.0 * (S * T + 2 * f_state) * n, "B"
    return None


def sampler_report(sc, device, reps=3):
    """the demand sampler (csrc/sampler.hip) is outside the training step (as data generation is outside the
    reference's), so the step's kernel timer never sees it: the launch that drew this workload's trace is repeated into a scratch
    buffer between HIP events on the launch stream.  HBM-write bound: 4 bytes per (scenario, store, period)."""
    args_ = getattr(sc, "_device_sampler_args", None)
    if args_ is None or getattr(sc, "demands_soa", None) is None:
        return None
    from neural_inventory_control_amd import _lib
    scratch = torch.empty_like(sc.demands_soa)
    T, S, _ = scratch.shape
    sc._generate_on_device(*args_, out=scratch)   # warm-up
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for pair in ev:
        sc._generate_on_device(*args_, out=scratch, events=pair)
    torch.cuda.synchronize(device)
    name = (_lib.lib().nic_last_kernel() or b"").decode()
    ms = statistics.median(a.elapsed_time(b) for a, b in ev)
    # (events sit right around the launch, its operands already on the device; the rocprofv3 summary under profiles/ has the
    # kernel's own duration)
    nbytes = 4.0 * T * S * sc.num_samples
    same = bool(torch.equal(scratch, sc.demands_soa))
    return {"kernel": name, "launches_per_step": 0, "launches_timed": reps, "mean_ms": round(ms, 5), "total_ms_per_step": 0.0,
            "algorithmic_bytes_per_launch": nbytes, "bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "outside_the_step": True, "reproduces_the_trace": same}


def kernel_report(timer, shape, steps):
    summ = timer.summary()
    out = {}
    for tag, (cnt, mean_ms) in summ.items():
        name = timer.names.get(tag, "")
        rec = {"kernel": name, "launches_per_step": cnt / steps, "launches_timed": len(timer.events[tag]),
               "mean_ms": round(mean_ms, 5), "total_ms_per_step": round(cnt * mean_ms / steps, 4)}
        w = algorithmic_work(tag, name, shape)
        if w is not None:
            bound, amount, unit = w[:3]

            def rate(b, a):
                return (a / mean_ms / 1e9, MFMA_F32_PEAK_TFLOPS, "TFLOP/s") if b == "mfma" else \
                       (a / (mean_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s")
            ach, peak, u = rate(bound, amount)
            rec["algorithmic_flops_per_launch" if bound == "mfma" else "algorithmic_bytes_per_launch"] = amount
            rec.update(bound=bound, achieved=round(ach, 2), peak=peak, unit=u, frac=round(ach / peak, 4))
            if len(w) > 3:   # the roofline that does NOT bind this shape, for reference
                ob, oa, _ = w[3]
                oach, opeak, ou = rate(ob, oa)
                rec["other"] = {"bound": ob, "achieved": round(oach, 2), "peak": opeak, "unit": ou, "frac": round(oach / opeak, 4)}
                rec["flop_per_byte"] = round((amount if bound == "mfma" else oa) / (oa if bound == "mfma" else amount), 2)
            nbytes = amount if bound == "hbm" else (w[3][1] if len(w) > 3 else None)
            tr_ = _pmc_traffic(name, shape["n"], tag) if nbytes else None
            if tr_ is not None:
                # counter-measured HBM bytes of this class vs the algorithmic bytes: from COMMITTED rocprofv3 --pmc passes of the
                # same workload (profiles/), not collected in this run.  A launch that contracts over ALL periods (wgradT_*) moves
                # bytes in proportion to the horizon: the counter pass may have run a shorter one (`periods` of the pass), so its
                # bytes are scaled to this run's T before they are compared with this run's algorithmic bytes.
                tb = tr_["bytes_per_launch"]
                if tag.startswith("wgradT_") and tr_.get("periods") and tr_["periods"] != shape["T"]:
                    tb = tb * shape["T"] / tr_["periods"]
                    rec["traffic_scaled_from_periods"] = tr_["periods"]
                rec.update(traffic=tb, traffic_over_algorithmic=round(tb / nbytes, 3),
                           traffic_measured="offline", traffic_source=tr_["source"])
                if bound == "hbm" and tb < 0.95 * nbytes:
                    # gather kernels: rows shared by several entities are counted once per reader in the algorithmic bytes but
                    # come from L2 after the first read; the HBM fraction is then rated on the bytes that actually crossed the
                    # memory interface (both are printed)
                    ach_hbm = tb / (mean_ms * 1e-3) / 1e9
                    rec.update(achieved_algorithmic=rec["achieved"], frac_algorithmic=rec["frac"],
                               achieved=round(ach_hbm, 2), frac=round(ach_hbm / HBM_PEAK_GBS, 4),
                               frac_basis="counter bytes (below the algorithmic bytes: shared gathered rows hit L2)")
        out[tag] = rec
    return out


def bench_epoch(args):
    """`--workload cfg3_yaml | cfg5_yaml`: the reference's SHIPPED configuration through the public epoch loop
    (`main_run.build` -> `Trainer.do_one_epoch`, trainer.py:143-179): 8,192 training samples, shuffled batches of 1,024, 50
    periods, rollout + backward + Adam per batch.  A "step" is one batch; K steps = ceil(K / 8) whole epochs.  The same epoch is
    also timed with the rollout launch sequence replayed from a HIP graph / launched eagerly (whichever the default is not) and
    with the reference's torch `DataLoader` (per-sample collate on the host + H2D per batch, main_run.py:101-103) instead of the
    device-resident batches - reported beside the main figure, not inside it."""
    from neural_inventory_control_amd import _lib, main_run, parallel, workloads
    from neural_inventory_control_amd.rollout import KernelTimer
    rank, world, device = parallel.init_from_env()
    _lib.require_device()
    setting, hyper, desc = workloads.get_epoch(args.workload)
    torch.manual_seed(1234)
    c = main_run.build(setting, hyper, device, rank, world)
    tr, model, opt, loaders = c["trainer"], c["model"], c["optimizer"], c["data_loaders"]
    pp, pbd = c["problem_params"], c["params_by_dataset"]["train"]
    T, S, n_batches = pbd["periods"], pp["n_stores"], len(loaders["train"])
    if args.graph:
        tr.use_rollout_graph = True
    elif args.no_graph:
        tr.use_rollout_graph = False
    if args.tail != "auto":
        tr.fuse_tail = args.tail == "on"
    if args.wide != "auto":
        tr.use_wide = args.wide == "on"
    if args.gnn_period != "auto":
        tr.use_period_kernel = args.gnn_period == "on"
    if args.gnn_bwd != "auto":
        tr.use_period_bwd = args.gnn_bwd == "on"

    def epoch(loader=None):
        return tr.do_one_epoch(opt, loader or loaders["train"], c["loss_function"], c["simulator"], model, T, pp,
                               c["observation_params"], train=True, ignore_periods=pbd["ignore_periods"])

    def timed(n_epochs, loader=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_epochs):
            last = epoch(loader)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_epochs, last

    n_epochs = max(1, (args.steps + n_batches - 1) // n_batches)
    for _ in range(max(1, args.warmup) + 2):   # eager run, auto-graph measurement / capture run, steady state
        epoch()
    # the training-pass engine (MLP / GNN engines: the one of the epoch's main batch shape; tape / closed-form: their own keys)
    eng = tr._engines.get((id(model), True)) or next((e for k, e in tr._engines.items() if k[0] == id(model) and k[-1] is True), None)
    is_gnn, is_tape = type(eng).__name__ == "GnnRollout", type(eng).__name__ == "TapeRollout"
    if world > 1:
        torch.distributed.barrier()
    dt, last = timed(n_epochs)
    if world > 1:
        torch.distributed.barrier()
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    if rank != 0:
        return
    n_samples = len(loaders["train"].dataset)
    ms_batch = dt / n_batches * 1e3
    out = {"metric": "scenario-steps/sec (scenarios x stores x T) per training step", "value": n_samples * S * T / dt,
           "unit": "scenario-steps/s", "n_gpus": world, "steps": n_epochs * n_batches, "warmup": args.warmup, "ms_per_step": ms_batch,
           "ms_per_epoch": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "library_id": (_lib_id() or None),
           "config": {"workload": desc + "; training step = one shuffled batch: rollout fwd + bwd + Adam", "name": args.workload,
                      "samples": n_samples, "batch_size": pbd["batch_size"], "batches_per_epoch": n_batches, "stores": S, "periods": T,
                      "parallelism": f"scenario-sharded dp{world}", "train_loss_per_store_period": last[1],
                      "route": (type(eng).__name__ + (" (whole-horizon kernels)" if getattr(eng, "horizon", None) is not None else ""))
                      if eng is not None else "generic",
                      "rollout_graph": {"setting": tr.use_rollout_graph,
                                        "replaying": bool(eng is not None and getattr(eng, "_graph_on", lambda: False)()),
                                        "auto_probe": getattr(eng, "auto_graph_probe", None)},
                      "step_graph": {"setting": tr.use_step_graph, "captured": bool(tr._step_graphs)}}}
    if eng is not None and not args.no_kernel_timing:
        # the other launch mode, and the reference-style host loader, on the same model (after the main figure)
        alt = {}
        was = tr.use_rollout_graph
        if is_tape:   # the tape route's switches: the whole step as a HIP graph or not; the reference-style loop (Simulator.step) instead
            was_step = tr.use_step_graph
            for label, step_graph, tape in (("tape_route_step_graph_off", False, True), ("generic_route", False, False)):
                tr.use_step_graph, tr.use_tape_rollout = step_graph, tape
                epoch(); epoch()
                alt[label + "_ms_per_epoch"] = round(timed(max(1, n_epochs // 2))[0] * 1e3, 3)
            tr.use_step_graph, tr.use_tape_rollout = was_step, True
            epoch()
        else:
            for label, mode in (("eager", False), ("graph", True)):
                tr.use_rollout_graph = mode
                epoch(); epoch(); epoch()
                alt[label + "_ms_per_epoch"] = round(timed(n_epochs)[0] * 1e3, 3)
            tr.use_rollout_graph = was
        if getattr(eng, "horizon", None) is not None:   # the same epoch on the per-period kernels (eager, then replayed)
            eng.use_horizon = False
            for label, mode in (("per_period_route_eager", False), ("per_period_route_graph", True)):
                tr.use_rollout_graph = mode
                epoch(); epoch(); epoch()
                alt[label + "_ms_per_epoch"] = round(timed(n_epochs)[0] * 1e3, 3)
            eng.use_horizon, tr.use_rollout_graph = True, was
            epoch()
        from torch.utils.data import DataLoader
        host = DataLoader(loaders["train"].dataset, batch_size=pbd["batch_size"], shuffle=True)
        epoch(host)
        alt["torch_dataloader_ms_per_epoch"] = round(timed(1, host)[0] * 1e3, 3)
        out["config"]["epoch_variants"] = alt
        # per-kernel figures of one eager epoch
        tr.use_rollout_graph = False
        was_step, tr.use_step_graph = tr.use_step_graph, (False if is_tape else tr.use_step_graph)
        if is_tape:   # (without a step graph the trainer keeps one engine per train / eval context)
            epoch()
            eng = next((e for k, e in tr._engines.items() if k[0] == id(model) and k[-1] is True and len(k) == 3), eng)
        epoch()
        timer = eng.timer = KernelTimer(stride=args.timing_stride or (1 if is_tape else 10))
        epoch()
        torch.cuda.synchronize()
        eng.timer = None
        tr.use_rollout_graph, tr.use_step_graph = was, was_step
        prob = eng.prob
        dims = [0, 0] if (is_gnn or is_tape) else eng.dims
        shape = dict(n=pbd["batch_size"], T=T, S=S, Wn=prob.Wn, E=prob.E, Ws=prob.Ws, Ww=prob.Ww, We=prob.We, F=dims[0],
                     nh=len(dims) - 2, n_out=dims[-1], train=True, hidden=list(dims[1:-1]),
                     gnn={m.name: (m.K, m.n_out, m.n_live, getattr(m, "fold_rows", 0)) for m in eng.mlp.values()} if is_gnn else None)
        kernels = kernel_report(timer, shape, n_batches)
        rated = {k: v for k, v in kernels.items() if "bound" in v}
        if rated:
            dom = max(rated, key=lambda k: rated[k]["total_ms_per_step"])
            d = rated[dom]
            out["roofline"] = {"bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"], "frac": d["frac"],
                               "traffic": None, "kernel": f"{d['kernel']} ({dom})", "mean_launch_ms": d["mean_ms"],
                               "launches_per_step": d["launches_per_step"],
                               ("algorithmic_flops_per_launch" if d["bound"] == "mfma" else "algorithmic_bytes_per_launch"):
                                   d.get("algorithmic_flops_per_launch", d.get("algorithmic_bytes_per_launch"))}
        out["kernels"] = kernels
    if world == 1 and not args.no_cpu_baseline:
        # the oracle on ONE batch of the epoch (same setting, batch size and horizon; gradient step without the optimizer)
        try:
            import copy
            st2, hy2, _ = workloads.get_epoch(args.workload)
            out["cpu_baseline"] = cpu_baseline(args.workload, pbd["batch_size"], T, model=model, setting_policy=(st2, hy2["nn_params"]))
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "scenario-steps/s", "cores": None, "host_cores": os.cpu_count(),
                                   "kind": "port", "sample": f"failed: {e!r}"}
    if parallel.active():
        torch.distributed.destroy_process_group()
    _emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3",
                    help="one training step of: cfg1..cfg5 (BASELINE's configs; default cfg3), cfg3_shard8, cfg3_batch1024, base_stock, "
                         "base_stock_1m, echelon_stock, gnn, gnn_many_warehouses, real_data_driven, cfg3_symmetry_aware; or one training "
                         "EPOCH of a shipped YAML pair through Trainer.do_one_epoch: cfg3_yaml, cfg5_yaml, gnn_yaml, real_data_yaml, "
                         "one_store_real_yaml, one_store_real_transformed_nv_yaml (neural_inventory_control_amd/workloads.py)")
    ap.add_argument("--scenarios", type=int, default=None, help="scenarios per GPU (default: the workload's)")
    ap.add_argument("--periods", type=int, default=None)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = the workload's scenario count PER GPU (65,536 per GPU for cfg3; the default), strong = "
                         "the workload's scenario count in TOTAL, sharded over the N ranks (cfg3: 8,192 per GPU at N = 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=None,
                    help="bracket every n-th launch of each kernel class with HIP events (default: 40 on the per-period route, "
                         "1 on the whole-horizon route; each pair costs ~5 us)")
    ap.add_argument("--launch-order-out", default=None,
                    help="write the (kernel class, kernel) sequence of the timed steps as JSON and skip the event timing: what "
                         "tools/collect_profiles.py joins the profiler's per-dispatch counter rows to")
    ap.add_argument("--graph", action="store_true", help="replay the launch sequence from a HIP graph (implies --no-kernel-timing)")
    ap.add_argument("--no-graph", action="store_true", help="epoch workloads: never replay (default: the Trainer's measured choice)")
    ap.add_argument("--generic-route", action="store_true",
                    help="A/B: run the workload on the generic per-period route even where a fused engine exists")
    ap.add_argument("--lane-scenarios", type=int, default=0, choices=(0, 16, 32),
                    help="whole-horizon route: scenarios per wavefront (0 = chosen by the library from the batch size)")
    ap.add_argument("--gnn-period", choices=("auto", "on", "off"), default="auto",
                    help="gnn: the forward of a period as ONE launch (csrc/gnn_period.hip): the engine's choice, forced on, forced off")
    ap.add_argument("--gnn-bwd", choices=("auto", "on", "off"), default="auto",
                    help="gnn: the backward of a period as ONE launch (csrc/gnn_period_bwd.hip): the engine's choice, forced on, forced off")
    ap.add_argument("--gnn-keep-inputs", action="store_true",
                    help="gnn: keep a copy of the gathered MLP inputs for the backward instead of reading them again (A/B)")
    ap.add_argument("--eval", action="store_true",
                    help="SURVEY 8(f2): time the forward-only evaluation pass (Trainer.test: no gradients, discrete allocation "
                         "for Poisson demand) instead of a training step; use with --periods 5000 for the reference's test horizon")
    ap.add_argument("--adam", choices=["fused", "foreach"], default="fused",
                    help="torch.optim.Adam implementation: torch's one-kernel form (default) or its for-each form (torch's own default)")
    ap.add_argument("--no-dist-init", action="store_true",
                    help="N = 1: do not create the one-rank RCCL process group (default: created, so that the gradient all-reduce of the "
                         "sharded path really runs and the line's `collective` object describes it)")
    ap.add_argument("--tail", choices=("auto", "on", "off"), default="auto",
                    help="fused per-period tail launches (csrc/period_tail.hip): the engine's choice, forced on, forced off")
    ap.add_argument("--wide", choices=("auto", "on", "off"), default="auto",
                    help="whole-horizon forward kernel of the 512-wide policy (csrc/wide_rollout.hip): the engine's choice, on, off")
    ap.add_argument("--no-horizon", action="store_true",
                    help="data_driven workloads: the per-period kernels instead of the whole-horizon kernels (A/B)")
    ap.add_argument("--horizon-max-scenarios", type=int, default=0,
                    help="data_driven workloads: largest batch that takes the whole-horizon kernels (default: the engine's 8192)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    _quiet_stdout()
    from neural_inventory_control_amd import workloads as _w
    if args.workload in _w.EPOCH_WORKLOADS:
        return bench_epoch(args)
    from neural_inventory_control_amd import _lib, parallel
    from neural_inventory_control_amd.rollout import KernelTimer
    dist_error = None
    os.environ["NCCL_DEBUG"] = os.environ.get("NIC_NCCL_DEBUG", "WARN")   # (no RCCL version banner between the lines of stdout)
    if args.gpus == 1 and "RANK" not in os.environ and not args.no_dist_init:
        os.environ.setdefault("NIC_DIST_FORCE_INIT", "1")   # one rank, real backend: the collective path is exercised and described
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    try:
        rank, world, device = parallel.init_from_env()
    except Exception as e:   # (a one-rank group that cannot be created must not cost the single-GPU number)
        if args.gpus > 1:
            raise
        dist_error = f"{type(e).__name__}: {e}"[:300]
        os.environ["NIC_DIST_FORCE_INIT"] = "0"
        rank, world, device = parallel.init_from_env()
    _lib.require_device()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)

    per_gpu = args.scenarios
    if args.scaling == "strong" and world > 1:
        total = args.scenarios or _w.get(args.workload)[2]   # --scenarios then names the GLOBAL count
        if total % world:
            print(f"bench.py: --scaling strong needs a scenario count ({total}) divisible by --gpus ({world})", file=sys.stderr)
            sys.exit(2)
        per_gpu = total // world
    global _TRAFFIC_WORKLOAD
    _TRAFFIC_WORKLOAD = args.workload
    setting, policy, sc, data, model, eng, n, T, desc = build_case(args.workload, device, rank, world, per_gpu,
                                                                   args.periods, args.generic_route)
    pp = setting["problem_params"]
    S = pp["n_stores"]
    closed_form = eng is not None and type(eng).__name__ == "ClosedFormRollout"
    gnn = eng is not None and type(eng).__name__ == "GnnRollout"
    if gnn:
        if args.gnn_keep_inputs:
            eng.keep_inputs = True
        eng.use_period_kernel = {"auto": "auto", "on": True, "off": False}[args.gnn_period]
        eng.use_period_bwd = {"auto": "auto", "on": True, "off": False}[args.gnn_bwd]
        eng.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
        parallel.broadcast_model(model, src=0)
    elif closed_form:
        with torch.no_grad():
            eng.model.closed_form_levels()  # materialises the policy's one lazy layer
        parallel.broadcast_model(model, src=0)
    elif eng is not None:  # materialise the lazy layers now so that replicas can be synchronised before the first step
        eng.materialize(eng.input_rows(data, setting["observation_params"]))
        eng.small_lane_scenarios = args.lane_scenarios
        eng.inputs_versioned = True   # (the bench's batch is written once, before the first step: unchanged tensors are not re-copied)
        if args.no_horizon:
            eng.use_horizon = False
        if args.wide != "auto" and hasattr(eng, "use_wide"):
            eng.use_wide = args.wide == "on"
        if args.tail != "auto" and hasattr(eng, "fuse_tail"):
            eng.fuse_tail = args.tail == "on"
        if args.horizon_max_scenarios:
            eng.horizon_max_scenarios = args.horizon_max_scenarios
        parallel.broadcast_model(model, src=0)
    # the reference's optimizer (main_run.py: torch.optim.Adam); torch's single-kernel implementation of it where the parameters
    # allow (same update rule; the default for-each form is six launches per step, 30 us of the small policies' ~1-ms steps)
    from torch.nn.parameter import UninitializedParameter
    fused_adam = (args.adam == "fused" and device.type == "cuda"
                  and not any(isinstance(p_, UninitializedParameter) for p_ in model.parameters()))
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, **({"fused": True} if fused_adam else {}))
    # closed-form policies: what `Trainer` does by default (`use_step_graph = "auto"`: their training step - level network, one
    # whole-horizon launch, autograd through the level network, Adam - is captured once and replayed; the launch is 0.1 ms inside
    # ~0.6 ms of host work per eager step).  `--no-graph` times the eager step; the kernel itself is timed in a separate eager pass
    # behind the timed region either way (`kernels`, `roofline`).
    args.no_kernel_timing_flag = bool(args.no_kernel_timing)
    closed_form_replay = closed_form and not args.no_graph and not args.eval and not args.launch_order_out
    if closed_form_replay:
        args.graph = True
    # GNN: what `Trainer` sets on its engines ("auto": the second training run measures host enqueue time against GPU time and later
    # runs are replayed from HIP graphs if the host is the slower one - with three launches per period that depends on the HOST:
    # 22 ms steps on a quiet box, 35+ ms with eager launches when its cores are busy).  Eager launches: --no-graph.  The per-kernel
    # timer needs eager launches, so the kernels are timed in one extra step behind the timed region.
    gnn_auto = gnn and not args.graph and not args.no_graph and not args.eval and not args.launch_order_out
    if gnn_auto:
        eng.use_graph = "auto"
    if args.graph and eng is not None and not closed_form:
        eng.use_graph = True
        args.no_kernel_timing = True
    sharded = parallel.active()  # a process group exists (N ranks, or one rank under NIC_DIST_FORCE_INIT=1)
    reducer = parallel.GradientAllReducer.get(model) if sharded else None
    # every rank's device, gathered once: an N-rank job must have run on N distinct devices
    idents = parallel.device_identities(device)
    # (enforced for the real backend; the shared-GPU test mode - NIC_DIST_BACKEND=gloo, N ranks on one device - only records it)
    if len({(h, d) for _, h, d in idents}) != world and (not sharded or torch.distributed.get_backend() == "nccl"):
        print(f"bench.py: {world} ranks but only {len({(h, d) for _, h, d in idents})} distinct devices: {idents}", file=sys.stderr)
        sys.exit(2)
    global_b = n * world
    grad_scale = 1.0 / (global_b * T * S)

    discrete = bool(args.eval) and setting["store_params"]["demand"].get("distribution") == "poisson"

    def eval_step():
        with torch.no_grad():
            total, _ = eng.run(data, T, 0, train=False, observation_params=setting["observation_params"],
                               demand_soa=sc.demands_soa, discrete_allocation=discrete)
        return total

    if eng is None or (closed_form and args.graph):  # reference-style loop through the Trainer (with --graph: whole step replayed)
        from neural_inventory_control_amd.environment import Simulator
        from neural_inventory_control_amd.loss_functions import PolicyLoss
        from neural_inventory_control_amd.trainer import Trainer
        sim, tr, loss_fn = Simulator(device=device), Trainer(device=device), PolicyLoss()
        tr.inputs_versioned = True   # (the bench's batch is written once, before the first step: unchanged tensors are not re-copied)
        args.no_kernel_timing = True
        tr._global_batch = global_b
        if args.generic_route:
            tr.use_fused_rollout = False

        def generic_step():
            if args.graph:  # whole training step (all periods + autograd sweep) replayed from one HIP graph
                opt.zero_grad(set_to_none=False)
                total, _ = tr._graphed_generic_step(loss_fn, sim, model, T, pp, data, setting["observation_params"], 0,
                                                    global_batch=global_b)
            else:
                opt.zero_grad(set_to_none=True)
                total, _ = tr.simulate_batch(loss_fn, sim, model, T, pp, data, setting["observation_params"], 0, False)
                (total * grad_scale).backward()
            if reducer is not None:
                total, _ = reducer.all_reduce(total.detach(), total.detach())
            clip = getattr(model, "gradient_clipping_norm_value", None)
            if clip is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
            return total.detach()

    def step():
        if eng is None or (closed_form and args.graph and not args.eval):
            return generic_step()
        if args.eval:
            return eval_step()
        opt.zero_grad(set_to_none=True)
        if closed_form:  # one launch: rollout + forward-mode gradient; autograd only chains through the policy's tiny net
            total, reported = eng.run(data, T, 0, train=True, observation_params=setting["observation_params"],
                                      demand_soa=sc.demands_soa)
            (total * grad_scale).backward()
            total, reported = total.detach(), reported.detach()
            if reducer is not None:
                total, reported = reducer.all_reduce(total, reported)
            opt.step()
            return total
        total, reported = eng.run(data, T, 0, train=True, observation_params=setting["observation_params"],
                                  demand_soa=sc.demands_soa, grad_scale=grad_scale)
        if reducer is not None:
            total, reported = reducer.all_reduce(total, reported)
        clip = getattr(model, "gradient_clipping_norm_value", None)
        if clip is not None:  # (gnn.yml clips the gradient norm at 1.0, trainer.py:175-176)
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        return total

    for _ in range(max(args.warmup, 0) + (2 if args.graph else 0)):  # graph mode: eager run + capture run before timing
        step()
    if eng is None and args.warmup == 0:
        step()  # lazy layers materialise on the first forward; keep that out of the timed region
    timer = None
    if args.launch_order_out and eng is not None:
        timer = eng.timer = KernelTimer(record_order=True)
    elif closed_form_replay or gnn_auto:
        pass   # (timed behind the region, below)
    elif not args.no_kernel_timing:
        whole = closed_form or (not gnn and (eng.small is not None or getattr(eng, "horizon", None) is not None))
        stride = args.timing_stride or (1 if whole else (10 if gnn else 40))
        timer = eng.timer = KernelTimer(stride=stride)
    if reducer is not None:
        reducer.timing, reducer._events, reducer.calls = True, [], 0
    if sharded:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if sharded:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if sharded:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    timer_steps = args.steps
    if closed_form_replay and not getattr(args, "no_kernel_timing_flag", False):
        # the whole-horizon kernel of the replayed step, timed in eager launches of the same engine on the same batch
        timer = eng.timer = KernelTimer(stride=1)
        timer_steps = 3
        for _ in range(timer_steps):
            eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        torch.cuda.synchronize()
        eng.timer = None
    if gnn_auto and not getattr(args, "no_kernel_timing_flag", False):
        timer = eng.timer = KernelTimer(stride=1)   # (with a timer attached the engine launches eagerly)
        timer_steps = 1
        eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa, grad_scale=grad_scale)
        torch.cuda.synchronize()
        eng.timer = None
    loss = float(last) / (global_b * T * S)
    collective = {"backend": torch.distributed.get_backend() if sharded else None, "world_size": world,
                  "ranks_seen": len(idents), "distinct_devices": len({(h, d) for _, h, d in idents}),
                  "devices": [d for _, _, d in idents], "op": "all_reduce(SUM) of one flat fp32 buffer [gradients..., total, reported]"}
    if reducer is not None:
        reducer.timing = False
        st_ = reducer.collective_stats()
        collective.update(allreduce_bytes=st_["allreduce_bytes"], allreduces_per_step=st_["allreduce_calls"] / max(args.steps, 1),
                          allreduce_ms=None if st_["allreduce_ms"] is None else round(st_["allreduce_ms"], 4))
    if dist_error:
        collective["init_error"] = dist_error

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "scenario-steps/sec (scenarios x stores x T) per " + ("evaluation pass" if args.eval else "training step"),
            "value": global_b * S * T * args.steps / dt, "unit": "scenario-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32",
            "scaling_note": ("BASELINE's '>= 6x further at 8 GPUs' is stated on the fixed 65,536-scenario problem = --scaling strong "
                             "(predicted from 1-GPU shard steps: profiles/r06_scaling_prediction.json); the default, and this line"
                             + ("" if (world > 1 and args.scaling == "strong") else " unless run with --scaling strong") +
                             ", is WEAK scaling: the workload's scenario count PER GPU"),
            "data": "synthetic", "collective": collective,
            # what the numbers were measured on: the content hash of the HIP sources + flags + headers the library was built from
            # (nic_build_id(); `python -c "from neural_inventory_control_amd import build; print(build.source_id())"` at a commit)
            "library_id": (_lib_id() or None),
            "config": {"workload": desc + ("; evaluation pass = forward rollout only" + (", discrete allocation" if discrete else "")
                                           if args.eval else
                                           "; training step = rollout fwd + bwd + Adam" + (" + RCCL grad all-reduce" if sharded else "")),
                       "name": args.workload, "scenarios_per_gpu": n, "global_scenarios": global_b, "stores": S,
                       "periods": T, "parallelism": f"scenario-sharded dp{world}",
                       "route": ("generic (Simulator.step + autograd)" if eng is None else
                                 ("whole-horizon closed-form kernel (forward-mode gradient)" +
                                  (", training step replayed from a HIP graph (the Trainer's default for these policies)" if closed_form_replay else "")) if closed_form else
                                 ((("one forward launch per period (five MLPs on LDS-resident embeddings + allocation + env step)"
                                    if getattr(eng, "_period", False) else "per-period fused gather-MLP forward kernels over the static supply graph")
                                   + (", one backward launch per period behind the env / allocation adjoint (five MLP adjoints, adjoint gathers, in-kernel weight gradients)"
                                      if getattr(eng, "_period_bwd", False) else ", per-MLP backward launches"))) if gnn else
                                 "whole-horizon kernels" if eng.small is not None else
                                 "whole-horizon kernels (16 scenarios per workgroup) + (period x scenario) GEMMs"
                                 if getattr(eng, "horizon", None) is not None else "per-period kernels"),
                       "mean_cost_per_store_period": loss},
        }
        if gnn and getattr(eng, "auto_graph_probe", None):   # host enqueue time against GPU time of the second step, and the decision
            out["config"]["launch_mode"] = dict(eng.auto_graph_probe, note="replay = later steps replayed from HIP graphs (`use_graph = 'auto'`)")
        if not args.eval:
            out["config"]["optimizer"] = "torch.optim.Adam(lr=3e-4" + (", fused=True)" if fused_adam else ")")
        if timer is not None and timer.order is not None:
            json.dump({"workload": args.workload, "n_scenarios": n, "periods": T, "steps": args.steps, "order": timer.order},
                      open(args.launch_order_out, "w"))
        elif timer is not None:
            Wn_, E_ = pp["n_warehouses"], pp["n_extra_echelons"]
            shape = dict(n=n, T=T, S=S, Wn=Wn_, E=E_, Ws=data["initial_inventories"].shape[2],
                         Ww=data["initial_warehouse_inventories"].shape[2] if Wn_ else 0,
                         We=data["initial_echelon_inventories"].shape[2] if E_ else 0,
                         F=0 if (closed_form or gnn) else eng.dims[0], nh=0 if (closed_form or gnn) else len(eng.dims) - 2,
                         n_out=0 if (closed_form or gnn) else eng.dims[-1], train=not args.eval,
                         hidden=[] if (closed_form or gnn) else list(eng.dims[1:-1]),
                         gnn={m.name: (m.K, m.n_out, m.n_live, getattr(m, "fold_rows", 0)) for m in eng.mlp.values()} if gnn else None)
            kernels = kernel_report(timer, shape, timer_steps)
            rated = {k: v for k, v in kernels.items() if "bound" in v}
            if rated:
                dom = max(rated, key=lambda k: rated[k]["total_ms_per_step"])
                d = rated[dom]
                out["roofline"] = {
                    "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"], "frac": d["frac"],
                    "traffic": None, "kernel": f"{d['kernel']} ({dom})",
                    ("algorithmic_flops_per_launch" if d["bound"] == "mfma" else "algorithmic_bytes_per_launch"):
                        d.get("algorithmic_flops_per_launch", d.get("algorithmic_bytes_per_launch")),
                    "mean_launch_ms": d["mean_ms"], "launches_per_step": d["launches_per_step"],
                    "launches_timed": d["launches_timed"], "share_of_step": round(d["total_ms_per_step"] / ms, 4),
                }
                tr_ = _pmc_traffic(d["kernel"], n, dom)
                if tr_ is not None:   # (HBM bytes per launch from the committed counter passes, see kernel_report)
                    out["roofline"]["traffic"] = d.get("traffic", tr_["bytes_per_launch"])
                    out["roofline"]["traffic_measured"] = "offline"
                    out["roofline"]["traffic_source"] = tr_["source"]
                if "other" in d:
                    out["roofline"]["other"] = d["other"]
            smp = sampler_report(sc, device)
            if smp is not None:
                kernels["sampler"] = smp
            out["kernels"] = kernels
            env_tag = next((t_ for t_ in ("env_fwd", "head_env_fwd", "alloc_env_fwd") if t_ in kernels), "env_fwd")
            if env_tag in kernels and "bound" in kernels[env_tag]:
                e = kernels[env_tag]
                out["roofline_env_step"] = {"bound": "hbm", "achieved": e["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": e["frac"], "kernel": e["kernel"],
                                            "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
                                            "mean_launch_ms": e["mean_ms"]}
        if world == 1 and not args.no_cpu_baseline and not args.eval:
            sample = args.cpu_sample or {"cfg3": 4096, "cfg5": 1024, "cfg2": 32768, "cfg4": 16384, "cfg1": 256, "base_stock": 32768, "gnn": 512, "gnn_many_warehouses": 256,
                                         "real_data_driven": 72}.get(args.workload, 1024)
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, min(sample, n), T, model=model)
                out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "scenario-steps/s", "cores": None, "host_cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
    if sharded:
        torch.distributed.destroy_process_group()
    if rank == 0:   # (the only thing this process writes to its real stdout, see _quiet_stdout)
        _emit(out)


if __name__ == "__main__":
    main()
