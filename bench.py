#!/usr/bin/env python
"""Benchmark of the hot path: one TRAINING STEP of the differentiable inventory rollout
(reset + T x (policy forward + env step) + cost reduction + backward through the horizon [+ gradient all-reduce] + Adam)
on synthetic demand, BASELINE.json's metric: scenario-steps/s = scenarios x n_stores x T / wall-time.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3]

N = 1: plain process.  N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, one
rank per GPU; scenarios are sharded (weak scaling: 65,536 scenarios PER GPU for cfg3), one RCCL all-reduce of the flat
gradient per step.  Rank 0 prints ONE JSON line.  Inputs (demand traces from the HIP Philox sampler, initial state,
weights) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X dense FP32 matrix peak (MI355X_MICROARCH.md)


def build_case(workload, device, rank, scenarios=None, periods=None):
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from neural_inventory_control_amd.data_handling import Scenario
    from neural_inventory_control_amd.neural_networks import NeuralNetworkCreator
    from neural_inventory_control_amd.rollout import FusedRollout
    setting, policy, n, T, desc = workloads.get(workload)
    if scenarios or periods:
        desc += f" [overridden: {scenarios or n} scenarios/GPU x T={periods or T}]"
    n, T = scenarios or n, periods or T
    obs = defaultdict(lambda: None, setting["observation_params"])
    sc = Scenario(T, setting["problem_params"], setting["store_params"], setting["warehouse_params"],
                  setting["echelon_params"], n, obs, setting["seeds"], sampler="hip", device=device,
                  scenario_offset=rank * n)
    data = {k: v.to(device) for k, v in sc.get_data().items()}
    torch.manual_seed(1234)  # identical initial weights on every rank
    model = NeuralNetworkCreator().create_neural_network(sc, policy, device=device)
    eng = FusedRollout(model, setting["problem_params"], device) if FusedRollout.supports(model) else None
    return setting, policy, sc, data, model, eng, n, T, desc


def _pmc_traffic(kind, N, K, n_scenarios):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_traffic.json):
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, FETCH doubled (gfx950 reports half of a wide coalesced
    read stream), both x1024 (KiB units).  None if no pass matches this kernel / shape."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            for e in json.load(open(f)):
                if e["kind"] == kind and e["N"] == N and e["K"] == K and e["n_scenarios"] == n_scenarios:
                    return {"bytes_per_launch": e["hbm_bytes_per_launch"], "source": os.path.basename(f)}
        except Exception:
            pass
    return None


def _pick_threads(avail):
    import torch.nn.functional as F
    x, w = torch.randn(4096, 512), torch.randn(512, 512)
    best, best_t = 1, None
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        for _ in range(2):
            F.elu(F.linear(x, w))
        t0 = time.perf_counter()
        for _ in range(10):
            F.elu(F.linear(x, w))
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    return best


def cpu_baseline(workload, sample_scenarios, periods):
    """The oracle (CPU restatement of the reference path, PyTorch eager) timed on this host's cores: one training step
    (rollout + backward) of the same workload on a bounded sample of scenarios."""
    from collections import defaultdict
    from neural_inventory_control_amd import workloads
    from oracle import inventory_oracle as orc
    # eager PyTorch on small tensors collapses when oversubscribed (256 threads on this path ran 40x slower than 8), so
    # the thread count is calibrated on a short rollout and the best one is used and reported
    avail = os.cpu_count() or 1
    cores = _pick_threads(avail)
    torch.set_num_threads(cores)
    setting, policy, _, _, _ = workloads.get(workload)
    obs = defaultdict(lambda: None, setting["observation_params"])
    data = orc.generate_scenario_data(periods, setting["problem_params"], setting["store_params"],
                                      setting["warehouse_params"], setting["echelon_params"], sample_scenarios, obs,
                                      setting["seeds"])
    S, Wn, E = (setting["problem_params"][k] for k in ("n_stores", "n_warehouses", "n_extra_echelons"))
    F = data["initial_inventories"].shape[1] * data["initial_inventories"].shape[2]
    if policy["name"] != "vanilla_one_store":
        F += sum(data[k].shape[1] * data[k].shape[2] for k in ("initial_warehouse_inventories", "initial_echelon_inventories")
                 if k in data)
    pol = orc.init_policy(policy, setting["problem_params"], F, 1234, setting["store_params"])
    warm = {k: v[:max(8, sample_scenarios // 16)] for k, v in data.items()}
    orc.train_step_gradients(pol, min(periods, 10), setting["problem_params"], warm, obs)
    t0 = time.perf_counter()
    orc.train_step_gradients(pol, periods, setting["problem_params"], data, obs)
    dt = time.perf_counter() - t0
    return {"value": sample_scenarios * S * periods / dt, "unit": "scenario-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch-CPU eager restatement of the reference path), 1 training step fwd+bwd, "
                      f"{sample_scenarios} scenarios x {S} stores x T={periods}, {cores} threads, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scenarios", type=int, default=None, help="scenarios per GPU (default: the workload's)")
    ap.add_argument("--periods", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=10,
                    help="bracket every n-th launch of each kernel class with HIP events (1 = all; each pair costs ~5 us)")
    ap.add_argument("--graph", action="store_true", help="replay the launch sequence from a HIP graph (implies --no-kernel-timing)")
    ap.add_argument("--eval", action="store_true",
                    help="SURVEY 8(f2): time the forward-only evaluation pass (Trainer.test: no gradients, discrete allocation "
                         "for Poisson demand) instead of a training step; use with --periods 5000 for the reference's test horizon")
    args = ap.parse_args()

    from neural_inventory_control_amd import _lib, parallel
    from neural_inventory_control_amd.rollout import KernelTimer
    rank, world, device = parallel.init_from_env()
    _lib.require_device()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    setting, policy, sc, data, model, eng, n, T, desc = build_case(args.workload, device, rank, args.scenarios, args.periods)
    S = setting["problem_params"]["n_stores"]
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    if args.graph and eng is not None:
        eng.use_graph = True
        args.no_kernel_timing = True
    reducer = parallel.GradientAllReducer.get(model) if world > 1 else None
    global_b = n * world
    grad_scale = 1.0 / (global_b * T * S)

    discrete = bool(args.eval) and setting["store_params"]["demand"].get("distribution") == "poisson"

    def eval_step():
        with torch.no_grad():
            total, _ = eng.run(data, T, 0, train=False, observation_params=setting["observation_params"],
                               demand_soa=sc.demands_soa, discrete_allocation=discrete)
        return total

    if eng is None:  # policies outside the fused engine (GNN, closed-form, user plugins): the reference-style loop
        from neural_inventory_control_amd.environment import Simulator
        from neural_inventory_control_amd.loss_functions import PolicyLoss
        from neural_inventory_control_amd.trainer import Trainer
        sim, tr, loss_fn = Simulator(device=device), Trainer(device=device), PolicyLoss()
        args.no_kernel_timing = True

        tr._global_batch = global_b

        def generic_step():
            if args.graph:  # whole training step (all periods + autograd sweep) replayed from one HIP graph
                opt.zero_grad(set_to_none=False)
                total, _ = tr._graphed_generic_step(loss_fn, sim, model, T, setting["problem_params"], data,
                                                    setting["observation_params"], 0)
            else:
                opt.zero_grad(set_to_none=True)
                total, _ = tr.simulate_batch(loss_fn, sim, model, T, setting["problem_params"], data,
                                             setting["observation_params"], 0, False)
                (total * grad_scale).backward()
            if reducer is not None:
                total, _ = reducer.all_reduce(total.detach(), total.detach())
            clip = getattr(model, "gradient_clipping_norm_value", None)
            if clip is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
            return total.detach()

    def step():
        if eng is None:
            return generic_step()
        if args.eval:
            return eval_step()
        opt.zero_grad(set_to_none=True)
        total, reported = eng.run(data, T, 0, train=True, observation_params=setting["observation_params"],
                                  demand_soa=sc.demands_soa, grad_scale=grad_scale)
        if reducer is not None:
            total, reported = reducer.all_reduce(total, reported)
        opt.step()
        return total

    for _ in range(max(args.warmup, 0) + (2 if args.graph else 0)):  # graph mode: eager run + capture run before timing
        step()
    timer = None
    if not args.no_kernel_timing:
        timer = eng.timer = KernelTimer(stride=args.timing_stride)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    loss = float(last) / (global_b * T * S)

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "scenario-steps/sec (scenarios x stores x T) per " + ("evaluation pass" if args.eval else "training step"),
            "value": global_b * S * T * args.steps / dt, "unit": "scenario-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc + ("; evaluation pass = forward rollout only" + (", discrete allocation" if discrete else "")
                                           if args.eval else
                                           "; training step = rollout fwd + bwd + Adam" + (" + RCCL grad all-reduce" if world > 1 else "")),
                       "name": args.workload, "scenarios_per_gpu": n, "global_scenarios": global_b, "stores": S,
                       "periods": T, "parallelism": f"scenario-sharded dp{world}",
                       "mean_cost_per_store_period": loss},
        }
        if timer is not None:
            summ = timer.summary()
            dims = eng.dims
            gemm = {}
            for tag, (cnt, mean_ms) in summ.items():
                kind, _, shape = tag.partition("_")
                if kind in ("fwd", "dgrad", "wgrad", "wgradT") and "x" in shape:
                    N_, K_ = (int(v) for v in shape.split("x"))
                    flops = 2.0 * N_ * K_ * n * (T if kind == "wgradT" else 1)  # wgradT: all T periods in one launch
                    gemm[tag] = {"launches": cnt, "mean_ms": mean_ms, "total_ms_per_step": cnt * mean_ms / args.steps,
                                 "tflops": flops / mean_ms / 1e9}
            if gemm:
                dom = max(gemm, key=lambda k: gemm[k]["total_ms_per_step"])
                kind, _, shape = dom.partition("_")
                N_, K_ = (int(v) for v in shape.split("x"))
                out["roofline"] = {
                    "bound": "mfma", "achieved": gemm[dom]["tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": gemm[dom]["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                    "kernel": {"fwd": "gemm_wx_dma_kernel<2,4,4,2,EPI_BIAS_ACT>", "dgrad": "gemm_wx_dma_kernel<2,4,4,2,EPI_DGRAD>",
                               "wgrad": "gemm_wgrad_dma_kernel<2,4,4,2>",
                               "wgradT": "gemm_wgrad_dma_kernel<2,4,4,2> over all periods"}.get(kind, kind) + f" ({dom})",
                    "algorithmic_flops_per_launch": 2.0 * N_ * K_ * n * (T if kind == "wgradT" else 1),
                    "mean_launch_ms": gemm[dom]["mean_ms"],
                    "launches": gemm[dom]["launches"], "launches_timed": len(timer.events[dom]),
                }
                tr = _pmc_traffic(kind, N_, K_, n)
                if tr is not None:
                    out["roofline"]["traffic"] = tr["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tr["source"]
                out["kernels"] = {k: {kk: round(vv, 5) if isinstance(vv, float) else vv for kk, vv in v.items()}
                                  for k, v in sorted(gemm.items())}
                pp = setting["problem_params"]
                Wn_, E_ = pp["n_warehouses"], pp["n_extra_echelons"]
                Ws_ = data["initial_inventories"].shape[2]
                Ww_ = data["initial_warehouse_inventories"].shape[2] if Wn_ else 0
                We_ = data["initial_echelon_inventories"].shape[2] if E_ else 0
                # state read + write, demand, orders, reward (static tables amortised over T)
                env_bytes = 4.0 * (2 * (S * Ws_ + Wn_ * Ww_ + E_ * We_) + S + (S * max(Wn_, 1) + Wn_ + E_) + 1) * n
                for tag in ("env_fwd", "env_bwd", "small_rollout_fwd", "small_rollout_bwd"):
                    if tag in summ:
                        out["kernels"][tag] = {"launches": summ[tag][0], "mean_ms": round(summ[tag][1], 5)}
                for tag, (cnt, mean_ms) in summ.items():  # fused thin-layer backward: HBM-bound, X in + dX out
                    if tag.startswith("bwd_thin_"):
                        N_t, K_t = (int(v) for v in tag[len("bwd_thin_"):].split("x"))
                        out["kernels"][tag] = {"launches": cnt, "mean_ms": round(mean_ms, 5),
                                               "total_ms_per_step": round(cnt * mean_ms / args.steps, 5),
                                               "gb_per_s": round(2.0 * K_t * n * 4 / (mean_ms * 1e-3) / 1e9, 1)}
                if "env_fwd" in summ:
                    gbs = env_bytes / (summ["env_fwd"][1] * 1e-3) / 1e9
                    out["roofline_env_step"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                                "frac": gbs / 8000.0, "kernel": "env_step_fwd_kernel",
                                                "algorithmic_bytes_per_launch": env_bytes,
                                                "mean_launch_ms": summ["env_fwd"][1]}
        if world == 1 and not args.no_cpu_baseline and not args.eval:
            sample = args.cpu_sample or {"cfg3": 4096, "cfg5": 1024, "cfg2": 32768, "cfg4": 16384, "cfg1": 256}.get(args.workload, 1024)
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, min(sample, n), T)
                out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "scenario-steps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
