"""Scenario-sharded training on the device: two ranks (sharing the one GPU of the test box; collectives over gloo, kernels
on HIP) must train to the same parameters as one process — DeviceBatches slices every global batch by rank, each rank
scales by the GLOBAL batch, one flat all-reduce per optimizer step.  The CPU suite covers the collective and
sharding logic (test_distributed_gloo.py); this covers it around the real kernels."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(cmd, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DDP_RESULT ")][-1]
    return json.loads(line[len("DDP_RESULT "):])


def test_two_rank_training_equals_single_process():
    helper = os.path.join(HERE, "ddp_helper.py")
    single = _run([sys.executable, helper])
    double = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", "29541", helper], {"NIC_DIST_BACKEND": "gloo"})
    assert single["world"] == 1 and double["world"] == 2
    assert abs(single["test_loss"] - double["test_loss"]) <= 1e-6 * abs(single["test_loss"]), (single, double)


@pytest.mark.parametrize("name,n_total,T", [("cfg3", 300, 9), ("cfg1", 130, 7)])
def test_sharded_device_scenarios_equal_single_process(tmp_path, name, n_total, T):
    """SURVEY §8e 'extra exchange': two ranks generate their rows with the HIP sampler; the initial inventories use the GLOBAL
    per-store demand mean (one S-float all-reduce) and the global multiplier rows — the concatenation is the 1-process set."""
    import torch
    helper = os.path.join(HERE, "shard_helper.py")
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir()
    two.mkdir()
    args = [name, str(n_total), str(T)]
    r = subprocess.run([sys.executable, helper, str(one)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, NIC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", helper, str(two)] + args, capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    whole = torch.load(one / "rank0.pt")
    parts = [torch.load(two / f"rank{k}.pt") for k in (0, 1)]
    for k, v in whole.items():
        cat = torch.cat([p[k] for p in parts], dim=0)
        assert torch.equal(cat, v), k


def test_training_over_a_one_rank_rccl_group_equals_no_group():
    """RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so the test box cannot run a 2-rank nccl job; what it CAN
    run is the whole sharded code path - parameter broadcast, global demand mean, the flat gradient all-reduce, barriers -
    over a ONE-rank RCCL group (NIC_DIST_FORCE_INIT=1).  Same numbers as without a group, and RCCL really was the backend."""
    helper = os.path.join(HERE, "ddp_helper.py")
    single = _run([sys.executable, helper])
    forced = _run([sys.executable, helper], {"NIC_DIST_FORCE_INIT": "1", "MASTER_PORT": "29553"})
    assert forced["backend"] == "nccl" and single["backend"] is None
    assert abs(single["test_loss"] - forced["test_loss"]) <= 1e-6 * abs(single["test_loss"]), (single, forced)


@pytest.mark.parametrize("workload", ["cfg2", "cfg3", "base_stock", "gnn", "gnn_many_warehouses", "real_data_driven"])
def test_bench_over_a_one_rank_rccl_group(workload):
    """`bench.py`'s multi-GPU leg (broadcast, reducer inside the timed step, barrier + MAX-over-ranks clock) on RCCL."""
    root = os.path.dirname(HERE)
    size = ["--scenarios", "48", "--periods", "12"] if workload == "real_data_driven" else ["--scenarios", "2048", "--periods", "12"]
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + size
    outs = []
    # (round 5: the default N = 1 run creates the one-rank group itself; `--no-dist-init` is the run without any process group)
    for more, extra in ((["--no-dist-init"], {}), ([], {"NIC_DIST_FORCE_INIT": "1", "MASTER_PORT": "29557"})):
        r = subprocess.run(cmd + more, capture_output=True, text=True, env=dict(os.environ, **extra), timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    plain, forced = outs
    assert forced["n_gpus"] == 1 and "RCCL" in json.dumps(forced["config"]) and "RCCL" not in json.dumps(plain["config"])
    assert forced["collective"]["backend"] == "nccl" and plain["collective"]["backend"] is None
    a, b = (o["config"]["mean_cost_per_store_period"] for o in outs)
    assert abs(a - b) <= 1e-6 * abs(a), (a, b)


@pytest.mark.parametrize("workload", ["cfg3", "cfg2", "base_stock"])
def test_bench_line_contract(workload):
    """`python bench.py` prints ONE JSON line with the driver's keys; value = units / time, the roofline object is internally
    consistent (frac = achieved / peak, bound-specific unit), and the CPU baseline leg ran on a bounded sample."""
    root = os.path.dirname(HERE)
    small = {"cfg3": ["--scenarios", "4096", "--periods", "10"], "cfg2": ["--scenarios", "8192", "--periods", "20"],
             "base_stock": ["--scenarios", "8192", "--periods", "20"]}[workload]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "256"] + small, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    cfg = out["config"]
    assert "workload" in cfg and "model" not in cfg
    units = cfg["global_scenarios"] * cfg["stores"] * cfg["periods"] * out["steps"]
    assert abs(out["value"] - units / (out["ms_per_step"] * 1e-3 * out["steps"])) <= 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == ("GB/s" if rf["bound"] == "hbm" else "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 2e-3 and 0 < rf["frac"] < 1 and "kernel" in rf and "traffic" in rf
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["host_cores"] >= cb["cores"] and "sample" in cb
    assert cb["unit"] == out["unit"]


def test_bench_strong_and_weak_scaling_over_ranks_sharing_the_gpu():
    """`bench.py --gpus N --scaling strong|weak` under a launcher (round 4; ranks share the test box's one GPU, collectives over
    gloo - timings are meaningless, the SHARDING is what is checked): strong = the scenario count in total, split over the ranks;
    weak = per rank.  The first step's mean cost depends only on the global scenario set (sharded generation is keyed by the
    global scenario index), so 4 x 4,096 strong == 2 x 8,192 strong == 1 x 16,384, and weak 4 x 4,096 == the same 16,384."""
    root = os.path.dirname(HERE)

    def run(n, extra):
        cmd = [sys.executable]
        if n > 1:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                    "--master-port", str(29560 + n)]
        cmd += [os.path.join(root, "bench.py"), "--gpus", str(n), "--workload", "cfg3", "--periods", "12", "--steps", "1", "--warmup", "0",
                "--no-cpu-baseline", "--no-kernel-timing"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, NIC_DIST_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = run(1, ["--scenarios", "16384"])
    outs = {"strong2": run(2, ["--scenarios", "16384", "--scaling", "strong"]),
            "strong4": run(4, ["--scenarios", "16384", "--scaling", "strong"]),
            "weak4": run(4, ["--scenarios", "4096"])}
    for k, o in outs.items():
        n = int(k[-1])
        assert o["n_gpus"] == n and o["scaling"] == k[:-1] and o["config"]["global_scenarios"] == 16384
        assert o["config"]["scenarios_per_gpu"] == 16384 // n
        a, b = o["config"]["mean_cost_per_store_period"], one["config"]["mean_cost_per_store_period"]
        assert abs(a - b) <= 1e-6 * abs(b), (k, a, b)
        assert abs(o["value"] - 16384 * 16 * 12 / (o["ms_per_step"] * 1e-3)) <= 1e-6 * o["value"]


def test_bench_epoch_workload_line():
    """`bench.py --workload cfg3_yaml`: the reference's shipped YAML pair through Trainer.do_one_epoch (8,192 samples, batches of
    1,024 x 50 periods).  One JSON line with the driver's keys; a step is one batch; the epoch is also reported eager / replayed
    and with the reference-style host DataLoader."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg3_yaml", "--steps", "8", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_epoch", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels"):
        assert k in out, k
    cfg = out["config"]
    assert cfg["samples"] == 8192 and cfg["batch_size"] == 1024 and cfg["batches_per_epoch"] == 8 and cfg["periods"] == 50
    assert cfg["stores"] == 5 and cfg["route"] == "FusedRollout" and out["steps"] == 8
    assert abs(out["value"] - 8192 * 5 * 50 / (out["ms_per_epoch"] * 1e-3)) <= 1e-6 * out["value"]
    assert abs(out["ms_per_step"] * 8 - out["ms_per_epoch"]) <= 1e-6 * out["ms_per_epoch"]
    ev = cfg["epoch_variants"]
    assert ev["eager_ms_per_epoch"] > 0 and ev["graph_ms_per_epoch"] > 0 and ev["torch_dataloader_ms_per_epoch"] > ev["graph_ms_per_epoch"]
    assert cfg["rollout_graph"]["setting"] == "auto" and cfg["rollout_graph"]["auto_probe"] is not None
    rf = out["roofline"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 2e-3 and 0 < rf["frac"] < 1


def test_bench_real_data_epoch_workload_line():
    """`bench.py --workload real_data_yaml`: the shipped real-data YAML pair (stand-in files) through Trainer.do_one_epoch - 288
    products in 4 batches of 72 x 95 weeks on the whole-horizon kernels, with the same epoch on the per-period kernels (eager and
    replayed) reported beside it."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "real_data_yaml", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    cfg = out["config"]
    assert cfg["samples"] == 288 and cfg["batch_size"] == 72 and cfg["batches_per_epoch"] == 4 and cfg["periods"] == 95
    assert cfg["stores"] == 21 and cfg["route"] == "FusedRollout (whole-horizon kernels)"
    assert abs(out["value"] - 288 * 21 * 95 / (out["ms_per_epoch"] * 1e-3)) <= 1e-6 * out["value"]
    ev = cfg["epoch_variants"]
    # one launch per direction instead of ~1,050: well below the per-period kernels however they are launched
    assert out["ms_per_epoch"] < 0.6 * ev["per_period_route_graph_ms_per_epoch"] < ev["per_period_route_eager_ms_per_epoch"]
    assert {"horizon_fwd", "horizon_bwd"} <= set(out["kernels"])
    assert out["kernels"]["horizon_fwd"]["launches_per_step"] == 1.0 and out["kernels"]["horizon_bwd"]["launches_per_step"] == 1.0


def test_bench_one_store_real_data_quantile_epoch_workload_line():
    """`bench.py --workload one_store_real_transformed_nv_yaml`: the shipped one-store real-data YAML with `transformed_nv.yml`
    (stand-in files and forecaster weights) through Trainer.do_one_epoch - 4 batches of 8,192 series x 95 weeks on the tape route,
    training steps replayed from a HIP graph; the reference-style loop (Simulator.step per period) reported beside it."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "one_store_real_transformed_nv_yaml", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    cfg = out["config"]
    assert cfg["samples"] == 32768 and cfg["batch_size"] == 8192 and cfg["batches_per_epoch"] == 4 and cfg["periods"] == 95
    assert cfg["stores"] == 1 and cfg["route"] == "TapeRollout" and cfg["step_graph"] == {"setting": "auto", "captured": True}
    ev = cfg["epoch_variants"]
    assert out["ms_per_epoch"] < 0.2 * ev["generic_route_ms_per_epoch"]
    assert {"horizon_fwd", "horizon_bwd"} <= set(out["kernels"])


def test_library_mapped_before_any_torch_device_use_still_launches():
    """build() and smoke() in one process: the C-ABI library is mapped (and its code objects registered with the HIP runtime)
    before PyTorch has touched the device.  `load_library` initialises torch's device first; without that every launch from
    the library failed with "no ROCm-capable device is detected"."""
    root = os.path.dirname(HERE)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from neural_inventory_control_amd import _lib\n"
            "_lib.load_library()\n"
            "import torch\n"
            "from neural_inventory_control_amd import ops\n"
            "y = torch.full((3, 64), 2.4, device='cuda'); y[1] = 3.5\n"
            "ops.round_orders(y, 64)\n"
            "torch.cuda.synchronize(); assert float(y.sum()) == 64 * (2.0 + 4.0 + 2.0); print('LAUNCH_OK')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "LAUNCH_OK" in r.stdout, r.stderr[-2000:]


def test_bench_two_ranks_under_real_rccl_describes_its_collective():
    """`bench.py --gpus 2 --workload cfg4` under the real `nccl` (= RCCL) backend: two ranks on two distinct devices, one flat
    gradient all-reduce per step, described in the line's `collective` object.  Needs two GPUs: skipped on a one-GPU box."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cfg4", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["collective"]
    assert line["n_gpus"] == 2 and c["backend"] == "nccl" and c["world_size"] == 2 and c["ranks_seen"] == 2
    assert c["distinct_devices"] == 2 and c["allreduces_per_step"] == 1 and c["allreduce_bytes"] > 0 and c["allreduce_ms"] > 0
    assert line["config"]["global_scenarios"] == 2 * line["config"]["scenarios_per_gpu"]


def test_bench_single_rank_line_carries_the_collective_of_a_one_rank_rccl_group():
    """The default N = 1 run creates a one-rank RCCL group (NIC_DIST_FORCE_INIT), so the sharded path's all-reduce really runs
    and the line says which backend, how many bytes and how long."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg4", "--scenarios", "2048", "--periods", "10",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["distinct_devices"] == 1, c
    assert c["allreduces_per_step"] == 1 and c["allreduce_bytes"] == 4 * (1700 + 2) and c["allreduce_ms"] is not None, c
