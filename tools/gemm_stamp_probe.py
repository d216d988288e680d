"""Where does a 512 x 512 policy-GEMM launch spend its time, and at what clock?  Compiles its own copy of the library with
-DNIC_TUNING_BUILD (in-kernel timestamps: s_memtime + the 100 MHz wall clock at kernel entry, in front of the k loop, behind it
and at exit, per workgroup), runs the forward / dgrad kernels at BASELINE cfg3's shape and prints per phase the mean duration
and the s_memtime rate.  PROBE_TUNES=0,1,2,3,4,7 additionally runs timing-only variants of the k loop (results invalid):
1 = no A-tile copies inside the loop, 2 = no B-tile copies, 4 = no barrier per k tile; 32 = s_setprio 1 for the younger half of
the workgroup (valid results); 8 = dgrad without the Hprev prefetch (valid), 16 = the k-loop stamp moved behind the plain tiles
(the phase then covers the four prefetch tiles of dgrad / nothing in the forward kernel).  PROBE_PADS=0,64,... adds floats to the scenario stride of every operand; PROBE_SIZES=256,2048,... sets the scenario counts
(256 = one workgroup column: what prologue / epilogue cost when no other CU is bursting).  The product library has no such instrumentation."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_inventory_control_amd import _lib, ops
from neural_inventory_control_amd.layout import pad_ld
from gemm_probe import _tuning_library


def run_case(lib, name, fn, n_wg, K, stamps):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    assert lib.nic_tuning_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    lib.nic_tuning_set_stamps(None)
    # un-instrumented timing of the same launch (mean of 50)
    s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s2.record()
    for _ in range(50):
        fn()
    e2.record()
    torch.cuda.synchronize()
    st = stamps.view(n_wg, 4, 2).double().cpu()
    t0 = st[:, 0, 1].min()
    ph = {}
    for i, nm in enumerate(("prologue", "k_loop", "epilogue")):
        wall = (st[:, i + 1, 1] - st[:, i, 1]) / 100.0          # us
        cyc = st[:, i + 1, 0] - st[:, i, 0]
        ph[nm] = {"us_mean": round(float(wall.mean()), 2), "us_max": round(float(wall.max()), 2),
                  "memtime_ticks_per_us": round(float((cyc / wall.clamp_min(1e-9)).mean()), 1)}
    end = (st[:, 3, 1] - t0) / 100.0
    start = (st[:, 0, 1] - t0) / 100.0
    first = start < 5.0   # workgroups of the first round (dispatched at launch)
    kl = (st[:, 2, 1] - st[:, 1, 1]) / 100.0
    rate = (st[:, 2, 0] - st[:, 1, 0]) / kl.clamp_min(1e-9)
    rounds = {}
    for nm, m in (("round1", first), ("later", ~first)):
        if int(m.sum()):
            q = torch.quantile(kl[m], torch.tensor([0.0, 0.5, 1.0], dtype=kl.dtype))
            rounds[nm] = {"n": int(m.sum()), "k_loop_us_min_med_max": [round(float(v), 1) for v in q],
                          "memtime_ticks_per_us": round(float(rate[m].mean()), 1),
                          "start_us_min_max": [round(float(start[m].min()), 1), round(float(start[m].max()), 1)]}
    # MFMA issue cycles of ONE workgroup's k loop per SIMD: waves per SIMD x (K / 2) steps x MFMA tiles per wave x 64 cycles
    # (256 x 256 tile: 2 waves x 8 tiles; 128 x 128, 8 waves: 2 waves x 2 tiles - and a co-resident workgroup issues as much again)
    issue_cycles = 2 * (K // 2) * int(os.environ.get("PROBE_TILES_PER_WAVE", "2")) * 64
    ph["k_loop"]["mfma_issue_cycles_per_us"] = round(issue_cycles / float(((st[:, 2, 1] - st[:, 1, 1]) / 100.0).mean()), 1)
    return {"event_us_stamped": round(s.elapsed_time(e) * 1e3, 1), "event_us": round(s2.elapsed_time(e2) * 1e3 / 50, 1),
            "workgroups": n_wg, "phases": ph, "rounds": rounds, "last_exit_us": round(float(end.max()), 1)}


def main():
    lib = _lib._lib = _lib.load_library(_tuning_library())
    lib.nic_tuning_set_stamps.argtypes = [ctypes.c_void_p]
    dev = "cuda"
    N = K = 512
    out = {}
    tunes = [int(v) for v in os.environ.get("PROBE_TUNES", "0").split(",")]
    for tune in tunes:
        os.environ["NIC_GEMM_TUNE"] = str(tune)
        pads = [int(v) for v in os.environ.get("PROBE_PADS", "0").split(",")]
        sizes = tuple(int(v) for v in os.environ["PROBE_SIZES"].split(",")) if os.environ.get("PROBE_SIZES") else \
            ((65536, 32768, 98304) if tune == 0 and len(pads) == 1 and not os.environ.get('PROBE_ONE_SIZE') else (65536,))
        for B, pad in [(b_, p_) for b_ in sizes for p_ in pads]:
            ldb = pad_ld(B) + pad
            W = torch.randn(N, K, device=dev) * 0.05
            Wt = W.t().contiguous()
            b = torch.randn(N, device=dev)
            X = torch.nn.functional.elu(torch.randn(K, ldb, device=dev))
            Y = torch.zeros(N, ldb, device=dev)
            dX = torch.zeros(K, ldb, device=dev)
            # workgroup tile of the launch (round 4: 128 x 128 with two workgroups co-resident per CU; 256 x 256 with NIC_GEMM_TUNE & 128)
            tr_, tc_ = (256, 256) if (tune & 128) else tuple(int(v) for v in os.environ.get("PROBE_TILE", "128,128").split(","))
            n_wg = (N // tr_) * (B // tc_)
            stamps = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
            cases = [("fwd", lambda: ops.linear_fwd(W, b, X, Y, B, 1))]
            if tune in (0, 8, 16):
                cases.append(("dgrad", lambda: ops.linear_dgrad(Wt, Y, X, dX, B, 1, False)))
            for name, fn in cases:
                r = run_case(lib, name, fn, n_wg, K, stamps)
                r["tflops"] = round(2.0 * N * K * B / (r["event_us"] * 1e-6) / 1e12, 1)
                out[f"{name}_{B}_pad{pad}_tune{tune}"] = r
                print(f"{name}_{B}_pad{pad}_tune{tune}", "event", r["event_us"], "tflops", r["tflops"], "| pro", r["phases"]["prologue"]["us_mean"], "| kloop", r["phases"]["k_loop"], "| epi",
                      r["phases"]["epilogue"]["us_mean"], "|", r["rounds"], flush=True)
    if os.environ.get("PROBE_JSON"):
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
