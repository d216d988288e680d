"""EXPERIMENT: does running the two halves of a small batch as two HIP graphs on two streams beat one engine on the whole batch?
(Launch gaps and the ramp-up / ramp-down of every launch are idle time on one stream; a second stream's kernels can fill them.)
    python tools/experiments/split_streams_probe.py [scenarios] [periods]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def main():
    import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda")
    res = {"scenarios": n, "periods": T}

    def make(nn, graph):
        setting, policy, sc, data, model, eng, nn2, T2, desc = bench.build_case("cfg3", dev, 0, 1, nn, T, False)
        eng.materialize(eng.input_rows(data, setting["observation_params"]))
        eng.use_graph = graph

        def run():
            return eng.run(data, T, 0, train=True, observation_params=setting["observation_params"], demand_soa=sc.demands_soa)
        return run, eng

    def timeit(fn, reps=5):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    whole_eager, _ = make(n, False)
    res["one_engine_eager_ms"] = round(timeit(whole_eager), 3)
    whole_graph, _ = make(n, True)
    res["one_engine_graph_ms"] = round(timeit(whole_graph), 3)
    a, ea = make(n // 2, True)
    b, eb = make(n // 2, True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both_serial():
        a()
        b()
    res["two_halves_one_stream_ms"] = round(timeit(both_serial), 3)

    def both_streams():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a()
        with torch.cuda.stream(s2):
            b()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    # graphs were captured on the default stream during the warm-up of `both_serial`; replaying them from other streams is allowed
    res["two_halves_two_streams_ms"] = round(timeit(both_streams), 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
