"""The route each engine picks for the benchmark workloads, in ONE table: which engine `bench.build_case` / the Trainer
instantiates and which route its `_setup` selects per batch size.  Thresholds live in the engines; this pins what they amount to."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None

# workload, scenarios per GPU, periods -> (engine class, route)
TABLE = [
    ("cfg1", 256, 6, "FusedRollout", "small"),                      # whole-horizon kernels of the 32-wide policies
    ("cfg2", 4096, 6, "FusedRollout", "small"),
    ("cfg4", 2048, 6, "FusedRollout", "small"),
    ("cfg3", 1024, 4, "FusedRollout", "per-period+tail"),           # fused per-period tail up to 16,384 scenarios
    ("cfg3", 8192, 4, "FusedRollout", "per-period+tail"),
    ("cfg3", 32768, 2, "FusedRollout", "per-period"),               # beyond: the separate bandwidth-efficient launches
    ("cfg5", 1024, 3, "FusedRollout", "per-period"),                # 64 stores per warehouse: outside the tail's shapes
    ("real_data_driven", None, None, "FusedRollout", "horizon"),    # data_driven on the real-data batch: one launch per direction
    ("base_stock", 4096, 6, "ClosedFormRollout", "closed_form_kernel<1,4,false,"),   # compiled-in pipeline length variant
    ("echelon_stock", 2048, 6, "ClosedFormRollout", "closed_form_kernel<4,4,true>"),
    ("gnn", 512, 3, "GnnRollout", "period-kernel+period-bwd"),      # graph fits in LDS: one forward launch per period; one backward launch
    ("gnn_many_warehouses", 256, 3, "GnnRollout", "period-kernel(spill)+period-bwd"),   # 19 nodes + 70 edges: edge tiles in global scratch
]


@pytest.mark.parametrize("workload,n,T,engine,route", TABLE)
def test_route_table(workload, n, T, engine, route):
    import bench
    from neural_inventory_control_amd import _lib
    setting, policy, sc, data, model, eng, n2, T2, desc = bench.build_case(workload, DEV, 0, 1, n, T, False)
    assert type(eng).__name__ == engine
    obs = setting["observation_params"]
    if engine == "GnnRollout":
        eng.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
        eng.run(data, T2, 0, train=True, observation_params=obs, demand_soa=sc.demands_soa)
        got = ("period-kernel" + ("(spill)" if eng.edge_scratch is not None else "") if eng._period else "per-mlp") + \
            ("+period-bwd" if eng._period_bwd else "")
        assert got == route
        return
    if engine == "ClosedFormRollout":
        with torch.no_grad():
            eng.model.closed_form_levels()
        eng.run(data, T2, 0, train=True, observation_params=obs, demand_soa=sc.demands_soa)
        assert (_lib.lib().nic_last_kernel() or b"").decode().startswith(route)
        return
    eng.materialize(eng.input_rows(data, obs))
    eng.run(data, T2, 0, train=True, observation_params=obs, demand_soa=sc.demands_soa)
    got = ("small" if eng.small is not None else "horizon" if getattr(eng, "horizon", None) is not None
           else "per-period+tail" if eng._use_tail() else "per-period")
    assert got == route


@pytest.mark.parametrize("workload,n,T", [("base_stock", 2048, 8), ("echelon_stock", 1024, 6), ("cfg1", 1024, 6), ("gnn", 512, 3)])
def test_a_batch_that_is_already_a_kernel_layout_trace_is_used_in_place(workload, n, T):
    """`data["demands"]` of a device-sampled scenario is a (B, S, T) view of the [T][S][ldb] trace: the engines read it in place
    (`layout.demand_trace_soa`) - same totals and gradients, bit for bit, as the same numbers handed over as a contiguous
    (B, S, T) tensor (which is transposed into a fresh buffer), and as the trace passed explicitly."""
    import bench
    from neural_inventory_control_amd.layout import demand_trace_soa
    setting, policy, sc, data, model, eng, n2, T2, desc = bench.build_case(workload, DEV, 0, 1, n, T, False)
    obs = setting["observation_params"]
    assert demand_trace_soa(data["demands"], sc.demands_soa.shape[2]).data_ptr() == sc.demands_soa.data_ptr()
    plain = dict(data)
    plain["demands"] = data["demands"].contiguous()
    assert demand_trace_soa(plain["demands"], sc.demands_soa.shape[2]).data_ptr() != plain["demands"].data_ptr()
    if type(eng).__name__ == "GnnRollout":
        eng.materialize(max(data["initial_inventories"].shape[2], data["initial_warehouse_inventories"].shape[2]) + 4)
    elif type(eng).__name__ == "ClosedFormRollout":
        with torch.no_grad():
            eng.model.closed_form_levels()
    else:
        eng.materialize(eng.input_rows(data, obs))
    res = []
    for batch, soa in ((data, None), (plain, None), (data, sc.demands_soa)):
        model.zero_grad(set_to_none=True)
        out = eng.run(batch, T2, 0, train=True, observation_params=obs, demand_soa=soa)
        total = out[0]
        if type(eng).__name__ == "ClosedFormRollout":
            total.backward()
            grads = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        else:
            grads = [g.clone() for _, g in eng.param_grads()]
        res.append((float(total), grads))
    for tot, grads in res[1:]:
        assert tot == res[0][0]
        assert all(torch.equal(a, b) for a, b in zip(grads, res[0][1]))
