"""ctypes binding of include/nic_rollout.h (libnic_hip.so).

The shared library is built in-tree by `neural_inventory_control_amd/build.py` (hipcc --offload-arch=gfx950) and
loaded lazily on first use.  There is NO CPU fallback: if the library is missing, or no HIP device is visible,
every compute entry point raises `NicUnavailableError`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnic_hip.so")
NIC_MAX_SLOTS = 16
NIC_ACT_NONE, NIC_ACT_ELU = 0, 1


class NicUnavailableError(RuntimeError):
    pass


class NicError(RuntimeError):
    pass


class NicTable2(C.Structure):
    _fields_ = [("p", C.c_void_p), ("loc_stride", C.c_int64), ("scn_stride", C.c_int64)]


class NicTable3(C.Structure):
    _fields_ = [("p", C.c_void_p), ("loc_stride", C.c_int64), ("sup_stride", C.c_int64), ("scn_stride", C.c_int64)]


class NicEnvDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_scenarios", "ldb", "n_stores", "n_warehouses", "n_echelons", "store_slots", "warehouse_slots",
        "echelon_slots", "lost_demand", "maximize_profit")]


class NicEnvStepIO(C.Structure):
    _fields_ = [
        ("dims", NicEnvDims),
        ("store_inv", C.c_void_p), ("wh_inv", C.c_void_p), ("ech_inv", C.c_void_p),
        ("demand", NicTable2),
        ("store_orders", NicTable3), ("wh_orders", NicTable2), ("ech_orders", NicTable2),
        ("underage", NicTable2), ("holding", NicTable2), ("lead_times", NicTable3),
        ("wh_holding", NicTable2), ("wh_lead_times", NicTable2), ("wh_edge_costs", NicTable2),
        ("ech_holding", NicTable2), ("ech_lead_times", NicTable2),
    ]


class NicPeriodTail(C.Structure):
    _fields_ = [("io", NicEnvStepIO), ("adjacency", C.c_void_p), ("upper_bound", C.c_float), ("transshipment", C.c_int32),
                ("W_out", C.c_void_p), ("ldw_out", C.c_int64), ("b_out", C.c_void_p), ("n_out", C.c_int32), ("K", C.c_int32),
                ("Wt_in", C.c_void_p), ("ldwt_in", C.c_int64), ("N1", C.c_int32)]


class NicWideRollout(C.Structure):
    _fields_ = ([("io", NicEnvStepIO), ("adjacency", C.c_void_p), ("upper_bound", C.c_float), ("transshipment", C.c_int32)]
                + [(n, C.c_int32) for n in ("T", "H", "n_hidden", "n_out")]
                + [("demand", C.c_void_p), ("ps_demand", C.c_int64), ("ld_demand", C.c_int64), ("states", C.c_void_p),
                   ("orders", C.c_void_p), ("logits", C.c_void_p), ("rewards", C.c_void_p), ("hidden", C.c_void_p * 4)]
                + [(n, C.c_int64) for n in ("ps_state", "ps_orders", "ps_logits", "ps_hidden")]
                + [("Wt_in", C.c_void_p), ("ldwt_in", C.c_int64), ("Wp_hidden", C.c_void_p * 4), ("b_hidden", C.c_void_p * 4),
                   ("Wq_out", C.c_void_p), ("b_out", C.c_void_p)])


class NicSmallRolloutDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in (
        "n_scenarios", "ldb", "T", "t0", "F", "n_hidden", "n_out", "head", "Ws", "Wn", "Ww", "E", "We", "lost_demand",
        "maximize_profit", "detach_input", "round_orders")] + [("upper_bound", C.c_float), ("lane_scenarios", C.c_int32),
                                               ("weights", C.c_void_p), ("demand", C.c_void_p),
                                               ("state0", C.c_void_p)]
                + [(n, NicTable2) for n in ("underage", "holding", "lead", "wh_holding", "wh_lead", "wh_edge",
                                            "ech_holding", "ech_lead")])


class NicHorizonDesc(C.Structure):
    _fields_ = ([("io", NicEnvStepIO)] + [(n, C.c_int32) for n in ("T", "t0", "H1", "H2", "n_out", "round_orders")]
                + [("W1", C.c_void_p), ("ldw1", C.c_int64), ("W2", C.c_void_p), ("ldw2", C.c_int64), ("W3", C.c_void_p),
                   ("ldw3", C.c_int64), ("b2", C.c_void_p), ("b3", C.c_void_p), ("mask", C.c_void_p), ("demand", C.c_void_p),
                   ("hist_stride", C.c_int64), ("head_mode", C.c_int32), ("allow_negative", C.c_int32), ("tape", C.c_void_p)])


class NicClosedFormDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in (
        "n_scenarios", "ldb", "T", "t0", "ignore_periods", "policy", "n_levels", "S", "Ws", "Wn", "Ww", "E", "We",
        "lost_demand", "maximize_profit", "round_orders")]
                + [("levels", C.c_void_p), ("demand", C.c_void_p), ("state0", C.c_void_p)]
                + [(n, NicTable2) for n in ("underage", "holding", "lead", "wh_holding", "wh_lead", "wh_edge",
                                            "ech_holding", "ech_lead")])


class NicMlp3Seg(C.Structure):
    _fields_ = [("base", C.c_void_p), ("map", C.c_void_p), ("row_stride", C.c_int64), ("ent_stride", C.c_int64),
                ("scn_stride", C.c_int64), ("n_rows", C.c_int32), ("reserved", C.c_int32)]


class NicSegTerm(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_row_stride", C.c_int64), ("offsets", C.c_void_p), ("items", C.c_void_p),
                ("scale", C.c_void_p)]


class NicMlp3Desc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("n_entities", "n_scenarios", "ldb", "K", "n_out", "out_act", "n_segs", "hist_native")]
                + [("seg", NicMlp3Seg * 4), ("weights", C.c_void_p), ("weights_t", C.c_void_p), ("hist_row_stride", C.c_int64),
                   ("ent_row_stride", C.c_int64)])


class NicGnnPeriodMlp(C.Structure):
    _fields_ = [("wpk", C.c_void_p), ("Y", C.c_void_p), ("Ysum", C.c_void_p), ("H1", C.c_void_p), ("H2", C.c_void_p),
                ("row_stride", C.c_int64)]


class NicGnnPeriod(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("n_nodes", "n_edges", "n_live", "n_scenarios", "ldb", "Dn", "max_inv", "store_feat",
                                          "fuse_env", "e_self", "e_supplier", "cap_at_one", "n_agg_items", "wb0_floats", "wb1_floats", "tab_words")]
                + [(n, C.c_void_p) for n in ("src", "tgt", "agg_off", "agg_items", "agg_scale", "lead", "node_row0", "node_slots",
                                             "state", "feat", "agg")]
                + [("mlp", NicGnnPeriodMlp * 5), ("io", NicEnvStepIO)]
                + [(n, C.c_void_p) for n in ("orders", "sums", "ratio", "scale", "store_out", "wh_out", "reward", "edge_scratch")])


class NicGnnPeriodBwdMlp(C.Structure):
    _fields_ = [("wpk_t", C.c_void_p), ("Y", C.c_void_p), ("H1", C.c_void_p), ("H2", C.c_void_p), ("row_stride", C.c_int64),
                ("slab1", C.c_void_p), ("lds1", C.c_int64), ("slab2", C.c_void_p), ("lds2", C.c_int64),
                ("slab3", C.c_void_p), ("lds3", C.c_int64)]


class NicGnnPeriodBwd(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("n_nodes", "n_edges", "n_live", "n_scenarios", "ldb", "Dn", "n_sub", "reserved")]
                + [("n_items", C.c_int32 * 4)]
                + [(n, C.c_void_p) for n in ("src", "tgt", "lead", "node_row0", "node_slots", "agg_scale")]
                + [("list_off", C.c_void_p * 4), ("list_items", C.c_void_p * 4)]
                + [(n, C.c_void_p) for n in ("feat", "nodes0", "nodes1", "edges0", "edges1", "agg")]
                + [("node_row_stride", C.c_int64), ("edge_row_stride", C.c_int64)]
                + [(n, C.c_void_p) for n in ("d_out", "g_state", "scratch")]
                + [("mlp", NicGnnPeriodBwdMlp * 5)]
                + [(n, C.c_int32) for n in ("fuse_env", "e_self", "e_supplier", "cap_at_one")]
                + [("io", NicEnvStepIO)]
                + [(n, C.c_void_p) for n in ("sums", "ratio", "scale", "g_store_out", "g_wh_out")]
                + [("g_reward", NicTable2)]
                + [(n, C.c_void_p) for n in ("g_store_in", "g_wh_in", "g_orders")])


NIC_MLP3_MAX_K, NIC_MLP3_ACT_NONE, NIC_MLP3_ACT_ELU, NIC_MLP3_ACT_SOFTPLUS = 96, 0, 1, 2
NIC_CF_MAX_LEVELS = 5
NIC_CF_BASE_STOCK, NIC_CF_CAPPED, NIC_CF_ECHELON = 0, 1, 2
NIC_SR_MAX_INPUTS, NIC_SR_HIDDEN, NIC_SR_MAX_OUTPUTS = 16, 32, 8
NIC_THIN_MAX_ROWS = 32
_vp, _i32, _i64, _f32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
_IOP = C.POINTER(NicEnvStepIO)

# name -> (restype, argtypes); every symbol include/nic_rollout.h declares
PROTOTYPES = {
    "nic_abi_version": (C.c_int, []),
    "nic_build_id": (C.c_char_p, []),
    "nic_last_error": (C.c_char_p, []),
    "nic_last_kernel": (C.c_char_p, []),
    "nic_device_count": (C.c_int, []),
    "nic_env_step_fwd": (C.c_int, [_IOP, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "nic_env_step_bwd": (C.c_int, [_IOP, _vp, _vp, _vp, NicTable2, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "nic_head_env_fwd": (C.c_int, [_IOP, _vp, _vp, _vp, _i32, C.c_float, _i32, _vp, _vp, _vp, _vp]),
    "nic_head_env_bwd": (C.c_int, [_IOP, _vp, _vp, _vp, _i32, C.c_float, _i32, _vp, _vp, NicTable2, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nic_period_tail_ok": (C.c_int, [C.POINTER(NicEnvDims), _i32, _i32, _i32]),
    "nic_period_tail_fwd": (C.c_int, [C.POINTER(NicPeriodTail), _vp, _vp, _vp, _vp, _vp, _vp]),
    "nic_period_tail_bwd_slots": (C.c_int, [_i32]),
    "nic_period_tail_bwd": (C.c_int, [C.POINTER(NicPeriodTail), _vp, _vp, _vp, _vp, NicTable2, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "nic_linear_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_linear_fwd_thin_in_ok": (C.c_int, [_i32, _i32]),
    "nic_linear_fwd_thin_in": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_linear_dgrad": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_wgrad_num_splits": (C.c_int, [_i32, _i32, _i32]),
    "nic_wgrad_periods_num_splits": (C.c_int, [_i32, _i32, _i32, _i32]),
    "nic_linear_wgrad": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_linear_wgrad_periods": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp]),
    "nic_linear_bwd_thin": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_wgrad_reduce": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _i32, _i32, _f32, _vp]),
    "nic_head_warehouse_fwd": (C.c_int, [_vp, _vp, _vp, _f32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_head_warehouse_bwd": (C.c_int, [_vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                         _i32, _vp]),
    "nic_head_data_driven_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_head_data_driven_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_head_softplus_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "nic_head_softplus_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "nic_head_serial_fwd": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_head_serial_bwd": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                      _i32, _vp]),
    "nic_small_rollout_fwd": (C.c_int, [C.POINTER(NicSmallRolloutDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "nic_small_rollout_bwd": (C.c_int, [C.POINTER(NicSmallRolloutDesc), _vp, _vp, _vp, NicTable2, _vp, _vp, _vp]),
    "nic_small_rollout_bwd_wgrad_slots": (C.c_int, [C.c_int32]),
    "nic_small_rollout_bwd_wgrad": (C.c_int, [C.POINTER(NicSmallRolloutDesc), _vp, _vp, _vp, NicTable2, _vp, _i64, _vp]),
    "nic_small_rollout_reduce_scratch": (C.c_int, [C.c_int32, C.c_int32, _i64]),
    "nic_small_rollout_reduce": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "nic_horizon_rollout_ok": (C.c_int, [C.POINTER(NicHorizonDesc)]),
    "nic_horizon_rollout_fwd": (C.c_int, [C.POINTER(NicHorizonDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nic_horizon_rollout_bwd": (C.c_int, [C.POINTER(NicHorizonDesc), _vp, _vp, _vp, _vp, _vp, NicTable2, _vp, _vp, _vp, _vp]),
    "nic_gnn_alloc_env_fwd": (C.c_int, [_IOP, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "nic_gnn_alloc_env_bwd": (C.c_int, [_IOP, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, NicTable2, _vp, _vp, _vp, _vp, _vp]),
    "nic_gnn_period_pack_size": (C.c_int, [_i32, _i32]),
    "nic_gnn_period_ok": (C.c_int, [_i32, _i32, _i32]),
    "nic_gnn_period_fwd": (C.c_int, [C.POINTER(NicGnnPeriod), _vp]),
    "nic_gnn_period_bwd_scratch_floats": (C.c_int64, [_i32, _i32, _i32, _i32, _i32]),
    "nic_gnn_period_bwd": (C.c_int, [C.POINTER(NicGnnPeriodBwd), _vp]),
    "nic_round_orders": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "nic_mlp3_fwd": (C.c_int, [C.POINTER(NicMlp3Desc), _vp, _vp, _vp, _vp, _vp]),
    "nic_mlp3_fwd_residual": (C.c_int, [C.POINTER(NicMlp3Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nic_mlp3_bwd_fused_slots": (C.c_int, []),
    "nic_mlp3_bwd_fused": (C.c_int, [C.POINTER(NicMlp3Desc), _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "nic_mlp3_bwd_hist_slots": (C.c_int, []),
    "nic_mlp3_bwd_hist": (C.c_int, [C.POINTER(NicMlp3Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "nic_segment_sum_terms": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_gnn_alloc_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_gnn_alloc_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_gnn_alloc_groups_fwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "nic_gnn_alloc_groups_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_segment_sum": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "nic_closed_form_num_partials": (C.c_int, [_i32, _i32]),
    "nic_closed_form_rollout_sums": (C.c_int, [C.POINTER(NicClosedFormDesc), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "nic_sample_demand": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i64, _u64, _i32, _vp, _vp, _i32, _vp]),
    "nic_sample_demand_equicorrelated": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i64, _u64, _vp, _vp, _f32, _i32, _vp]),
}

# include/nic_experiments.h: present only in a library built with NIC_BUILD_EXPERIMENTS=1 (routes that lost their A/B)
EXPERIMENTAL_PROTOTYPES = {
    "nic_wide_rollout_ok": (C.c_int, [C.POINTER(NicEnvDims), _i32, _i32, _i32]),
    "nic_wide_rollout_fwd": (C.c_int, [C.POINTER(NicWideRollout), _vp]),
    "nic_wide_rollout_bwd": (C.c_int, [C.POINTER(NicWideRollout), NicTable2, C.POINTER(C.c_void_p * 4), _i64, _vp, _i64,
                                       C.POINTER(C.c_void_p * 4), _vp, _vp, _vp]),
}

_lib = None


def has_experiments():
    """True if the loaded library carries the experimental entry points (built with NIC_BUILD_EXPERIMENTS=1)."""
    try:
        getattr(lib(), "nic_wide_rollout_fwd")
        return True
    except AttributeError:
        return False


def library_built():
    return os.path.isfile(LIB_PATH)


def _init_torch_device_first():
    """PyTorch's HIP context must exist BEFORE this library is mapped: loading it registers its code objects with the HIP
    runtime, and when that happens ahead of torch's own (lazy) device initialisation every later launch from this library fails
    with "no ROCm-capable device is detected" while torch itself keeps working (measured: build() followed by smoke() in one
    process).  No-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def load_library(path=None):
    """dlopen the C-ABI library and attach prototypes.  Works without a GPU (symbols only)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise NicUnavailableError(
            f"{p} not found: build the HIP extension first (python -m neural_inventory_control_amd.build). "
            "This package has no CPU fallback.")
    _init_torch_device_first()
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENTAL_PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    if lib.nic_abi_version() != 1:
        raise NicError("ABI version mismatch")
    check_build_id(lib, os.path.dirname(p))
    if path is None:
        _lib = lib
    return lib


def check_build_id(lib, csrc_dir):
    """The library must have been built from the sources that lie next to it: `nic_build_id()` (baked in by build.py) against
    the hash of those files now.  A stale binary - a checkout or an edited header after the last build - raises here, at load
    time, instead of running an old kernel until some golden test fails.  Skipped when the sources are not there (an installed
    binary without its tree)."""
    from . import build as _build
    try:
        want = _build.source_id(csrc_dir)
    except OSError:
        return None
    got = (lib.nic_build_id() or b"").decode()
    if got != want:
        raise NicError(f"{os.path.join(csrc_dir, 'libnic_hip.so')} was built from other sources (library id {got}, sources {want}): "
                       "run `python -m neural_inventory_control_amd.build`")
    return got


def lib():
    return load_library()


def check(status):
    if status != 0:
        msg = lib().nic_last_error()
        raise NicError(msg.decode() if msg else f"nic call failed with status {status}")


def require_device():
    import torch
    if not torch.cuda.is_available():
        raise NicUnavailableError("no HIP device visible: the inventory-rollout engine only runs on the GPU "
                                  "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")


def ptr(t):
    """device pointer of a tensor (or None)"""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def table2(t, loc_stride, scn_stride):
    return NicTable2(ptr(t), loc_stride, scn_stride)


def table3(t, loc_stride, sup_stride, scn_stride):
    return NicTable3(ptr(t), loc_stride, sup_stride, scn_stride)
