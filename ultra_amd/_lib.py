"""ctypes binding of libultra_amd.so (include/ultra_rspmm.h).

PyTorch only supplies device memory (tensor.data_ptr()) and the current HIP stream; every kernel is
ours.  There is NO fallback: if the shared library is missing or fails to load, importing this
module raises, and CPU tensors are rejected by the callers.
"""
import ctypes
import os

# PyTorch-ROCm ships its own HIP runtime; it must be the one already resident when libultra_amd.so resolves
# libamdhip64 (a second runtime in the process sees "no ROCm-capable device").  Hence torch first.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# ULTRA_AMD_LIB: another build of the same sources (kernel A/B measurements, tools/build_variant.py)
LIB_PATH = os.environ.get("ULTRA_AMD_LIB") or os.path.join(HERE, "lib", "libultra_amd.so")

ULTRA_OK = 0
ULTRA_ERR_INVALID = 1
ULTRA_ERR_UNSORTED = 2
ULTRA_ERR_HIP = 3
ULTRA_ERR_UNSUPPORTED = 4

SUM_CODES = {"add": 0, "min": 1, "max": 2}
MUL_CODES = {"mul": 0, "add": 1}
F32, F64 = 0, 1

ARR_ROW_PTR, ARR_COL, ARR_TYPE, ARR_PERM, ARR_ITEM, ARR_SPLIT_ROW, ARR_SPLIT_PTR = range(7)
PLAN_EXACT_ORDER = 1
PLAN_TYPE_RUNS = 2
PLAN_DENSE = 4
LAYER_REFERENCE_ORDER = 8
DENSE_MAX_IN_ROW = 1024
ARR_DENSE = 7
ARR_DENSE_ORDER = 8


class UltraMat(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("n_outer", ctypes.c_int64), ("stride_outer", ctypes.c_int64),
                ("n_row", ctypes.c_int64), ("stride_row", ctypes.c_int64), ("row_len", ctypes.c_int64)]


class PlanOpts(ctypes.Structure):
    _fields_ = [("seg_len", ctypes.c_int32), ("g_max", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class PlanInfo(ctypes.Structure):
    _fields_ = [("num_edge", ctypes.c_int64), ("num_node", ctypes.c_int64), ("num_relation", ctypes.c_int64),
                ("n_item", ctypes.c_int64), ("n_wave_item", ctypes.c_int64), ("n_group_item", ctypes.c_int64),
                ("n_unit", ctypes.c_int64), ("n_split_row", ctypes.c_int64), ("n_partial_slot", ctypes.c_int64),
                ("seg_len", ctypes.c_int32), ("g_max", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("packed", ctypes.c_int32), ("on_device", ctypes.c_int32), ("has_transpose", ctypes.c_int32),
                ("n_type_run", ctypes.c_int64), ("dense_bytes", ctypes.c_int64), ("n_chain_row", ctypes.c_int64),
                ("dense_order_bytes", ctypes.c_int64)]


class ScheduleInfo(ctypes.Structure):
    _fields_ = [("nparts", ctypes.c_int32), ("reserved", ctypes.c_int32), ("n_chunk", ctypes.c_int64),
                ("n_unit", ctypes.c_int64), ("max_chunk_per_part", ctypes.c_int64), ("max_unit_per_part", ctypes.c_int64),
                ("max_cost", ctypes.c_double), ("mean_cost", ctypes.c_double)]


class Tuning(ctypes.Structure):
    _fields_ = [("threads", ctypes.c_int32), ("grid", ctypes.c_int32), ("rel_lds", ctypes.c_int32),
                ("x_lds", ctypes.c_int32), ("unroll", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ultra_amd: %s is missing. Build it with `python -m ultra_amd.build` (needs hipcc); "
            "there is no CPU or PyTorch fallback for the rspmm engine." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ultra_last_error.restype = ctypes.c_char_p
    lib.ultra_abi_version.restype = ctypes.c_int32
    lib.ultra_device_count.restype = ctypes.c_int32
    lib.ultra_device_error.restype = ctypes.c_int32
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    matp = ctypes.POINTER(UltraMat)
    lib.ultra_plan_create.argtypes = [ctypes.POINTER(vp), vp, vp, i64, i64, i64, i64, ctypes.POINTER(PlanOpts)]
    lib.ultra_plan_upload.argtypes = [vp]
    lib.ultra_plan_destroy.argtypes = [vp]
    lib.ultra_plan_pin.argtypes = [vp, i32]
    lib.ultra_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    lib.ultra_plan_export.argtypes = [vp, i32, vp, i64, ctypes.POINTER(i64)]
    lib.ultra_rspmm_forward.argtypes = [vp, i32, i32, i32, vp, matp, matp, matp, matp, vp]
    lib.ultra_rspmm_forward_masked.argtypes = [vp, i32, i32, i32, vp, matp, matp, matp, matp, vp]
    lib.ultra_rspmm_forward_onehot.argtypes = [vp, i32, vp, matp, matp, vp, matp, matp, vp]
    lib.ultra_rspmm_forward_point.argtypes = [vp, i32, i32, i32, vp, matp, matp, vp, matp, matp, vp]
    lib.ultra_rspmm_forward_update.argtypes = [vp, i32, i32, matp, matp, vp, matp, matp, vp, vp, vp, vp, ctypes.c_float, i32, matp, vp]
    lib.ultra_rspmm_forward_update_timed.argtypes = [vp, i32, i32, matp, matp, vp, matp, matp, vp, vp, vp, vp, ctypes.c_float, i32, matp, vp,
                                                     i32, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.ultra_nbf_dense_layer.argtypes = [vp, matp, matp, matp, vp, vp, vp, vp, vp, ctypes.c_float, i32, matp, vp]
    lib.ultra_nbf_layer0.argtypes = [vp, vp, matp, vp, vp, vp, vp, vp, vp, ctypes.c_float, i32, matp, vp]
    lib.ultra_rspmm_backward.argtypes = [vp, i32, i32, i32, vp, matp, matp, matp, matp, vp, matp, matp, vp]
    lib.ultra_rspmm_backward_add.argtypes = [vp, i32, i32, i32, vp, matp, matp, matp, matp, vp, matp, matp, matp, vp]
    lib.ultra_rspmm_dense_relation_grad.argtypes = [vp, matp, matp, matp, vp]
    lib.ultra_rspmm_rows_forward.argtypes = [vp, i32, vp, matp, matp, vp, i64, matp, vp, vp, vp, vp]
    lib.ultra_rspmm_rows_backward.argtypes = [vp, i32, vp, matp, matp, vp, i64, vp, matp, matp, vp]
    lib.ultra_rspmm_rows_backward_gather.argtypes = [vp, i32, vp, matp, matp, vp, i64, vp, vp, vp, vp, matp, matp, vp]
    lib.ultra_rspmm_weight_epoch.argtypes = [i64]
    lib.ultra_rspmm_onehot_backward.argtypes = [vp, vp, vp, vp, vp, matp, vp, vp, matp, vp, vp, vp]
    lib.ultra_rspmm_forward_timed.argtypes = [vp, i32, i32, i32, vp, matp, matp, matp, vp, matp, vp, i32, i32,
                                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.ultra_conv_update.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, ctypes.c_float, i32, vp]
    lib.ultra_conv_update_backward_workspace.argtypes = [i64]
    lib.ultra_conv_update_backward_workspace.restype = i64
    lib.ultra_conv_update_backward.argtypes = [vp] * 14 + [i64, i64, i32, i32, ctypes.c_float, i32, vp]
    lib.ultra_edge_keep_mask.argtypes = [vp, vp, vp, i64, vp, i64, i64, i64, vp, vp]
    lib.ultra_easy_edge_keep.argtypes = [vp, vp, vp, i64, vp, vp, vp, i64, i64, i64, i64, i64, vp, vp]
    lib.ultra_readout.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, i32, i32, vp]
    lib.ultra_stream_copy.argtypes = [vp, vp, i64, vp]
    lib.ultra_filtered_rank.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp]
    lib.ultra_strict_negatives.argtypes = [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp]
    lib.ultra_ranking_loss.argtypes = [vp, i64, i64, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    lib.ultra_readout_train_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]
    lib.ultra_readout_train_backward_workspace.argtypes = [i64, i64]
    lib.ultra_readout_train_backward_workspace.restype = i64
    lib.ultra_readout_train_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]
    lib.ultra_onehot_rows.argtypes = [vp, vp, vp, i64, i64, i64, vp]
    lib.ultra_batch_prologue.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.ultra_batch_prologue_rows.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.ultra_readout_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, i32, i32, vp]
    lib.ultra_query_boundary.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp]
    lib.ultra_relation_projection.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    pp = ctypes.POINTER(vp)
    lib.ultra_relation_projection_layers.argtypes = [vp, pp, pp, pp, pp, vp, i64, i32, i32, vp]
    lib.ultra_relation_projection_backward_workspace.argtypes = [i64, i32]
    lib.ultra_relation_projection_backward_workspace.restype = i64
    lib.ultra_relation_projection_backward.argtypes = [vp, pp, pp, pp, pp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]
    lib.ultra_plan_schedule_info.argtypes = [vp, i32, ctypes.POINTER(ScheduleInfo)]
    lib.ultra_order_trace.argtypes = [vp]
    lib.ultra_plan_schedule_export.argtypes = [vp, i32, i32, vp, i64, ctypes.POINTER(i64)]
    lib.ultra_relation_graph_bits.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.ultra_relation_graph_emit.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    lib.ultra_relation_graph_dense_adjacency.argtypes = [vp, i64, vp, vp]
    lib.ultra_set_tuning.argtypes = [ctypes.POINTER(Tuning)]
    lib.ultra_get_tuning.argtypes = [ctypes.POINTER(Tuning)]
    for s in ("add", "min", "max"):
        for m in ("mul", "add"):
            f = getattr(lib, "ultra_rspmm_%s_%s_forward_cuda" % (s, m))
            f.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]
            b = getattr(lib, "ultra_rspmm_%s_%s_backward_cuda" % (s, m))
            b.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]
    return lib


lib = _load()
if lib.ultra_abi_version() != 7:
    raise ImportError("ultra_amd: libultra_amd.so ABI version mismatch")


class UltraError(RuntimeError):
    pass


def check(rc):
    """Map ultra_status to the exceptions the reference raises for the same condition."""
    if rc == ULTRA_OK:
        return
    msg = lib.ultra_last_error().decode("utf-8", "replace")
    if rc == ULTRA_ERR_UNSORTED:
        raise AssertionError(msg)           # rspmm.py:18
    raise UltraError(msg)                    # c10::Error -> RuntimeError in the reference
