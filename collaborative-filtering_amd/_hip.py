"""ctypes binding of the C-ABI library (include/als_hip.h).

The library is built in-tree by `__graft_entry__.build()` /
`make -C collaborative-filtering_amd/csrc`.  There is no CPU fallback: if the
library is missing or a symbol is absent, import of the binding raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALS_HIP_LIB: an alternative build of the library (profiling / A-B experiments: profiles/ab_builds.sh)
LIB_PATH = os.environ.get("ALS_HIP_LIB") or os.path.join(_HERE, "csrc", "libals_hip.so")

SPLIT_CHUNK = 4096          # ALS_SPLIT_CHUNK
MAX_K = 160                 # ALS_MAX_K

EXPORTS = ("als_version", "als_padded_k", "als_perm_index", "als_partial_slot_bytes", "als_partial_slot_bytes_f64",
           "als_row_solve", "als_factor_scale", "als_gs_sweep", "als_gs_sweep_levels", "als_gs_sweep_dataflow", "als_residual_stats", "als_w_normal_equations", "als_spd_solve_workspace_bytes", "als_spd_solve_f64", "als_item_stats", "als_item_stats_f64", "als_sum_pairs", "als_sumsq_partials",
           "als_sumsq", "als_history_row", "als_compose_z", "als_predict_at", "als_predict_dense",
           "als_topk_similarity", "als_graph_classify", "als_normalize_features", "als_impute_col_median",
           "als_host_coo_to_sides", "als_host_row_tasks", "als_host_level_schedule")

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class RowSolveParams(C.Structure):
    """struct als_row_solve_params (include/als_hip.h)."""
    _fields_ = [
        ("k", _i32), ("ld", _i32), ("nrows", _i64), ("F_zero_row", _i32), ("reserved0", _i32),
        ("gram_mode", _i32), ("ndual_tail", _i32), ("ndual_mid", _i32), ("F_scale_ready", _i32),
        ("indptr", _vp), ("indices", _vp), ("vals", _vp), ("F", _vp),
        ("bias_self", _vp), ("bias_other", _vp), ("mu", _vp),
        ("lambda_scalar", _f32), ("lambda_row", _vp),
        ("lambda_bias_scalar", _f32), ("lambda_bias_row", _vp),
        ("rhs_extra", _vp), ("diag_extra", _vp),
        ("X_out", _vp), ("bias_out", _vp), ("gram_out", _vp), ("factor_out", _vp),
        ("rhs_out", _vp), ("colsum_out", _vp), ("sumr_out", _vp), ("sumr2_out", _vp), ("stat_out", _vp),
        ("status", _vp),
        ("tasks", _vp), ("ntasks", _i64), ("long_rows", _vp), ("nlong", _i64),
        ("workspace", _vp), ("cond_limit", _f32), ("byproducts_f64", _i32), ("redo_count", _vp), ("redo_rows", _vp),
        ("cond_out", _vp), ("F_scale", _vp), ("F_planes", _vp),
    ]


class GsSweepParams(C.Structure):
    """struct als_gs_sweep_params (include/als_hip.h)."""
    _fields_ = [
        ("k", _i32), ("ld", _i32), ("items", _vp), ("nitems", _i64),
        ("S_ptr", _vp), ("S_idx", _vp), ("S_val", _vp), ("alpha", _f32),
        ("factor", _vp), ("rhs", _vp), ("colsum", _vp), ("sumr", _vp),
        ("indptr", _vp), ("lambda_bias_scalar", _f32), ("lambda_bias_row", _vp),
        ("V", _vp), ("bias", _vp), ("sumr2", _vp), ("lambda_eff", _vp), ("stat_out", _vp),
        ("f64", _i32), ("reserved", _i32),
    ]


class WParams(C.Structure):
    """struct als_w_params (include/als_hip.h)."""
    _fields_ = [
        ("k", _i32), ("ld", _i32), ("phase", _i32), ("nfeat", _i32),
        ("item_begin", _i64), ("item_end", _i64),
        ("gram", _vp), ("rhs", _vp), ("colsum", _vp), ("V", _vp), ("b_new", _vp), ("b_old", _vp),
        ("D", _i32), ("f64", _i32), ("X", _vp), ("feat_off", _vp), ("W", _vp), ("H", _vp),
        ("nrows_h", _i64),
        ("feat_index", _i32), ("feat_col0", _i32), ("feat_d", _i32), ("nchunks", _i32),
        ("partA", _vp), ("partB", _vp), ("A_out", _vp), ("B_out", _vp),
    ]


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def load():
    """Load libals_hip.so and declare prototypes (idempotent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C collaborative-filtering_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipLibraryMissing(f"{LIB_PATH} does not export {name}")
    lib.als_version.restype = C.c_int
    lib.als_padded_k.argtypes = [C.c_int]
    lib.als_perm_index.argtypes = [C.c_int, C.c_int]
    lib.als_partial_slot_bytes.argtypes = [C.c_int]
    lib.als_partial_slot_bytes.restype = _i64
    lib.als_partial_slot_bytes_f64.argtypes = [C.c_int]
    lib.als_partial_slot_bytes_f64.restype = _i64
    lib.als_row_solve.argtypes = [C.POINTER(RowSolveParams), _vp]
    lib.als_factor_scale.argtypes = [_vp, _i64, _vp, _vp]
    lib.als_gs_sweep.argtypes = [C.POINTER(GsSweepParams), _vp]
    lib.als_gs_sweep_levels.argtypes = [C.POINTER(GsSweepParams), _vp, _i64, _vp]
    lib.als_gs_sweep_dataflow.argtypes = [C.POINTER(GsSweepParams), _vp, _vp, _i64, _vp, _vp, _vp]
    lib.als_residual_stats.argtypes = [C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _i64, _vp, _vp, _vp]
    lib.als_w_normal_equations.argtypes = [C.POINTER(WParams), _vp]
    lib.als_sumsq_partials.restype = C.c_int
    lib.als_item_stats.argtypes = [C.c_int, C.c_int, _i64, _i64] + [_vp] * 11
    lib.als_item_stats_f64.argtypes = [C.c_int, C.c_int, _i64, _i64] + [_vp] * 11
    lib.als_spd_solve_workspace_bytes.argtypes = [_i64]
    lib.als_spd_solve_f64.argtypes = [_i64, _vp, _i64, _vp, C.c_double, _vp, _vp, _vp, _vp]
    lib.als_sumsq.argtypes = [_vp, _i64, _vp, _vp, _vp]
    lib.als_sum_pairs.argtypes = [_vp, _i64, _vp, _vp, _vp]
    lib.als_history_row.argtypes = [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]
    lib.als_compose_z.argtypes = [_i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.als_predict_at.argtypes = [C.c_int, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.als_predict_dense.argtypes = [C.c_int, C.c_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.als_topk_similarity.argtypes = [_i64, _i64, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]
    lib.als_graph_classify.argtypes = [_i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.als_normalize_features.argtypes = [_i64, C.c_int, _vp, C.c_int, C.c_double, _vp, _vp, _vp, _vp]
    lib.als_impute_col_median.argtypes = [_i64, C.c_int, _vp, _vp, _vp]
    lib.als_host_coo_to_sides.argtypes = [_i64, _i64, _i64] + [_vp] * 9
    lib.als_host_row_tasks.argtypes = [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]
    lib.als_host_level_schedule.argtypes = [_i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]
    for name in EXPORTS:
        if name not in ("als_partial_slot_bytes", "als_partial_slot_bytes_f64", "als_spd_solve_workspace_bytes"):
            getattr(lib, name).restype = C.c_int
    lib.als_spd_solve_workspace_bytes.restype = C.c_size_t
    _lib = lib
    return lib
