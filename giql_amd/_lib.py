"""ctypes binding of ``libgiql_hip.so`` (C ABI declared in ``include/giql_hip.h``).

The product path has NO CPU fallback: if the HIP library is missing or fails to
load, importing the engine raises :class:`GiqlHipUnavailable` loudly.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

GIQL_OK = 0
GIQL_ERR_INVALID = -1
GIQL_ERR_HIP = -2
GIQL_ERR_NOMEM = -3
GIQL_ERR_CHROM = -4
GIQL_ERR_SPAN = -5
GIQL_ERR_CAPACITY = -6
GIQL_ERR_STATE = -7

PHASES = [
    "span", "linearize", "sort_hist", "sort_scan", "sort_scatter", "count", "scan",
    "partition", "fill", "irregular", "aux", "sort_local",
]
N_PHASES = 16

#: every symbol include/giql_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "giql_hip_abi_version", "giql_hip_last_error", "giql_hip_device_count",
    "giql_hip_create", "giql_hip_destroy", "giql_hip_reserve", "giql_hip_set_profiling",
    "giql_hip_get_stats", "giql_hip_inner_plan_dev", "giql_hip_inner_fill_dev",
    "giql_hip_semi_anti_dev", "giql_hip_count_dev", "giql_hip_nearest_dev", "giql_hip_chrom_spans_dev",
    "giql_hip_inner", "giql_hip_semi_anti", "giql_hip_count", "giql_hip_nearest",
    "giql_hip_free_host", "giql_hip_pairs_checksum_dev",
    "giql_hip_take_dev", "giql_hip_take_utf8_plan_dev", "giql_hip_take_utf8_fill_dev",
    "giql_hip_select_dev", "giql_hip_select_expr_dev", "giql_hip_mark_dev", "giql_hip_cluster_dev", "giql_hip_cluster_pred_dev", "giql_hip_merge_dev", "giql_hip_merge_pred_dev",
    "giql_hip_group_rows_dev", "giql_hip_segment_sum_dev", "giql_hip_inner_join_dev",
    "giql_hip_inner_plan_export_dev", "giql_hip_fill_from_plan_dev", "giql_hip_copy_probe_dev",
    "giql_hip_nearest_k_dev", "giql_hip_stream_probe_dev", "giql_hip_host_pool_trim",
    "giql_hip_nearest32_dev", "giql_hip_index_create_dev", "giql_hip_index_destroy", "giql_hip_index_info",
    "giql_hip_inner_join_indexed_dev",
]


class GiqlHipUnavailable(RuntimeError):
    """libgiql_hip.so could not be loaded (not built, or no ROCm runtime)."""


class GiqlHipError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"giql_hip error {code}: {message}")
        self.code = code


class CSide(ctypes.Structure):
    """``giql_side`` (include/giql_hip.h)."""

    _fields_ = [
        ("chrom", ctypes.c_void_p),
        ("start", ctypes.c_void_p),
        ("end", ctypes.c_void_p),
        ("n", ctypes.c_int64),
        ("start_off", ctypes.c_int32),
        ("end_off", ctypes.c_int32),
    ]


class CStats(ctypes.Structure):
    """``giql_hip_stats`` (include/giql_hip.h)."""

    _fields_ = [
        ("n_a", ctypes.c_int64),
        ("n_b", ctypes.c_int64),
        ("n_out", ctypes.c_int64),
        ("n_irregular_a", ctypes.c_int64),
        ("n_irregular_b", ctypes.c_int64),
        ("workspace_bytes", ctypes.c_int64),
        ("span", ctypes.c_int64),
        ("phase_ms", ctypes.c_float * N_PHASES),
        ("phase_launches", ctypes.c_int32 * N_PHASES),
        ("phase_bytes", ctypes.c_int64 * N_PHASES),
        ("total_ms", ctypes.c_float),
        ("profiled", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


def lib_path() -> str:
    return os.environ.get("GIQL_HIP_LIB") or os.path.join(_HERE, "libgiql_hip.so")


class COperand(ctypes.Structure):
    """``giql_operand`` (include/giql_hip.h)."""

    _fields_ = [
        ("side", ctypes.c_int32),
        ("type", ctypes.c_int32),
        ("data", ctypes.c_void_p),
        ("valid", ctypes.c_void_p),
        ("lit_i", ctypes.c_int64),
        ("lit_f", ctypes.c_double),
        ("lit_is_float", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class CPred(ctypes.Structure):
    """``giql_pred`` (include/giql_hip.h)."""

    _fields_ = [("lhs", COperand), ("rhs", COperand), ("op", ctypes.c_int32), ("group", ctypes.c_int32)]


OPS = {"=": 0, "==": 0, "!=": 1, "<>": 1, "<": 2, "<=": 3, ">": 4, ">=": 5, "isnull": 6, "notnull": 7, "istrue": 8}
SIDE_A, SIDE_B, SIDE_LIT, SIDE_EXPR = 0, 1, 2, 3
T_I32, T_I64, T_F32, T_F64, T_U8 = range(5)

_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library once and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    # One process, one HIP runtime: libgiql_hip.so needs libamdhip64.so.7, and so
    # does torch (which bundles its own copy under the same soname).  Whichever
    # is loaded first serves both, so import torch first when it is installed:
    # device tensors handed to the C ABI then belong to the runtime that runs the
    # kernels.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch-less deployment
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise GiqlHipUnavailable(
            f"{path} not found: build it with giql_amd/csrc/build.sh "
            "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        L = ctypes.CDLL(path)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise GiqlHipUnavailable(f"cannot load {path}: {exc}") from exc
    P = ctypes.POINTER
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.giql_hip_abi_version.restype = ctypes.c_int
    L.giql_hip_last_error.restype = ctypes.c_char_p
    L.giql_hip_device_count.argtypes = [P(ctypes.c_int)]
    L.giql_hip_create.argtypes = [ctypes.c_int, P(vp)]
    L.giql_hip_destroy.argtypes = [vp]
    L.giql_hip_reserve.argtypes = [vp, i64]
    L.giql_hip_set_profiling.argtypes = [vp, ctypes.c_int]
    L.giql_hip_get_stats.argtypes = [vp, P(CStats)]
    L.giql_hip_inner_plan_dev.argtypes = [vp, P(CSide), P(CSide), i32, vp, P(i64)]
    L.giql_hip_inner_fill_dev.argtypes = [vp, vp, vp, i64, vp]
    L.giql_hip_inner_join_dev.argtypes = [vp, P(CSide), P(CSide), i32, vp, vp, i64, vp, P(i64)]
    L.giql_hip_semi_anti_dev.argtypes = [vp, P(CSide), P(CSide), i32, ctypes.c_int, vp, P(i64), vp]
    L.giql_hip_count_dev.argtypes = [vp, P(CSide), P(CSide), i32, vp, vp]
    L.giql_hip_nearest_dev.argtypes = [vp, P(CSide), P(CSide), i32, ctypes.c_int, i64, vp, vp, vp]
    L.giql_hip_index_create_dev.argtypes = [vp, P(CSide), i32, vp, P(vp)]
    L.giql_hip_index_destroy.argtypes = [vp]
    L.giql_hip_index_info.argtypes = [vp, P(i64), P(i64), P(i32), P(i64)]
    L.giql_hip_inner_join_indexed_dev.argtypes = [vp, vp, P(CSide), vp, vp, i64, vp, P(i64)]
    L.giql_hip_nearest32_dev.argtypes = [vp, P(CSide), P(CSide), i32, ctypes.c_int, i64, vp, vp]
    L.giql_hip_nearest_k_dev.argtypes = [vp, P(CSide), P(CSide), i32, i32, ctypes.c_int, i64, vp, vp, vp]
    L.giql_hip_chrom_spans_dev.argtypes = [vp, P(CSide), P(CSide), i32, vp, vp]
    L.giql_hip_inner.argtypes = [vp, P(CSide), P(CSide), i32, P(i64), P(vp), P(vp)]
    L.giql_hip_semi_anti.argtypes = [vp, P(CSide), P(CSide), i32, ctypes.c_int, P(i64), P(vp)]
    L.giql_hip_count.argtypes = [vp, P(CSide), P(CSide), i32, vp]
    L.giql_hip_nearest.argtypes = [vp, P(CSide), P(CSide), i32, ctypes.c_int, i64, vp, vp]
    L.giql_hip_free_host.argtypes = [vp]
    L.giql_hip_free_host.restype = None
    L.giql_hip_pairs_checksum_dev.argtypes = [vp, vp, vp, i64, vp, P(ctypes.c_uint64)]
    L.giql_hip_take_dev.argtypes = [vp, P(vp), P(i32), i32, i64, vp, i64, P(vp), vp]
    L.giql_hip_take_utf8_plan_dev.argtypes = [vp, vp, i64, vp, i64, vp, P(i64), vp]
    L.giql_hip_take_utf8_fill_dev.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, vp]
    L.giql_hip_select_dev.argtypes = [vp, P(CPred), i32, vp, i64, vp, i64, i64, vp, vp, P(i64), vp]
    L.giql_hip_select_expr_dev.argtypes = [vp, P(CPred), i32, P(COperand), i32, vp, i64, vp, i64, i64, vp, vp, P(i64), vp]
    L.giql_hip_mark_dev.argtypes = [vp, vp, i64, vp, i64, vp]
    L.giql_hip_group_rows_dev.argtypes = [vp, P(CSide), i32, vp, vp, P(i64), vp]
    L.giql_hip_segment_sum_dev.argtypes = [vp, vp, vp, i64, vp, i64, vp]
    L.giql_hip_cluster_dev.argtypes = [vp, P(CSide), i32, i64, vp, vp]
    L.giql_hip_cluster_pred_dev.argtypes = [vp, P(CSide), i32, i64, P(CPred), i32, vp, vp]
    L.giql_hip_inner_plan_export_dev.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, P(i32), P(i64), P(i64), vp]
    L.giql_hip_fill_from_plan_dev.argtypes = [vp, vp, vp, vp, i64, vp, i64, vp, vp, i64, i64, vp, P(i64)]
    L.giql_hip_copy_probe_dev.argtypes = [vp, vp, vp, i64, i32, vp, P(ctypes.c_double)]
    L.giql_hip_stream_probe_dev.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, P(ctypes.c_double)]
    L.giql_hip_host_pool_trim.argtypes = [i64, P(i64)]
    L.giql_hip_merge_dev.argtypes = [vp, P(CSide), i32, i64, vp, vp, vp, vp, i64, P(i64), vp]
    L.giql_hip_merge_pred_dev.argtypes = [vp, P(CSide), i32, i64, P(CPred), i32, vp, vp, vp, vp, i64, P(i64), vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != GIQL_OK:
        msg = load().giql_hip_last_error()
        raise GiqlHipError(rc, msg.decode("utf-8", "replace") if msg else "")
