"""ctypes binding of ``librk_hip.so`` (the C ABI declared in ``include/rk.h``).

The library is built in-tree by ``make -C rlaopt_amd/csrc`` (``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or fails to load, every product
entry point raises ``RuntimeError`` at first use.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# RK_HIP_LIBRARY points at another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RK_HIP_LIBRARY") or os.path.join(_HERE, "librk_hip.so")

RK_F32, RK_F64 = 0, 1
GRAM_MAX_NORM2 = 32.0  # RK_GRAM_MAX_NORM2 of include/rk.h
KIND_IDS = {"rbf": 0, "laplace": 1, "matern12": 2, "matern32": 3, "matern52": 4}
ABI_VERSION = 23

# name -> (restype, argtypes); must list every symbol declared in include/rk.h
SIGNATURES = {
    "rk_last_error": (c_char_p, []),
    "rk_abi_version": (c_int, []),
    "rk_kernel_materialize": (
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
         c_int64, c_double, c_void_p, c_double, c_void_p, c_int64, c_void_p],
    ),
    "rk_kernel_matmat_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_kernel_matmat_sym_ws_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "rk_kernel_matmat": (
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
         c_int64, c_double, c_void_p, c_double, c_void_p, c_int64, c_int64, c_void_p, c_int64,
         c_int, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "rk_kernel_center": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "rk_kernel_gram_bound": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "rk_sap_update": (
        c_int,
        [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
         c_int64, c_void_p, c_double, c_int, c_double, c_double, c_double, c_void_p],
    ),
    "rk_power_iteration_ws_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "rk_power_iteration": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_void_p, c_int,
         c_double, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "rk_sympack_elems": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_kernel_materialize_sympack": (
        c_int,
        [c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_double,
         c_int64, c_int64, c_void_p, c_void_p],
    ),
    "rk_symm_matmat_ws_bytes": (c_size_t, [c_int, c_int, c_int64, c_int64, c_int64, c_int64]),
    "rk_symm_matmat": (
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_double,
         c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_symm_matmat_f16x2_ws_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "rk_symm_matmat_f16x2": (
        c_int,
        [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_double, c_double, c_void_p,
         c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_kernel_materialize_mc_ws_bytes": (c_size_t, [c_int64, c_int64]),
    "rk_kernel_materialize_mc": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_double, c_void_p,
         c_void_p, c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_kernel_materialize_sympack_mc_ws_bytes": (c_size_t, [c_int64]),
    "rk_kernel_materialize_sympack_mc": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_void_p, c_double, c_int64, c_int64, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    "rk_symm_gemm_ws_bytes": (c_size_t, [c_int64, c_int64]),
    "rk_symm_gemm": (
        c_int,
        [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64,
         c_void_p, c_size_t, c_void_p],
    ),
    "rk_symm_gemm_rhs16": (
        c_int,
        [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64,
         c_void_p, c_size_t, c_void_p],
    ),
    "rk_kernel_sketch_ws_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "rk_kernel_sketch_matmat": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_void_p,
         c_double, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_kernel_sketch_matmat_rhs16": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_void_p,
         c_double, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_nystrom_inv_apply_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_nystrom_inv_apply": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_double, c_void_p, c_int64, c_int64, c_void_p,
         c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_comm_unique_id": (c_int, [c_void_p]),
    "rk_comm_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "rk_comm_destroy": (c_int, [c_void_p]),
    "rk_comm_allreduce_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rk_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rk_comm_reduce_scatter_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rk_spmd_pack": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p],
    ),
    "rk_matmat_skinny_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_matmat_skinny": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_double,
         c_double, c_void_p, c_int64, c_double, c_int64, c_void_p, c_int64, c_void_p, c_size_t,
         c_void_p],
    ),
    "rk_matmat_skinny_t_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_matmat_skinny_t": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_double,
         c_double, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p],
    ),
    "rk_gemm_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_gemm": (
        c_int,
        [c_int, c_int, c_int, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_void_p,
         c_int64, c_double, c_void_p, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_gram_tn_ws_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "rk_gram_tn": (
        c_int,
        [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
         c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "rk_pcg_update_wr": (
        c_int,
        [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
         c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "rk_pcg_update_p": (
        c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]
    ),
    "rk_gram_small_ws_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "rk_gram_small": (
        c_int,
        [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
         c_size_t, c_void_p],
    ),
    "rk_axpby_cols": (
        c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    ),
    "rk_normal_pass_supported": (c_int, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "rk_normal_pass_ws_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "rk_normal_pass": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "rk_small_solve": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rk_sparse_sign_apply": (
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_double,
         c_void_p, c_int64, c_void_p],
    ),
    "rk_randn": (
        c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_uint64, c_uint64, c_double, c_void_p]
    ),
    "rk_sample_distinct": (c_int, [c_int64, c_int64, c_uint64, c_void_p, c_void_p]),
    "rk_kernel_x2_image_bytes": (c_size_t, [c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int]),
    "rk_kernel_matmat_cached": (
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
         c_double, c_void_p, c_double, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
         c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_void_p],
    ),
    "rk_lsqr_tail": (
        c_int,
        [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
}

_lib = None
_load_error: Exception | None = None


def load() -> ctypes.CDLL:
    """Load (once) and return the library, with argtypes/restypes bound."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise RuntimeError(str(_load_error)) from _load_error
    try:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not found - build it with `make -C rlaopt_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`)"
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = restype, argtypes
        if lib.rk_abi_version() != ABI_VERSION:
            raise RuntimeError(
                f"librk_hip.so ABI {lib.rk_abi_version()} != binding ABI {ABI_VERSION}: rebuild"
            )
    except Exception as e:  # remember and re-raise on every later use
        _load_error = RuntimeError(
            f"rlaopt_amd: the HIP extension is unavailable ({e}). There is no CPU fallback."
        )
        raise _load_error from e
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    """Map a non-zero rk_status to a Python exception (ValueError for bad arguments)."""
    if rc == 0:
        return
    msg = load().rk_last_error().decode() or f"{what}: rk_status {rc}"
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError(msg)
