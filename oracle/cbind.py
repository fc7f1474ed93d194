"""ctypes binding of the oracle's C restatement (``oracle/rk_oracle.c``) - TEST / CPU-BASELINE
INFRASTRUCTURE (see ``oracle/__init__.py``)."""

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "librk_oracle.so")
_KINDS = {"rbf": 0, "laplace": 1, "matern12": 2, "matern32": 3, "matern52": 4}
_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        for name in ("rk_oracle_kernel_matmat_f32", "rk_oracle_kernel_matmat_f64"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                           ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p,
                           ctypes.c_double, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        for name in ("rk_oracle_kernel_matrix_f32", "rk_oracle_kernel_matrix_f64"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                           ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p,
                           ctypes.c_double, ctypes.c_void_p]
        for name in ("rk_oracle_dense_matmat_f32", "rk_oracle_dense_matmat_f64"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                           ctypes.c_void_p]
        _lib.rk_oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib.rk_oracle_set_num_threads.restype = None
        _lib.rk_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(load().rk_oracle_num_threads())


def kernel_matmat(kind, X1, X2, lengthscale, V, const_scaling=1.0):
    """const_scaling * K(X1, X2) @ V on the host cores (OpenMP), float32 or float64."""
    lib = load()
    X1, X2 = X1.contiguous(), X2.contiguous()
    V2 = (V[:, None] if V.ndim == 1 else V).contiguous()
    ell_vec, ell = None, 1.0
    if isinstance(lengthscale, torch.Tensor):
        ell_vec = lengthscale.to(X1.dtype).contiguous()
    else:
        ell = float(lengthscale)
    Y = torch.empty((X1.shape[0], V2.shape[1]), dtype=X1.dtype)
    fn = lib.rk_oracle_kernel_matmat_f32 if X1.dtype == torch.float32 else lib.rk_oracle_kernel_matmat_f64
    rc = fn(_KINDS[kind], X1.data_ptr(), X1.shape[0], X2.data_ptr(), X2.shape[0], X1.shape[1], ell,
            None if ell_vec is None else ell_vec.data_ptr(), float(const_scaling), V2.data_ptr(),
            V2.shape[1], Y.data_ptr())
    if rc != 0:
        raise ValueError("rk_oracle_kernel_matmat: bad arguments")
    return Y[:, 0] if V.ndim == 1 else Y


def kernel_matrix(kind, X1, X2, lengthscale, const_scaling=1.0):
    """Dense const_scaling * K(X1, X2) on the host cores (OpenMP), float32 or float64."""
    lib = load()
    X1, X2 = X1.contiguous(), X2.contiguous()
    ell_vec, ell = None, 1.0
    if isinstance(lengthscale, torch.Tensor):
        ell_vec = lengthscale.to(X1.dtype).contiguous()
    else:
        ell = float(lengthscale)
    K = torch.empty((X1.shape[0], X2.shape[0]), dtype=X1.dtype)
    fn = lib.rk_oracle_kernel_matrix_f32 if X1.dtype == torch.float32 else lib.rk_oracle_kernel_matrix_f64
    rc = fn(_KINDS[kind], X1.data_ptr(), X1.shape[0], X2.data_ptr(), X2.shape[0], X1.shape[1], ell,
            None if ell_vec is None else ell_vec.data_ptr(), float(const_scaling), K.data_ptr())
    if rc != 0:
        raise ValueError("rk_oracle_kernel_matrix: bad arguments")
    return K


def dense_matmat(K, V):
    """K @ V for a dense row-major K and at most 16 columns of V, on the host cores (OpenMP)."""
    lib = load()
    V2 = (V[:, None] if V.ndim == 1 else V).contiguous()
    if not K.is_contiguous() or K.dtype != V2.dtype or K.shape[1] != V2.shape[0] or V2.shape[1] > 16:
        raise ValueError("dense_matmat: K must be contiguous, of V's dtype and shape, with at most 16 columns in V")
    Y = torch.empty((K.shape[0], V2.shape[1]), dtype=K.dtype)
    fn = lib.rk_oracle_dense_matmat_f32 if K.dtype == torch.float32 else lib.rk_oracle_dense_matmat_f64
    if fn(K.data_ptr(), K.shape[0], K.shape[1], V2.data_ptr(), V2.shape[1], Y.data_ptr()) != 0:
        raise ValueError("rk_oracle_dense_matmat: bad arguments")
    return Y[:, 0] if V.ndim == 1 else Y


def host_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box gives
    one GPU's share of a 256-thread host; threads beyond the quota only get throttled)."""
    import math
    import os

    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def set_num_threads(t: int) -> None:
    load().rk_oracle_set_num_threads(int(t))
