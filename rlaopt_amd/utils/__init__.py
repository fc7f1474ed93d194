"""Validators and the iteration logger (mirrors ``rlaopt.utils``)."""
from . import input_checkers, logger
from .input_checkers import *  # noqa: F401,F403
from .logger import *  # noqa: F401,F403

__all__ = list(input_checkers.__all__) + list(logger.__all__)


import contextlib as _contextlib

import torch as _torch


@_contextlib.contextmanager
def small_host_lapack(order: int):
    """Run the float64 r x r host factorizations (Cholesky, eigh, triangular solves) with a
    thread count that suits their size: LAPACK on a 100 x 100 matrix with the 128 OpenMP
    threads of a GPU host is ~100x slower than with one (measured: 156 ms vs ~1 ms per eigh)."""
    want = 1 if order <= 256 else (4 if order <= 1024 else 16)
    prev = _torch.get_num_threads()
    if prev > want:
        _torch.set_num_threads(want)
    try:
        yield
    finally:
        if _torch.get_num_threads() != prev:
            _torch.set_num_threads(prev)
