"""Validators and the iteration logger (mirrors ``rlaopt.utils``)."""
from . import input_checkers, logger
from .input_checkers import *  # noqa: F401,F403
from .logger import *  # noqa: F401,F403

__all__ = list(input_checkers.__all__) + list(logger.__all__)


import contextlib as _contextlib
import gc as _gc

import torch as _torch


_quiet_depth = 0
_quiet_saved = None


@_contextlib.contextmanager
def quiet_host(max_threads: int = 8):
    """Keep torch's CPU thread pool small while a solver loop runs.  The host side of this
    engine only does r x r algebra and bookkeeping; with the 128 OpenMP threads PyTorch starts on
    a GPU host, every excursion into a parallel region leaves ~128 spinning workers behind, and
    a loop that synchronises with the device (SAP: eigh + power iteration per step) was measured
    to lose 45 ms out of every 100 ms to them (33 ms -> 18 ms per ASkotch step once limited).
    Re-entrant; the previous thread count is restored when the outermost context exits.

    The same context parks the objects that already exist in the garbage collector's permanent
    generation (``gc.freeze``): a full collection over the ~10^6 objects a process holds after
    importing torch takes ~40 ms - 25 LSQR iterations - and otherwise strikes every few dozen
    steps.  Collection stays enabled; it just no longer re-scans the frozen objects."""
    global _quiet_depth, _quiet_saved
    if _quiet_depth == 0:
        prev = _torch.get_num_threads()
        _quiet_saved = prev if prev > max_threads else None
        if _quiet_saved is not None:
            _torch.set_num_threads(max_threads)
        _gc.freeze()
    _quiet_depth += 1
    try:
        yield
    finally:
        _quiet_depth -= 1
        if _quiet_depth == 0:
            _gc.unfreeze()
            if _quiet_saved is not None:
                _torch.set_num_threads(_quiet_saved)
                _quiet_saved = None


@_contextlib.contextmanager
def small_host_lapack(order: int):
    """Run the float64 r x r host factorizations (Cholesky, eigh, triangular solves) with a
    thread count that suits their size: LAPACK on a 100 x 100 matrix with the 128 OpenMP
    threads of a GPU host is ~100x slower than with one (measured: 156 ms vs ~1 ms per eigh)."""
    want = 1 if order <= 256 else (4 if order <= 1024 else 16)
    prev = _torch.get_num_threads()
    # (only when the pool is far too wide: every change of the thread count makes the next parallel region rebuild its
    # team - a 512 x 512 Cholesky was measured at 66 ms instead of 2 ms right after a switch; inside quiet_host's
    # 8-thread pool an order-512 factorisation simply keeps the 8 threads)
    if prev > 2 * want:
        _torch.set_num_threads(want)
    try:
        yield
    finally:
        if _torch.get_num_threads() != prev:
            _torch.set_num_threads(prev)
