"""Callable-backed operators (rlaopt/linops/simple.py:15-104): ``LinOp`` (A @ x),
``TwoSidedLinOp`` (adds x @ A and .T) and ``SymmetricLinOp`` (.T is the operator itself).

When no mat-mat callable is supplied the columns are pushed through the mat-vec one at a
time (the reference uses ``torch.vmap``, simple.py:32,62, which cannot trace calls into a
native library)."""

from __future__ import annotations

from typing import Callable

import torch

from ..utils import _is_callable
from .base import _BaseLinOp

__all__ = ["LinOp", "TwoSidedLinOp", "SymmetricLinOp"]

_DEFAULT_DTYPE = torch.get_default_dtype()


def _columnwise(matvec: Callable) -> Callable:
    def matmat(X: torch.Tensor) -> torch.Tensor:
        return torch.stack([matvec(X[:, j]) for j in range(X.shape[1])], dim=1)

    return matmat


class LinOp(_BaseLinOp):
    def __init__(self, device, shape, matvec, matmat=None, dtype=_DEFAULT_DTYPE):
        super().__init__(device=device, shape=shape, dtype=dtype)
        _is_callable(matvec, "matvec")
        if matmat is not None:
            _is_callable(matmat, "matmat")
        self._matvec_fn = matvec
        self._matmat_fn = matmat if matmat is not None else _columnwise(matvec)

    def _matvec(self, x):
        return self._matvec_fn(x)

    def _matmat(self, x):
        return self._matmat_fn(x)


class TwoSidedLinOp(LinOp):
    def __init__(self, device, shape, matvec, rmatvec, matmat=None, rmatmat=None, dtype=_DEFAULT_DTYPE):
        super().__init__(device, shape, matvec, matmat, dtype)
        _is_callable(rmatvec, "rmatvec")
        if rmatmat is not None:
            _is_callable(rmatmat, "rmatmat")
        self._rmatvec_fn = rmatvec
        self._rmatmat_fn = rmatmat if rmatmat is not None else _columnwise(rmatvec)

    def _rmatvec(self, x):
        return self._rmatvec_fn(x)

    def _rmatmat(self, x):
        return self._rmatmat_fn(x)

    @property
    def T(self):
        rows, cols = self.shape
        return TwoSidedLinOp(
            device=self.device,
            shape=torch.Size((cols, rows)),
            matvec=self._rmatvec,
            rmatvec=self._matvec,
            matmat=self._rmatmat,
            rmatmat=self._matmat,
            dtype=self.dtype,
        )


class SymmetricLinOp(TwoSidedLinOp):
    def __init__(self, device, shape, matvec, matmat=None, dtype=_DEFAULT_DTYPE):
        super().__init__(device, shape, matvec, matvec, matmat, matmat, dtype)
        if shape[0] != shape[1]:
            raise ValueError(
                f"SymmetricLinOp requires the shape to be square. The received shape is {shape}."
            )

    @property
    def T(self):
        return self
