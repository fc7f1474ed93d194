"""Row / column distributed operators (rlaopt/linops/distributed.py:15-208).

Combination rules of the reference, kept exactly:

=========  ====================  =========================
mode       A @ x                 A.T @ x  (two-sided only)
=========  ====================  =========================
ROW        broadcast x, CONCAT   chunk x by rows, SUM
COLUMN     chunk x, SUM          broadcast x, CONCAT
=========  ====================  =========================

CONCAT is an all-gather and SUM an all-reduce when the shards live in different ranks
(see ``_BaseDistributedLinOp``).
"""

from __future__ import annotations

import torch

from .base import _BaseDistributedLinOp, _BaseLinOp
from .enums import _DistributionMode, _Operation

__all__ = ["DistributedLinOp", "DistributedTwoSidedLinOp", "DistributedSymmetricLinOp"]


class _DistributedLinOp(_BaseDistributedLinOp):
    def _matvec(self, w: torch.Tensor) -> torch.Tensor:
        if self._distribution_mode is _DistributionMode.ROW:
            parts = self._run(w, _Operation.MATVEC, chunk=False, by_dimension=0)
            return self._concat(parts, w, by_dimension=0)
        parts = self._run(w, _Operation.MATVEC, chunk=True, by_dimension=1)
        return self._sum(parts, w, self.shape[0])

    def _matmat(self, w: torch.Tensor) -> torch.Tensor:
        return self._matvec(w)


class _DistributedTwoSidedLinOp(_DistributedLinOp):
    def _rmatvec(self, w: torch.Tensor) -> torch.Tensor:
        if self._distribution_mode is _DistributionMode.ROW:
            parts = self._run(w, _Operation.RMATVEC, chunk=True, by_dimension=0)
            return self._sum(parts, w, self.shape[1])
        parts = self._run(w, _Operation.RMATVEC, chunk=False, by_dimension=1)
        return self._concat(parts, w, by_dimension=1)

    def _rmatmat(self, w: torch.Tensor) -> torch.Tensor:
        return self._rmatvec(w)

    @property
    def T(self) -> "_DistributedTwoSidedLinOp":
        flipped = (
            _DistributionMode.COLUMN
            if self._distribution_mode is _DistributionMode.ROW
            else _DistributionMode.ROW
        )
        local = set(self._local_ids())
        shards = [a.T if i in local else _transposed_placeholder(a) for i, a in enumerate(self._A)]
        return _DistributedTwoSidedLinOp(
            shape=torch.Size((self.shape[1], self.shape[0])),
            A=shards,
            distribution_mode=flipped,
            is_new=False,
            group=self._group,
            spmd=self._spmd,
        )


def _transposed_placeholder(a: _BaseLinOp):
    from .base import _RemoteShard

    return _RemoteShard(a._device, torch.Size((a.shape[1], a.shape[0])), a.dtype)


class _DistributedSymmetricLinOp(_DistributedTwoSidedLinOp):
    def __init__(self, shape, A, distribution_mode, is_new=True, **kw):
        super().__init__(shape=shape, A=A, distribution_mode=distribution_mode, is_new=is_new, **kw)
        if is_new and shape[0] != shape[1]:
            raise ValueError(
                "DistributedSymmetricLinOp requires the shape to be square. "
                f"The received shape is {shape}."
            )

    def _rmatvec(self, w):
        return self._matvec(w)

    def _rmatmat(self, w):
        return self._matmat(w)

    @property
    def T(self):
        return self


class DistributedLinOp(_DistributedLinOp):
    """Distributed operator over a list of per-device operators."""

    def __init__(self, shape: torch.Size, A: list[_BaseLinOp], distribution_mode: str, **kw):
        super().__init__(shape=shape, A=A, distribution_mode=distribution_mode, is_new=True, **kw)


class DistributedTwoSidedLinOp(_DistributedTwoSidedLinOp):
    """Distributed operator whose shards also support right multiplication."""

    def __init__(self, shape: torch.Size, A: list[_BaseLinOp], distribution_mode: str, **kw):
        super().__init__(shape=shape, A=A, distribution_mode=distribution_mode, is_new=True, **kw)


class DistributedSymmetricLinOp(_DistributedSymmetricLinOp):
    """Distributed symmetric operator (square; transpose is itself)."""

    def __init__(self, shape: torch.Size, A: list[_BaseLinOp], distribution_mode: str, **kw):
        super().__init__(shape=shape, A=A, distribution_mode=distribution_mode, is_new=True, **kw)
