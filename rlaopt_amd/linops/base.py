"""Operator protocol (rlaopt/linops/base.py:11-111) and the sharded-operator base class.

The protocol is the reference's: ``shape``, ``dtype``, ``device`` / ``devices``, ``A @ x``
for 1-D / 2-D ``x``, ``x @ A``, ``A.T``.  What differs is the transport of the distributed
operator: the reference spawns one worker process per device and ships pickled operators
and host-staged tensors through queues (base.py:157-267); here every shard is a
device-resident object driven from its own rank (``torch.distributed``, see
``rlaopt_amd/dist.py``) - or, inside one process, simply visited in turn on its device's
stream - and partial results are combined with collectives / device copies.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import torch

from .. import dist as _dist
from ..utils import _is_list, _is_torch_device, _is_torch_f32_f64, _is_torch_size
from .enums import _DistributionMode, _Operation

__all__: list[str] = []


class _BaseLinOp(ABC):
    """Base class of all linear operators."""

    def __init__(self, device: torch.device, shape: torch.Size, dtype: torch.dtype):
        self._check_inputs_base(device, shape, dtype)
        self._device, self._shape, self._dtype = device, shape, dtype

    @staticmethod
    def _check_inputs_base(device: Any, shape: Any, dtype: Any):
        _is_torch_device(device, "device")
        _is_torch_size(shape, "shape")
        if len(shape) != 2:
            raise ValueError(f"shape must have two elements. Received {len(shape)}")
        if any(not isinstance(s, int) or s <= 0 for s in shape):
            raise ValueError(f"shape must contain positive integers. Received {shape}")
        _is_torch_f32_f64(dtype, "dtype")

    device = property(lambda self: self._device, doc="Device of the operator.")
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)

    @property
    def devices(self) -> list[torch.device]:
        """All devices used (a one-element list for a single-device operator)."""
        return [self._device]

    @property
    def T(self) -> "_BaseLinOp":
        raise NotImplementedError("This linear operator doesn't support transposition")

    @abstractmethod
    def _matvec(self, x: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def _matmat(self, x: torch.Tensor) -> torch.Tensor: ...

    def _rmatvec(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This linear operator doesn't support right matvec")

    def _rmatmat(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This linear operator doesn't support right matmat")

    def __matmul__(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            return self._matvec(x)
        if x.ndim == 2:
            return self._matmat(x)
        raise ValueError(f"x must be a 1D or 2D tensor. Received {x.ndim}D tensor.")

    def __rmatmul__(self, x: torch.Tensor) -> torch.Tensor:
        # x @ A == (A^T x^T)^T   (linops/base.py:104-111)
        if x.ndim == 1:
            return self._rmatvec(x)
        if x.ndim == 2:
            return self._rmatmat(x.T).T
        raise ValueError(f"x must be a 1D or 2D tensor. Received {x.ndim}D tensor.")


class _BaseDistributedLinOp(_BaseLinOp):
    """An operator assembled from per-shard operators ``A[i]`` (one per device).

    Two execution layouts, chosen at construction:

    * one process (no ``torch.distributed`` group, or a group of size 1): the process owns
      every shard and visits them in turn; each shard's kernels are queued on its own
      device, inputs/outputs move with device-to-device copies;
    * SPMD (group of size p >= number of shards): rank i owns shard i only; ``A[j]`` for
      j != i are ``_RemoteShard`` placeholders that carry the shape only.  Combination is an
      all-gather (concat) or an all-reduce (sum) - see ``rlaopt_amd/dist.py``.

    ``shutdown()`` is kept for API compatibility (linops/base.py:278-288); there are no
    worker processes to stop.
    """

    def __init__(self, shape, A, distribution_mode, is_new=True, group=None, spmd=None, **_ignored):
        self._is_new = is_new
        _is_list(A, "A")
        live = list(A)
        if not live or not all(isinstance(a, _BaseLinOp) for a in live):
            raise ValueError("All elements in A must be instances of _BaseLinOp")
        if any(a.dtype != live[0].dtype for a in live):
            raise ValueError(
                "All linear operators must have the same dtype. "
                f"Received {', '.join(str(a.dtype) for a in live)}."
            )
        super().__init__(device=torch.device("cpu"), shape=shape, dtype=live[0].dtype)
        self._A = A
        self._devices = [a.device for a in live]
        self._distribution_mode = _DistributionMode._from_str(distribution_mode, "distribution_mode")
        self._group = group
        rank, size = _dist.world(group)
        self._spmd = _dist.spmd_active(group) if spmd is None else bool(spmd)
        if self._spmd and len(A) > size:
            raise ValueError(f"{len(A)} shards but only {size} ranks in the process group")
        self._rank = rank

    @property
    def device(self):
        raise AttributeError(
            "Distributed linear operators operate across multiple devices and don't have a "
            "single 'device'. Use the 'devices' property instead to get the list of all devices."
        )

    @property
    def devices(self):
        return self._devices

    # -- execution ---------------------------------------------------------------------------
    def _local_ids(self):
        if self._spmd:
            return [self._rank] if self._rank < len(self._A) else []
        return list(range(len(self._A)))

    def _bounds(self, by_dimension: int):
        """Cumulative [lo, hi) of each shard along ``by_dimension`` of the shard shapes."""
        sizes = self._shard_sizes(by_dimension)
        out, lo = [], 0
        for s in sizes:
            out.append((lo, lo + s))
            lo += s
        return out

    def _shard_sizes(self, by_dimension: int):
        return [a.shape[by_dimension] for a in self._A]

    def _run(self, x: torch.Tensor, operation: _Operation, chunk: bool, by_dimension: int):
        """Apply every locally owned shard; returns {shard id: result on x's device}."""
        bounds = self._bounds(by_dimension) if chunk else None
        out = {}
        for i in self._local_ids():
            op = self._A[i]
            xi = x[bounds[i][0] : bounds[i][1]] if chunk else x
            xi = xi.to(op.device, non_blocking=True)
            yi = (op @ xi) if operation is _Operation.MATVEC else (op.T @ xi)
            out[i] = yi.to(x.device, non_blocking=True)
        return out

    def _concat(self, parts: dict, x: torch.Tensor, by_dimension: int):
        """Stack shard results; ``by_dimension`` is the shard-shape axis the outputs follow."""
        if self._spmd:
            mine = parts.get(self._rank)
            if mine is None:
                k = () if x.ndim == 1 else (x.shape[1],)
                mine = torch.zeros((0, *k), dtype=x.dtype, device=x.device)
            return _dist.all_gather_rows(mine, self._shard_sizes(by_dimension), self._group)
        return torch.cat([parts[i] for i in sorted(parts)], dim=0)

    def _sum(self, parts: dict, x: torch.Tensor, out_rows: int):
        total = None
        for i in sorted(parts):
            total = parts[i] if total is None else total + parts[i]
        if total is None:
            k = () if x.ndim == 1 else (x.shape[1],)
            total = torch.zeros((out_rows, *k), dtype=x.dtype, device=x.device)
        if self._spmd:
            _dist.all_reduce_sum(total, self._group)
        return total

    def shutdown(self):
        """No worker processes exist; kept so reference scripts run unchanged."""
        return None


class _RemoteShard(_BaseLinOp):
    """Shape-only stand-in for a shard owned by another rank (SPMD layout)."""

    def _matvec(self, x):
        raise RuntimeError("this shard lives on another rank")

    _matmat = _matvec
