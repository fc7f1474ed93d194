"""Preconditioner protocol (rlaopt/preconditioners/preconditioner.py:19-180):
``_update(A, device)``, ``P @ x``, ``P._inv @ x``, ``_inverse_matmul_compose(fn)``,
``_update_damping(rho0)``."""

from abc import ABC, abstractmethod
from typing import Callable

import torch

from ..utils import _is_torch_tensor_1d_2d
from .configs import PreconditionerConfig

__all__ = ["Preconditioner"]


class Preconditioner(ABC):
    def __init__(self, config: PreconditionerConfig):
        self.config = config

    @abstractmethod
    def _update(self, A, device: torch.device, *args, **kwargs): ...

    @abstractmethod
    def _matmul(self, x: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def _inverse_matmul_1d(self, x: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def _inverse_matmul_2d(self, x: torch.Tensor) -> torch.Tensor: ...

    def __matmul__(self, x: torch.Tensor) -> torch.Tensor:
        _is_torch_tensor_1d_2d(x, "x")
        return self._matmul(x)

    def _inverse_matmul(self, x: torch.Tensor) -> torch.Tensor:
        _is_torch_tensor_1d_2d(x, "x")
        return self._inverse_matmul_1d(x) if x.ndim == 1 else self._inverse_matmul_2d(x)

    def _inverse_matmul_compose(self, fn: Callable) -> Callable:
        return lambda *args, **kwargs: self._inverse_matmul(fn(*args, **kwargs))

    def _update_damping(self, baseline_rho: float):
        """No-op except for the Nystrom preconditioner."""

    @property
    def _inv(self) -> "_InvPreconditioner":
        return _InvPreconditioner(self)


class _InvPreconditioner:
    """``P._inv @ x`` == ``P._inverse_matmul(x)``."""

    def __init__(self, preconditioner: Preconditioner):
        self.preconditioner = preconditioner

    def __matmul__(self, x: torch.Tensor) -> torch.Tensor:
        return self.preconditioner._inverse_matmul(x)
