"""Preconditioner configuration dataclasses (rlaopt/preconditioners/configs.py:25-132)."""

from abc import ABC
from dataclasses import asdict, dataclass
from typing import Any

from ..utils import _is_nonneg_float, _is_pos_int
from .enums import _DampingMode

__all__ = [
    "PreconditionerConfig", "IdentityConfig", "NewtonConfig", "NystromConfig", "SkPreConfig",
    "_is_precond_config",
]


@dataclass(kw_only=True, frozen=False)
class PreconditionerConfig(ABC):
    def to_dict(self) -> dict:
        return asdict(self)


@dataclass(kw_only=True, frozen=False)
class IdentityConfig(PreconditionerConfig):
    """No parameters."""


@dataclass(kw_only=True, frozen=False)
class NewtonConfig(PreconditionerConfig):
    rho: float

    def __post_init__(self):
        _is_nonneg_float(self.rho, "rho")


@dataclass(kw_only=True, frozen=False)
class NystromConfig(PreconditionerConfig):
    """rank, damping rho, sketch ('ortho' default) and damping mode ('adaptive' default)."""

    rank: int
    rho: float
    sketch: str = "ortho"
    damping_mode: str = "adaptive"

    def __post_init__(self):
        _is_pos_int(self.rank, "rank")
        _is_nonneg_float(self.rho, "rho")
        self.damping_mode = _DampingMode._from_str(self.damping_mode, "damping_mode")


@dataclass(kw_only=True, frozen=False)
class SkPreConfig(PreconditionerConfig):
    sketch_size: int
    rho: float
    sketch: str = "sparse"

    def __post_init__(self):
        _is_pos_int(self.sketch_size, "sketch_size")
        _is_nonneg_float(self.rho, "rho")


def _is_precond_config(param: Any, param_name: str):
    if not isinstance(param, PreconditionerConfig):
        raise TypeError(
            f"{param_name} is of type {type(param).__name__}, "
            "but expected type PreconditionerConfig"
        )
