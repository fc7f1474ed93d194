"""Solver configuration dataclasses (rlaopt/solvers/configs.py:31-126)."""

from abc import ABC
from dataclasses import asdict, dataclass, field
from typing import Any

import torch

from ..preconditioners import IdentityConfig, PreconditionerConfig, _is_precond_config
from ..utils import _is_bool, _is_nonneg_float, _is_pos_float, _is_pos_int, _is_torch_device

__all__ = [
    "SAPAccelConfig", "SolverConfig", "PCGConfig", "SAPConfig", "LSQRConfig", "_is_solver_config",
    "_get_solver_name",
]


@dataclass(kw_only=True, frozen=False)
class SAPAccelConfig:
    """Nesterov parameters: 0 < mu <= nu and mu * nu <= 1."""

    mu: float
    nu: float

    def __post_init__(self):
        _is_pos_float(self.mu, "mu")
        _is_pos_float(self.nu, "nu")
        if self.mu > self.nu:
            raise ValueError("mu must be less than or equal to nu")
        if self.mu * self.nu > 1:
            raise ValueError("mu * nu must be less than or equal to 1")


@dataclass(kw_only=True, frozen=False)
class SolverConfig(ABC):
    def to_dict(self) -> dict:
        out = asdict(self)
        for key, value in list(out.items()):
            if isinstance(value, torch.device):
                out[key] = str(value)
        if isinstance(getattr(self, "precond_config", None), PreconditionerConfig):
            out["precond_config"] = self.precond_config.to_dict()
        return out


def _check_common(cfg):
    _is_torch_device(cfg.device, "device")
    _is_pos_int(cfg.max_iters, "max_iters")
    _is_nonneg_float(cfg.atol, "atol")
    _is_nonneg_float(cfg.rtol, "rtol")
    _is_precond_config(cfg.precond_config, "precond_config")


@dataclass(kw_only=True, frozen=False)
class PCGConfig(SolverConfig):
    device: torch.device
    max_iters: int = 1000
    atol: float = 0.0
    rtol: float = 1e-5
    precond_config: PreconditionerConfig = field(default_factory=IdentityConfig)

    def __post_init__(self):
        _check_common(self)


@dataclass(kw_only=True, frozen=False)
class SAPConfig(SolverConfig):
    device: torch.device
    max_iters: int = 1000
    atol: float = 0.0
    rtol: float = 1e-5
    precond_config: PreconditionerConfig = field(default_factory=IdentityConfig)
    blk_sz: int
    accel: bool = True
    accel_config: SAPAccelConfig | None = None
    power_iters: int = 10

    def __post_init__(self):
        _check_common(self)
        _is_pos_int(self.blk_sz, "blk_sz")
        _is_bool(self.accel, "accel")
        if self.accel:
            if self.accel_config is None:
                raise ValueError("accel_config must be specified if accel is True")
            if not isinstance(self.accel_config, SAPAccelConfig):
                raise TypeError(
                    f"accel_config is of type {type(self.accel_config).__name__}, "
                    "but expected type SAPAccelConfig"
                )
        _is_pos_int(self.power_iters, "power_iters")


@dataclass(kw_only=True, frozen=False)
class LSQRConfig(SolverConfig):
    """LSQR for ``LeastSquares`` models (no reference counterpart: SURVEY 8a row a14; follows the
    SolverConfig pattern of configs.py:53-107).  ``precond_config``: IdentityConfig or
    SkPreConfig (right preconditioning with the sketch-and-precondition factor)."""

    device: torch.device
    max_iters: int = 1000
    atol: float = 0.0
    rtol: float = 1e-5
    precond_config: PreconditionerConfig = field(default_factory=IdentityConfig)

    def __post_init__(self):
        _check_common(self)


def _is_solver_config(param: Any, param_name: str):
    if not isinstance(param, SolverConfig):
        raise TypeError(
            f"{param_name} is of type {type(param).__name__}, but expected type SolverConfig"
        )


def _get_solver_name(solver_config: SolverConfig) -> str:
    return {PCGConfig: "pcg", SAPConfig: "sap", LSQRConfig: "lsqr"}.get(type(solver_config))
