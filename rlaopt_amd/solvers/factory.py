"""Solver config -> solver instance (rlaopt/solvers/factory.py:38-45)."""
from typing import TYPE_CHECKING

import torch

from .configs import LSQRConfig, PCGConfig, SAPConfig, SolverConfig
from .pcg import PCG
from .lsqr import LSQR
from .sap import SAP

if TYPE_CHECKING:
    from ..models import LinSys

__all__ = ["_get_solver"]


def _get_solver(model: "LinSys", W_init: torch.Tensor, solver_config: SolverConfig):
    if type(solver_config) is PCGConfig:
        return PCG(system=model, W_init=W_init, device=solver_config.device,
                   precond_config=solver_config.precond_config)
    if type(solver_config) is SAPConfig:
        return SAP(system=model, W_init=W_init, device=solver_config.device,
                   precond_config=solver_config.precond_config, blk_sz=solver_config.blk_sz,
                   accel=solver_config.accel, accel_config=solver_config.accel_config,
                   power_iters=solver_config.power_iters)
    if type(solver_config) is LSQRConfig:
        return LSQR(system=model, W_init=W_init, device=solver_config.device,
                    precond_config=solver_config.precond_config)
    return None
