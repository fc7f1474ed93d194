"""Model base class: logging wrapper and the training loop
(rlaopt/models/model.py:30-52, 54-87, 89-114)."""

from abc import ABC, abstractmethod
from typing import Callable
from warnings import warn

import torch

from ..solvers import Solver, SolverConfig
from ..utils import Logger, quiet_host

__all__ = ["Model"]


class Model(ABC):
    def __init__(self, *args, **kwargs):
        pass

    @abstractmethod
    def _check_inputs(self, *args, **kwargs): ...

    @abstractmethod
    def _compute_internal_metrics(self, *args, **kwargs): ...

    @abstractmethod
    def _check_termination_criteria(self, *args, **kwargs): ...

    def _get_log_fn(self, callback_fn: Callable | None, callback_args: list, callback_kwargs: dict):
        """log entry = {"internal_metrics": ...} plus {"callback": ...} when a user callback
        is given; the callback runs BEFORE the internal metrics, as in the reference."""

        def log_fn(w):
            entry = {}
            if callback_fn is not None:
                entry["callback"] = callback_fn(w, self, *callback_args, **callback_kwargs)
            entry["internal_metrics"] = self._compute_internal_metrics(w)
            return entry

        return log_fn

    def _get_wandb_kwargs(self, log_in_wandb, wandb_init_kwargs, solver_name, solver_config,
                          callback_freq):
        if not log_in_wandb:
            return None
        kwargs = {
            "config": {
                "solver_name": solver_name,
                "solver_config": solver_config.to_dict(),
                "callback_freq": callback_freq,
            }
        }
        for key, value in (wandb_init_kwargs or {}).items():
            if key == "config":
                warn("Found 'config' key in wandb_init_kwargs. "
                     "Merging with internally specified 'config' key.")
                kwargs["config"].update(value)
            else:
                kwargs[key] = value
        return kwargs

    def _train(self, logger: Logger, termination_fn: Callable, solver: Solver, max_iters: int):
        """Iteration 0 and every ``log_freq``-th iteration are logged; termination is tested
        only on logged iterations, so reported iteration counts are multiples of
        ``callback_freq``."""
        with quiet_host():  # no spinning CPU workers next to a loop that syncs with the device
            log = {0: logger._compute_log(0, solver.W)}
            if termination_fn(log[0]["metrics"]["internal_metrics"]):
                return solver.W, log
            for i in range(1, max_iters + 1):
                solver._step()
                entry = logger._compute_log(i, solver.W) if i % logger.log_freq == 0 else None
                if entry is not None:
                    log[i] = entry
                    if termination_fn(entry["metrics"]["internal_metrics"]):
                        break
            logger._terminate()
            return solver.W, log

    @abstractmethod
    def solve(self, solver_config: SolverConfig, W_init: torch.Tensor, callback_fn=None,
              callback_args=[], callback_kwargs={}, callback_freq: int = 10,
              log_in_wandb: bool = False, wandb_init_kwargs: dict | None = None): ...
