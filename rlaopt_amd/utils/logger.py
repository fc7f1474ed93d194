"""Iteration logger (rlaopt/utils/logger.py:10-51): every ``log_freq`` iterations it calls
``log_fn`` and records ``iter_time`` (wall time since the previous log, excluding the time
spent inside the log callback) and ``cum_time``.  wandb is optional: it is imported only
when ``wandb_kwargs`` is given (the GPU boxes have no network).

``sync``: solver steps are queued on the device without any host wait, so the host clock alone would
report the time to ENQUEUE a window of iterations (~50x too small at n = 200 000) and the device work
would land inside the excluded callback window.  The model passes a device synchronisation; it runs
before each clock reading - twice per logged iteration, where the termination test reads the device
anyway - so that ``iter_time`` / ``cum_time`` are device-complete wall times like the reference's
(whose CPU / host-staged steps are synchronous by construction)."""

import time
from typing import Callable

__all__ = ["Logger"]


class Logger:
    def __init__(self, log_freq: int, log_fn: Callable, wandb_kwargs: dict | None,
                 sync: Callable | None = None):
        self._sync = sync if sync is not None else (lambda: None)
        self.log_freq = log_freq
        self.log_fn = log_fn
        self._wandb = None
        if wandb_kwargs is not None:
            import wandb  # noqa: PLC0415 - optional dependency

            self._wandb = wandb
            wandb.init(**wandb_kwargs)
        self.log_in_wandb = self._wandb is not None
        self.iter_time = 0.0
        self.cum_time = 0.0
        self._sync()
        self._tick = time.time()

    def _compute_log(self, i: int, *args, **kwargs):
        if i % self.log_freq:
            return None
        self._sync()
        self.iter_time = time.time() - self._tick
        self.cum_time += self.iter_time
        entry = {
            "iter_time": self.iter_time,
            "cum_time": self.cum_time,
            "metrics": self.log_fn(*args, **kwargs),
        }
        if self._wandb is not None:
            self._wandb.log(entry, step=i)
        self._sync()
        self._tick = time.time()  # the callback's own time is not charged to the solver
        return entry

    def _terminate(self):
        if self._wandb is not None:
            self._wandb.finish()
