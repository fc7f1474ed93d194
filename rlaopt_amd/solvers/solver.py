"""What the training loop (``models/model.py::Model._train``) needs from a solver: an iterate
``W`` it can read at any time and a ``_step()`` that advances it by one iteration; solvers that
precondition also expose how their preconditioner is built.  Mirrors the reference protocol
(rlaopt/solvers/solver.py:6-16), with the ``W`` property made part of the contract."""

import abc

__all__ = ["Solver"]


class Solver(abc.ABC):
    """Base class of PCG, SAP and LSQR."""

    def __init__(self, *args, **kwargs):
        """Concrete solvers take ``(system, W_init, precond_config, device, ...)``."""

    @property
    @abc.abstractmethod
    def W(self):
        """Current iterate as a full (n x k) tensor (gathered from the row shards if needed)."""

    @abc.abstractmethod
    def _get_precond(self, *args, **kwargs):
        """Build (and damp) the preconditioner for the current system / block."""

    @abc.abstractmethod
    def _step(self, *args, **kwargs):
        """One iteration on the columns that ``system.mask`` still marks active."""
