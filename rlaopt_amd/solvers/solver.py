"""Solver protocol (rlaopt/solvers/solver.py:6-16)."""
from abc import ABC, abstractmethod

__all__ = ["Solver"]


class Solver(ABC):
    def __init__(self, *args, **kwargs):
        pass

    @abstractmethod
    def _get_precond(self, *args, **kwargs): ...

    @abstractmethod
    def _step(self, *args, **kwargs): ...
