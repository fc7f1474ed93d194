"""Solvers (mirrors ``rlaopt.solvers``)."""
from . import configs, factory, solver
from .configs import *  # noqa: F401,F403
from .factory import *  # noqa: F401,F403
from .solver import *  # noqa: F401,F403

__all__ = [*configs.__all__, *factory.__all__, *solver.__all__]
