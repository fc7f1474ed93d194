"""Kernel-matrix operators (mirrors ``rlaopt.kernels``)."""
from . import configs, standard
from .configs import *  # noqa: F401,F403
from .standard import *  # noqa: F401,F403

__all__ = [*configs.__all__, *standard.__all__]
