"""Preconditioners (mirrors ``rlaopt.preconditioners``)."""
from . import configs, factory, preconditioner
from .configs import *  # noqa: F401,F403
from .factory import *  # noqa: F401,F403
from .preconditioner import *  # noqa: F401,F403

__all__ = [*configs.__all__, *factory.__all__, *preconditioner.__all__]
