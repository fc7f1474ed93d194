"""Models (mirrors ``rlaopt.models``; ``LeastSquares`` is this build's addition)."""
from . import leastsquares, linsys, model
from .leastsquares import *  # noqa: F401,F403
from .linsys import *  # noqa: F401,F403
from .model import *  # noqa: F401,F403

__all__ = [*leastsquares.__all__, *linsys.__all__, *model.__all__]
