"""Models (mirrors ``rlaopt.models``)."""
from . import linsys, model
from .linsys import *  # noqa: F401,F403
from .model import *  # noqa: F401,F403

__all__ = [*linsys.__all__, *model.__all__]
