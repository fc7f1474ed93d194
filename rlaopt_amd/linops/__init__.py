"""Linear-operator protocol and the sharded operators (mirrors ``rlaopt.linops``)."""
from . import distributed, mixins, simple, types
from .distributed import *  # noqa: F401,F403
from .mixins import *  # noqa: F401,F403
from .simple import *  # noqa: F401,F403
from .types import *  # noqa: F401,F403

__all__ = [*distributed.__all__, *mixins.__all__, *simple.__all__, *types.__all__]
