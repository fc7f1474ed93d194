"""Validators and the iteration logger (mirrors ``rlaopt.utils``)."""
from . import input_checkers, logger
from .input_checkers import *  # noqa: F401,F403
from .logger import *  # noqa: F401,F403

__all__ = list(input_checkers.__all__) + list(logger.__all__)
