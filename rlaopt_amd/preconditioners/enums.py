"""Damping modes of the Nystrom preconditioner (rlaopt/preconditioners/enums.py)."""
from enum import Enum


class _DampingMode(Enum):
    ADAPTIVE = "adaptive"          # rho = baseline + smallest retained eigenvalue
    NON_ADAPTIVE = "non_adaptive"  # rho fixed

    @classmethod
    def _from_str(cls, value, param_name):
        if isinstance(value, cls):
            return value
        if isinstance(value, str):
            try:
                return cls(value.lower())
            except ValueError:
                pass
        raise ValueError(
            f"Invalid value for {param_name}: {value}. Expected 'adaptive', 'non_adaptive', "
            "_DampingMode.ADAPTIVE, or _DampingMode.NON_ADAPTIVE."
        )
