"""Config class -> preconditioner instance (rlaopt/preconditioners/factory.py:34-68)."""
from .configs import IdentityConfig, NewtonConfig, NystromConfig, PreconditionerConfig, SkPreConfig
from .identity import Identity
from .newton import Newton
from .nystrom import Nystrom
from .preconditioner import Preconditioner
from .skpre import SkPre

__all__ = ["_get_precond"]

CONFIG_TO_PRECONDITIONER = {
    IdentityConfig: Identity, NewtonConfig: Newton, NystromConfig: Nystrom, SkPreConfig: SkPre,
}


def _get_precond(precond_config: PreconditionerConfig) -> Preconditioner:
    cls = CONFIG_TO_PRECONDITIONER.get(type(precond_config))
    if cls is None:
        raise KeyError(f"No preconditioner found for configuration: {type(precond_config)}")
    return cls(precond_config)
