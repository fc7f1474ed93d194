"""rlaopt_amd - MI355X-native randomized-preconditioned Krylov engine behind rlaopt's
LinOp / Preconditioner / Solver API.

Sub-packages mirror the reference's public surface:
``linops, kernels, sketches, preconditioners, solvers, models, spectral_estimators, utils``.
All n-length arithmetic runs in ``librk_hip.so`` (``include/rk.h``); importing the package
does not need a GPU, using it does.
"""

__version__ = "0.1.0"

from . import (  # noqa: F401,E402
    kernels, linops, models, preconditioners, sketches, solvers, spectral_estimators, utils,
)
