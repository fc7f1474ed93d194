"""rlaopt_amd - MI355X-native randomized-preconditioned Krylov engine behind rlaopt's
LinOp / Preconditioner / Solver API.  See DESIGN.md."""

__version__ = "0.1.0"
