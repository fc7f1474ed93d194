"""Identity preconditioner (rlaopt/preconditioners/identity.py)."""
from .preconditioner import Preconditioner


class Identity(Preconditioner):
    def _update(self, A, device):
        return None

    def _matmul(self, x):
        return x

    def _inverse_matmul_1d(self, x):
        return x

    def _inverse_matmul_2d(self, x):
        return x

    def _right_solve(self, x):
        return x

    def _right_solve_t(self, x):
        return x
