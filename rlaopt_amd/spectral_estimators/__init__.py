"""Spectral estimators (mirrors ``rlaopt.spectral_estimators``).  Only
``randomized_powering`` is on the solver path (SAP step size); the reference's Hutchinson
estimators are unreachable from any solver and are not reproduced."""

import torch

__all__ = ["randomized_powering"]


def randomized_powering(A, max_iters: int = 10, rtol: float = 1e-3, v0: torch.Tensor | None = None):
    """Largest eigenvalue of a symmetric(-isable) operator by power iteration with a Rayleigh
    quotient stop rule |sig_new - sig| <= rtol * sig (spectral_norm.py:11-29).  ``v0``
    replaces the random start vector (parity runs); the default start is drawn in the
    operator's dtype (the reference draws in the DEFAULT dtype, spectral_norm.py:16, which
    breaks for float64 operators under a float32 default)."""
    d = A.shape[0]
    if v0 is None:
        dt = getattr(A, "dtype", None) or torch.get_default_dtype()
        if torch.device(A.device).type == "cuda":
            from ..sketches import _device_randn  # Philox kernel seeded from torch's generator

            v0 = _device_randn(d, 1, dt, A.device)[:, 0]
        else:
            v0 = torch.randn(d, device=A.device, dtype=dt)
    v = v0 / torch.linalg.norm(v0, 2)
    i, err, sig, sig_new = 0, float("inf"), 0.0, None
    while i < max_iters and err > rtol * sig:
        v_new = A @ v
        sig_new = torch.dot(v, v_new)
        v = v_new / torch.linalg.norm(v_new, 2)
        err = float(torch.abs(sig_new - sig))
        sig = float(sig_new)
        i += 1
    return sig_new, v
