"""Oracle: dense test matrices Omega.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/sketches/gauss.py:32-52, ortho.py:32-56, sparse.py:32-76 and the apply
rules of rlaopt/sketches/sketch.py:75-117.  The draws use the same torch calls in the
same order as the reference, so that with the same seed on the CPU the embeddings are
bit-identical to the reference's (checked by tests/test_oracle_golden.py).
"""

from __future__ import annotations

import torch


def gauss(mode: str, sketch_size: int, matrix_dim: int, dtype, device="cpu", generator=None):
    """randn(s, d) / sqrt(s); transposed to (d, s) for mode 'right' (gauss.py:46-52)."""
    Om = torch.randn(sketch_size, matrix_dim, dtype=dtype, device=device, generator=generator)
    Om = Om / sketch_size**0.5
    if mode == "right":
        Om = Om.T
    return Om.contiguous()


def ortho(mode: str, sketch_size: int, matrix_dim: int, dtype, device="cpu", generator=None):
    """Q factor of reduced QR of randn(d, s); transposed for mode 'left' (ortho.py:48-56)."""
    G = torch.randn(matrix_dim, sketch_size, dtype=dtype, device=device, generator=generator)
    Q = torch.linalg.qr(G, mode="reduced")[0]
    if mode == "left":
        Q = Q.T
    return Q.contiguous()


def sparse_sign(mode: str, sketch_size: int, matrix_dim: int, dtype, device="cpu", generator=None):
    """zeta = min(8, s) signed entries per column of the (s, d) matrix, placed with
    ``scatter_`` (collisions overwrite), scaled by zeta**-0.5; transposed (NOT made
    contiguous) for mode 'right' (sparse.py:52-76)."""
    zeta = 8 if sketch_size >= 8 else sketch_size
    Om = torch.zeros((sketch_size, matrix_dim), dtype=dtype, device=device)
    b = torch.bernoulli(
        0.5 * torch.ones((zeta, matrix_dim), dtype=dtype, device=device), generator=generator
    )
    signs = 2 * b - 1
    rows = torch.randint(sketch_size, (zeta, matrix_dim), device=device, generator=generator)
    Om.scatter_(0, rows, signs)
    Om = Om * zeta**-0.5
    if mode == "right":
        Om = Om.T
    return Om


_GENERATORS = {"gauss": gauss, "ortho": ortho, "sparse": sparse_sign}


def get_sketch_matrix(name, mode, sketch_size, matrix_dim, dtype, device="cpu", generator=None):
    """Name dispatch of rlaopt/sketches/factory.py:25-59 (case-insensitive names)."""
    key = name.lower() if isinstance(name, str) else name
    if key not in _GENERATORS:
        raise ValueError(f"Invalid value for name: {name}")
    if mode not in ("left", "right"):
        raise ValueError(f"Invalid value for mode: {mode}")
    return _GENERATORS[key](mode, sketch_size, matrix_dim, dtype, device, generator)
