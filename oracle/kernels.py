"""Oracle: kernel-matrix entries and kernel mat-mats.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates (reference paths relative to /root/reference):

* formulas              rlaopt/kernels/standard.py:31-85
* index semantics       rlaopt/kernels/base.py:88-128  (idx1 gathers rows of A1,
                        idx2 gathers rows of A2; row oracle = idx1 only, block
                        oracle = idx1 and idx2)
* constant scaling      rlaopt/linops/mixins.py:44-72 (result multiplied by
                        ``const_scaling`` after the product)
* row partitioning      rlaopt/kernels/base.py:297-307 (``torch.chunk`` of
                        ``arange(n)`` over the devices)
* distributed oracles   rlaopt/kernels/base.py:408-505, rlaopt/kernels/utils.py:10-56

The reference delegates the arithmetic to pykeops (pin 2.3, absent here); its own
test-suite pins the entries with the closed forms of tests/kernels/utils.py:4-60,
which is what ``kernel_matrix`` evaluates (vectorised, in the working dtype).
"""

from __future__ import annotations

import math

import torch

KINDS = ("rbf", "laplace", "matern12", "matern32", "matern52")


def _as_scale_row(lengthscale, d: int, dtype, device) -> torch.Tensor:
    """float or length-d tensor -> (1, 1, d) divisor (rlaopt/kernels/configs.py:50-60)."""
    if isinstance(lengthscale, torch.Tensor):
        ell = lengthscale.to(dtype=dtype, device=device).reshape(1, 1, -1)
        if ell.shape[-1] not in (1, d):
            raise ValueError("lengthscale tensor must have length d")
        return ell
    return torch.full((1, 1, 1), float(lengthscale), dtype=dtype, device=device)


def kernel_block(kind: str, X1: torch.Tensor, X2: torch.Tensor, lengthscale) -> torch.Tensor:
    """Unscaled kernel entries K[i, j] = kappa(X1[i], X2[j]) for one dense block.

    standard.py:31-35   D = (x_i - y_j) / lengthscale      (element-wise, per dim)
    standard.py:46-52   rbf      exp(-sum(D**2) / 2)
    standard.py:55-61   laplace  exp(-sum(abs(D)))
    standard.py:38-43   matern   dist = sqrt(sum(D**2))
    standard.py:64-85   matern12 exp(-dist); matern32 (1+sqrt3 dist) exp(-sqrt3 dist);
                        matern52 (1 + sqrt5 dist + 5/3 dist**2) exp(-sqrt5 dist)
    """
    if kind not in KINDS:
        raise ValueError(f"unknown kernel kind {kind!r}")
    d = X1.shape[1]
    ell = _as_scale_row(lengthscale, d, X1.dtype, X1.device)
    D = (X1[:, None, :] - X2[None, :, :]) / ell
    if kind == "laplace":
        return torch.exp(-D.abs().sum(dim=2))
    sq = (D * D).sum(dim=2)
    if kind == "rbf":
        return torch.exp(-sq / 2)
    dist = sq.sqrt()
    if kind == "matern12":
        return torch.exp(-dist)
    if kind == "matern32":
        c = math.sqrt(3.0)
        return (1 + c * dist) * torch.exp(-c * dist)
    c = math.sqrt(5.0)
    return (1 + c * dist + 5.0 / 3.0 * dist * dist) * torch.exp(-c * dist)


def kernel_matrix(kind, X1, X2, lengthscale, const_scaling: float = 1.0, row_block: int = 1024):
    """Dense const_scaling * K(X1, X2), evaluated in row blocks to bound memory."""
    out = torch.empty((X1.shape[0], X2.shape[0]), dtype=X1.dtype, device=X1.device)
    for lo in range(0, X1.shape[0], row_block):
        hi = min(lo + row_block, X1.shape[0])
        out[lo:hi] = kernel_block(kind, X1[lo:hi], X2, lengthscale)
    return out * const_scaling if const_scaling != 1.0 else out


def kernel_matmat(
    kind,
    X1,
    X2,
    lengthscale,
    V,
    const_scaling: float = 1.0,
    idx1=None,
    idx2=None,
    row_block: int = 1024,
):
    """const_scaling * K(X1[idx1], X2[idx2]) @ V, matrix-free in row blocks.

    V may be 1-D (n2,) or 2-D (n2, k): base.py:43-59 treats the trailing dimension as
    vector-valued, so mat-mat is the same map applied column-wise.
    """
    A1 = X1 if idx1 is None else X1[idx1]
    A2 = X2 if idx2 is None else X2[idx2]
    if V.shape[0] != A2.shape[0]:
        raise ValueError("V has the wrong leading dimension")
    out_shape = (A1.shape[0],) + tuple(V.shape[1:])
    out = torch.empty(out_shape, dtype=V.dtype, device=V.device)
    for lo in range(0, A1.shape[0], row_block):
        hi = min(lo + row_block, A1.shape[0])
        out[lo:hi] = kernel_block(kind, A1[lo:hi], A2, lengthscale) @ V
    # mixins.py:63-72: scaling is applied to the product, and skipped when 1.0
    return const_scaling * out if const_scaling != 1.0 else out


def kernel_rmatmat(kind, X1, X2, lengthscale, V, const_scaling: float = 1.0, row_block: int = 1024):
    """const_scaling * K(X1, X2).T @ V  (base.py:46-47); K(X1,X2).T == K(X2,X1)."""
    return kernel_matmat(kind, X2, X1, lengthscale, V, const_scaling, row_block=row_block)


# --------------------------------------------------------------------------------------
# Row partitioning and the distributed oracles
# --------------------------------------------------------------------------------------


def row_chunks(n: int, parts: int) -> list[tuple[int, int]]:
    """[lo, hi) ranges of ``torch.chunk(arange(n), parts)`` (kernels/base.py:297-302).

    torch.chunk uses ceil(n/parts)-sized chunks and may return FEWER than ``parts``
    chunks (n=5, parts=4 -> [2, 2, 1]); the reference then silently drops devices in
    its ``zip`` (kernels/base.py:305-307, 369).
    """
    if n <= 0 or parts <= 0:
        raise ValueError("n and parts must be positive")
    step = -(-n // parts)
    return [(lo, min(lo + step, n)) for lo in range(0, n, step)]


def distributed_matmat(kind, X1, X2, lengthscale, V, parts: int, const_scaling: float = 1.0):
    """Row-mode distributed product: each shard computes K(X1[rows_i], X2) @ V and the
    results are concatenated (linops/distributed.py:40-45, kernels/base.py:362-381)."""
    pieces = [
        kernel_matmat(kind, X1[lo:hi], X2, lengthscale, V, const_scaling)
        for lo, hi in row_chunks(X1.shape[0], parts)
    ]
    return torch.cat(pieces, dim=0)


def distributed_rmatmat(kind, X1, X2, lengthscale, V, parts: int, const_scaling: float = 1.0):
    """Row-mode transpose product: V is chunked by the row partition and the partial
    K(X1[rows_i], X2).T @ V[rows_i] are summed (linops/distributed.py:81-86)."""
    total = None
    for lo, hi in row_chunks(X1.shape[0], parts):
        part = kernel_rmatmat(kind, X1[lo:hi], X2, lengthscale, V[lo:hi], const_scaling)
        total = part if total is None else total + part
    return total


def distributed_row_oracle_matmat(kind, X1, X2, lengthscale, blk, V, parts: int, const_scaling=1.0):
    """Column-mode sum of K(X1[blk], X2[cols_i]) @ V[cols_i] (kernels/base.py:408-447,
    kernels/utils.py:10-29; summed by linops/distributed.py:46-50)."""
    total = None
    for lo, hi in row_chunks(X2.shape[0], parts):
        part = kernel_matmat(kind, X1[blk], X2[lo:hi], lengthscale, V[lo:hi], const_scaling)
        total = part if total is None else total + part
    return total


def distributed_blk_oracle_matmat(kind, X1, X2, lengthscale, blk, V, parts: int, const_scaling=1.0):
    """Row-mode concat of K(X1[blk[chunk_i]], X2[blk]) @ V (kernels/base.py:449-505,
    kernels/utils.py:32-56)."""
    pieces = []
    for lo, hi in row_chunks(blk.shape[0], parts):
        pieces.append(
            kernel_matmat(kind, X1[blk[lo:hi]], X2[blk], lengthscale, V, const_scaling)
        )
    return torch.cat(pieces, dim=0)
