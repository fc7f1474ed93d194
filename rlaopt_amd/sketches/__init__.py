"""Dense test matrices Omega (mirrors ``rlaopt.sketches``; sketch.py:16-117, gauss.py:32-52,
ortho.py:32-56, sparse.py:32-76, factory.py:25-59).

Differences from the reference, by design:

* On a cuda device the Gaussian draw is ``rk_randn`` (Philox4x32-10 + Box-Muller in a HIP
  kernel).  Its (seed, offset) are taken from torch's default CPU generator, so
  ``torch.manual_seed`` keeps runs reproducible, and - in SPMD runs - the pair is broadcast
  from rank 0 so every rank regenerates the SAME Omega without moving it.
* The orthonormal sketch is CholeskyQR2 on the device (two f64-accumulated Gram products +
  two tall GEMMs) instead of a Householder QR; the result spans the same space and differs
  from LAPACK's Q only by column signs, to which the Nystrom approximation is invariant.
* ``"colsample"`` is this build's addition (BASELINE.json's north_star names column-sampling sketches; the
  reference's factory knows gauss / ortho / sparse only, factory.py:25-59): Omega = sqrt(d / s) [e_j1 ... e_js]
  with s distinct uniform indices (``rk_sample_distinct``).  A Omega is then s COLUMNS of A - for a kernel
  operator one ``rk_kernel_materialize(idx2=J)`` call, n s pair evaluations instead of an n x n x s product -
  and Omega^T Y a gather of s rows; the dense matrix is only built on demand.
* ``inject(...)`` feeds externally generated matrices (parity tests / benchmarks use the
  reference's own CPU draws, SURVEY.md section 7 "RNG parity").
"""

from __future__ import annotations

import contextlib
import os
from abc import ABC, abstractmethod
from collections import deque
from enum import Enum

import torch

from .. import dist as _dist
from ..linops.base import _BaseLinOp
from ..ops import f16_grid, get_backend
from ..utils import _is_pos_int, small_host_lapack

__all__ = ["get_sketch", "inject"]


class _SketchSide(Enum):
    LEFT = "left"
    RIGHT = "right"

    @classmethod
    def _from_str(cls, value, param_name):
        if isinstance(value, cls):
            return value
        if isinstance(value, str) and value.lower() in ("left", "right"):
            return cls(value.lower())
        raise ValueError(
            f"Invalid value for {param_name}: {value}. Expected 'left', 'right', "
            "_SketchSide.LEFT, or _SketchSide.RIGHT."
        )


class _SketchMode(Enum):
    GAUSS = "gauss"
    ORTHO = "ortho"
    SPARSE = "sparse"
    COLSAMPLE = "colsample"

    @classmethod
    def _from_str(cls, value, param_name):
        if isinstance(value, cls):
            return value
        if isinstance(value, str) and value.lower() in ("gauss", "ortho", "sparse", "colsample"):
            return cls(value.lower())
        raise ValueError(
            f"Invalid value for {param_name}: {value}. Expected 'gauss', 'ortho', 'sparse', 'colsample', "
            "_SketchMode.GAUSS, _SketchMode.ORTHO, _SketchMode.SPARSE, or _SketchMode.COLSAMPLE."
        )


_injected: deque = deque()


@contextlib.contextmanager
def inject(*matrices: torch.Tensor):
    """Within the block, successive ``get_sketch`` calls use these matrices (in order, each
    already in its final orientation) instead of drawing new ones."""
    _injected.extend(matrices)
    try:
        yield
    finally:
        _injected.clear()


_raw_ortho = [False]


@contextlib.contextmanager
def raw_ortho():
    """Within the block an ``Ortho`` sketch keeps its Gaussian draw un-orthonormalised and says so
    (``sk.raw``): a caller whose result depends on range(Omega) only (the block Nystrom
    preconditioner, which carries Omega^T Omega through its r x r algebra instead) saves the two
    Cholesky-QR round trips.  Injected matrices are used as given."""
    prev = _raw_ortho[0]
    _raw_ortho[0] = True
    try:
        yield
    finally:
        _raw_ortho[0] = prev


def _draw_seed() -> tuple[int, int]:
    # (single process: from the default CPU generator; one rank per GPU: the same pair on every rank)
    return tuple(_dist.shared_randint(2))


def _device_randn(rows: int, cols: int, dtype, device) -> torch.Tensor:
    seed, offset = _draw_seed()
    return get_backend().randn(rows, cols, dtype, device, seed=seed, offset=offset)


class Sketch(ABC):
    """Omega with the four products the preconditioners use (sketch.py:75-117)."""

    def __init__(self, mode, sketch_size, matrix_dim, dtype, device):
        self.mode = _SketchSide._from_str(mode, "mode")
        self.s, self.d, self.dtype, self.device = sketch_size, matrix_dim, dtype, device
        _is_pos_int(sketch_size, "sketch_size")
        want = (self.s, self.d) if self.mode is _SketchSide.LEFT else (self.d, self.s)
        self._sparse = None  # (col int32, sgn int8, row_ptr int64, scale) for sparse-sign sketches
        self.raw = False     # True: an Ortho sketch left un-orthonormalised (see raw_ortho)
        if _injected:
            Om = _injected.popleft()
            if tuple(Om.shape) != want:
                raise ValueError(f"injected sketch has shape {tuple(Om.shape)}, expected {want}")
            self._Omega = Om.to(device=device, dtype=dtype)
        else:
            self._Omega = self._generate_embedding()

    @property
    def Omega_mat(self) -> torch.Tensor:
        """The embedding as a dense tensor (built on first use for structured sketches)."""
        if self._Omega is None:
            self._Omega = self._densify()
        return self._Omega

    @Omega_mat.setter
    def Omega_mat(self, value: torch.Tensor):
        self._Omega = value

    def _densify(self) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def _generate_embedding(self) -> torch.Tensor: ...

    # Omega @ x
    def _apply_left(self, x):
        if isinstance(x, _BaseLinOp):
            return self.Omega_mat @ x  # -> x.__rmatmul__
        return get_backend().matmat(self.Omega_mat, x)

    # x @ Omega
    def _apply_right(self, x):
        if isinstance(x, _BaseLinOp):
            return x @ self.Omega_mat
        return get_backend().matmat(x, self.Omega_mat)

    # Omega.T @ x
    def _apply_left_trans(self, x):
        if isinstance(x, _BaseLinOp):
            return self.Omega_mat.T @ x
        return get_backend().rmatmat(self.Omega_mat, x)

    # x @ Omega.T
    def _apply_right_trans(self, x):
        if isinstance(x, _BaseLinOp):
            return x @ self.Omega_mat.T
        return get_backend().gemm(x, self.Omega_mat, transB=True)


class Gauss(Sketch):
    """randn(s, d) / sqrt(s), stored (d, s) for mode 'right' (gauss.py:46-52).

    Mode 'right' (the Nystrom test matrix): the entries are rounded ONCE to the float16 grid (11 significant bits) and
    every later use sees that rounded matrix - still i.i.d., symmetric, sub-Gaussian entries of variance 1 / s to 2^-22,
    i.e. a Gaussian sketch for every bound that is used on one - because the sketch product Y = K Omega on the 16-bit
    matrix cores then splits Omega into ONE float16 piece instead of two: two exact products per pair instead of three
    and half the Omega bytes (``rk_symm_gemm_rhs16``: 89 against 114 ms at n = 200 000, r = 512).  RK_SKETCH_F16_GRID=0
    keeps the unrounded draw; an injected ``Omega_mat`` is used as it is."""

    def _generate_embedding(self):
        scale = self.s**-0.5
        if self.mode is _SketchSide.RIGHT:
            # entries are i.i.d.: draw directly in the (d, s) layout the products consume
            Om = _device_randn(self.d, self.s, self.dtype, self.device) * scale
            if os.environ.get("RK_SKETCH_F16_GRID", "1") != "0" and Om.dtype in (torch.float32, torch.float64):
                Om = f16_grid(Om)
            return Om
        return _device_randn(self.s, self.d, self.dtype, self.device) * scale


class Ortho(Sketch):
    """Orthonormal basis of a Gaussian (d, s) matrix (ortho.py:48-56) via CholeskyQR2."""

    def _generate_embedding(self):
        be = get_backend()
        Q = _device_randn(self.d, self.s, self.dtype, self.device)
        if _raw_ortho[0] and self.mode is _SketchSide.RIGHT:
            self.raw = True
            return Q
        for _ in range(2):
            G = be.gram_tn(Q, Q)  # float64 (s x s)
            with small_host_lapack(self.s):
                R = torch.linalg.cholesky(G.cpu(), upper=True)
                Rinv = torch.linalg.solve_triangular(
                    R, torch.eye(self.s, dtype=torch.float64), upper=True
                )
            Q = be.gemm(Q, Rinv.to(device=self.device, dtype=self.dtype))
        return Q.T.contiguous() if self.mode is _SketchSide.LEFT else Q


class Sparse(Sketch):
    """Sparse sign embedding: zeta = min(8, s) entries of +-zeta^-1/2 per column of the (s, d)
    matrix, colliding positions overwrite (sparse.py:52-76; one of the colliding draws survives,
    here the last one).  The structure is kept in row-sorted form; ``Omega @ A`` for a tensor A
    (the SkPre update) runs as a gather of rows of A (``rk_sparse_sign_apply``: zeta passes over
    A, no dense s x d matrix).  ``Omega_mat`` densifies on demand for every other use, like
    the reference.  The draws use a torch device generator (O(zeta d) work, outside the hot path)
    seeded through ``_draw_seed`` - broadcast from rank 0 like the Gaussian sketches - so that every
    rank of a row-sharded operator builds the SAME embedding (each rank's own device RNG would
    give the ranks different Omegas and an inconsistent all-reduced Nystrom core)."""

    def _generate_embedding(self):
        s, d, dev = self.s, self.d, self.device
        zeta = 8 if s >= 8 else s
        seed, _offset = _draw_seed()
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        signs = 2 * torch.bernoulli(torch.full((zeta, d), 0.5, dtype=self.dtype, device=dev), generator=gen) - 1
        rows = torch.randint(s, (zeta, d), device=dev, generator=gen)
        keep = torch.ones((zeta, d), dtype=torch.bool, device=dev)
        for t in range(zeta - 1):  # entry t is overwritten by a later draw on the same row
            keep[t] &= ~(rows[t + 1 :] == rows[t]).any(dim=0)
        self._zeta = zeta
        tgt = rows[keep]
        cols = torch.arange(d, device=dev).expand(zeta, d)[keep]
        sg = signs[keep]
        order = torch.argsort(tgt, stable=True)
        counts = torch.bincount(tgt, minlength=s)
        row_ptr = torch.zeros(s + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(counts, 0)
        self._sparse = (cols[order].to(torch.int32), sg[order].to(torch.int8), row_ptr, zeta**-0.5)
        self._coo = (tgt, cols, sg)
        return None  # dense form on demand

    def _densify(self):
        tgt, cols, sg = self._coo
        Om = torch.zeros((self.s, self.d), dtype=self.dtype, device=self.device)
        Om[tgt, cols] = sg * self._zeta**-0.5
        return Om.T.contiguous() if self.mode is _SketchSide.RIGHT else Om

    def _apply_left(self, x):
        if (self._sparse is not None and self.mode is _SketchSide.LEFT
                and isinstance(x, torch.Tensor) and x.ndim == 2):
            col, sgn, row_ptr, scale = self._sparse
            return get_backend().sparse_sign_apply(x, col, sgn, row_ptr, scale)
        return super()._apply_left(x)


class ColSample(Sketch):
    """Uniform column sampling without replacement: the (s, d) embedding has rows sqrt(d / s) e_j, j in J, |J| = s
    distinct (the scale makes E[Omega^T Omega] = I like the other sketches; the Nystrom approximation does not
    depend on it).  ``indices`` (int64, on the device) and ``col_scale`` describe it completely; products with it
    are gathers:

        A @ Omega   (mode 'right')  = col_scale * A[:, J]      - kernel operators: K(A1, A2[J]) written directly
        Omega^T @ Y (mode 'right')  = col_scale * Y[J, :]
        Omega @ A   (mode 'left')   = col_scale * A[J, :]      - the SkPre update

    s > d is clipped to d (every column once).  An injected matrix is used as a dense sketch."""

    def __init__(self, mode, sketch_size, matrix_dim, dtype, device):
        self.indices, self.col_scale = None, 1.0
        super().__init__(mode, sketch_size, matrix_dim, dtype, device)

    def _generate_embedding(self):
        be = get_backend()
        s, d, dev = self.s, self.d, torch.device(self.device)
        take = min(s, d)
        seed, _offset = _draw_seed()
        if dev.type == "cuda" and hasattr(be, "sample_distinct"):
            J = be.sample_distinct(d, take, dev, seed)
        else:  # host stand-in backends (CPU tests)
            g = torch.Generator().manual_seed(seed)
            J = torch.randperm(d, generator=g)[:take].to(dev)
        if take < s:  # pad with repeats of the first draws: zero information, but the declared shape holds
            J = torch.cat([J, J[: s - take]])
        self.indices = J
        self.col_scale = (d / s) ** 0.5
        return None  # dense form on demand

    def _densify(self):
        Om = torch.zeros((self.s, self.d), dtype=self.dtype, device=self.device)
        Om[torch.arange(self.s, device=self.device), self.indices] = self.col_scale
        return Om.T.contiguous() if self.mode is _SketchSide.RIGHT else Om

    def columns_of(self, A, local: bool = False):
        """col_scale * A[:, J] (``local``: only the rows of this rank's shard) without forming Omega, or None when
        A offers no column access (the caller then multiplies by the dense matrix)."""
        if self.indices is None or self.mode is not _SketchSide.RIGHT:
            return None
        if isinstance(A, torch.Tensor):
            return A[:, self.indices.to(A.device)] * self.col_scale
        fn = getattr(A, "_rk_local_columns" if local else "_rk_columns", None)
        if fn is None:
            return None
        Y = fn(self.indices)
        return None if Y is None else Y * self.col_scale

    def rows_of(self, Y, lo: int = 0, hi: int | None = None):
        """Omega^T Y restricted to the rows [lo, hi) that ``Y`` holds (row p = col_scale * Y[J[p] - lo] when
        lo <= J[p] < hi, else 0): the r x r Nystrom core, float64; per-rank partials add up."""
        J = self.indices.to(Y.device)
        hi = lo + Y.shape[0] if hi is None else hi
        if Y.shape[0] == 0:  # (a rank that owns no rows still takes part in the all-reduce of the core)
            return torch.zeros((J.numel(), Y.shape[1]), dtype=torch.float64, device=Y.device)
        inside = (J >= lo) & (J < hi)
        rows = Y[(J - lo).clamp(0, max(Y.shape[0] - 1, 0))].double() * self.col_scale
        return rows * inside[:, None].to(rows.dtype)

    def _apply_right(self, x):
        if self.indices is not None and self.mode is _SketchSide.RIGHT:
            Y = self.columns_of(x)
            if Y is not None:
                return Y
        return super()._apply_right(x)

    def _apply_left(self, x):
        if (self.indices is not None and self.mode is _SketchSide.LEFT
                and isinstance(x, torch.Tensor) and x.ndim == 2):
            return x[self.indices.to(x.device)] * self.col_scale
        return super()._apply_left(x)

    def _apply_left_trans(self, x):
        if (self.indices is not None and self.mode is _SketchSide.RIGHT
                and isinstance(x, torch.Tensor) and x.ndim == 2):
            return x[self.indices.to(x.device)] * self.col_scale
        return super()._apply_left_trans(x)


SKETCH_MODE_TO_CLASS = {_SketchMode.GAUSS: Gauss, _SketchMode.ORTHO: Ortho, _SketchMode.SPARSE: Sparse,
                        _SketchMode.COLSAMPLE: ColSample}


def get_sketch(name, mode, sketch_size, matrix_dim, dtype, device) -> Sketch:
    """Factory (factory.py:25-59): name in {'gauss','ortho','sparse'} (+ 'colsample', this build's addition),
    mode in {'left','right'}."""
    cls = SKETCH_MODE_TO_CLASS[_SketchMode._from_str(name, "name")]
    return cls(mode, sketch_size, matrix_dim, dtype, device)
