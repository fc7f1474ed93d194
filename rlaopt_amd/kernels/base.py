"""Kernel-matrix operators (mirrors rlaopt/kernels/base.py:23-520).

The reference builds a symbolic pykeops ``LazyTensor`` and lets pykeops JIT a fused
map-reduce for every product (base.py:43-47, 88-101, 183-214).  On MI355X the operator
has two execution modes, both through ``librk_hip.so``:

materialised (default when the block fits the device's HBM budget)
    K(A1, A2) - const_scaling folded in - is written once by ``rk_kernel_materialize`` and
    every ``A @ x`` streams it with ``rk_matmat_skinny`` (HBM-bound, the roofline kernel);
    288 GB of HBM3E hold a 200 000 x 200 000 fp32 block (160 GB) on ONE device.
matrix-free
    ``rk_kernel_matmat`` recomputes the entries tile by tile (compute-bound) - used when
    the block does not fit and for the row / block oracles whose index sets change per call
    (base.py:104-128, kernels/utils.py:10-56).

symmetric packed ("sympack", default when A1 is A2 and the triangle fits)
    only the strips on and above the diagonal are stored (half the bytes, see
    ``csrc/rk_symm.hip``) and ``rk_symm_matmat`` uses every stored entry for both y[i] and
    y[j]: ~2x fewer HBM bytes per product.  Across ranks the strips are dealt cyclically and
    the per-rank partial products are all-reduced.  Products with many columns (the Nystrom
    sketch) go through temporary row-major panels and the matrix cores.

``RK_KERNEL_MODE`` = "auto" | "sympack" | "materialize" | "matrix_free" overrides the choice.

``RK_STORAGE_DTYPE=float32`` (sympack mode, float64 data): keep the packed entries in float32
while every vector and every accumulation stays float64 - the product streams fp32 bytes, and
the Krylov recurrences keep float64 accuracy.  The operator is then K rounded entry-wise to
float32 (relative perturbation 6e-8); a pure float32 solve of the n = 200 000 benchmark system
stagnates near 2e-4 relative residual (eps_32 x condition number), this mode reaches 1e-6 at
the float32 iteration cost.  Off by default: results then match the reference dtype for dtype.
"""

from __future__ import annotations

import os
from typing import Any

import torch

from .. import dist as _dist
from ..linops import LinOp, ScaleMixin, TwoSidedLinOp
from ..linops.base import _RemoteShard
from ..linops.distributed import _DistributedLinOp, _DistributedTwoSidedLinOp
from ..ops import get_backend, on_f16_grid
from ..utils import _is_set, _is_torch_tensor
from .configs import KernelConfig, _is_kernel_config

_HBM_FRACTION = 0.85  # of the currently free device memory a materialised block may take


def _mode_override() -> str:
    mode = os.environ.get("RK_KERNEL_MODE", "auto").lower()
    if mode not in ("auto", "sympack", "materialize", "matrix_free"):
        raise ValueError(f"RK_KERNEL_MODE={mode!r}: expected auto|sympack|materialize|matrix_free")
    return mode


def _storage_dtype(dtype: torch.dtype) -> torch.dtype:
    want = os.environ.get("RK_STORAGE_DTYPE", "").lower()
    if want in ("", "same"):
        return dtype
    if want in ("float32", "f32"):
        return torch.float32 if dtype == torch.float64 else dtype
    raise ValueError(f"RK_STORAGE_DTYPE={want!r}: expected float32 or unset")


def _fits_in_hbm(rows: int, cols: int, dtype: torch.dtype, device: torch.device) -> bool:
    need = rows * cols * torch.empty((), dtype=dtype).element_size()
    if device.type != "cuda":
        return True  # host stand-in backends (tests) have no HBM budget
    free, _total = torch.cuda.mem_get_info(device)
    return need <= _HBM_FRACTION * free


def _same_points(A1: torch.Tensor, A2: torch.Tensor) -> bool:
    return A1 is A2 or (
        A1.device == A2.device and A1.data_ptr() == A2.data_ptr() and A1.shape == A2.shape
        and A1.stride() == A2.stride()
    )


_UNSET = object()


def _prefer_matrix_free(kind: str, X: torch.Tensor, expected_products: float) -> bool:
    """Mode choice by modelled end-to-end time (RK_KERNEL_MODE=auto, symmetric float32 data on one device, nothing
    materialised yet): strip-packed storage streams 2 n^2 bytes per product at ~6.9 TB/s but has to be allocated -
    memory that an earlier process freed is cleared by the driver before it is handed out, ~15 ms per GB, paid by
    roughly every third fresh process - and written (~1 ms per GB); the matrix-free product
    recomputes the entries on the 16-bit matrix cores at ~3.1e12 pairs/s (11 % slower per product at d <= 64), needs no
    storage, and its sketch product costs ~0.09 s more at n = 200 000.  With the expected allocation price
    RK_ALLOC_MS_PER_GB (default 5 = 15 / 3; 0: never prefer matrix-free) the storage pays off from ~300 products on,
    whatever n.  Memory the caching allocator already holds costs nothing."""
    if X.dtype != torch.float32 or not X.is_cuda or X.shape[1] > 64 or kind not in ("rbf", "matern32", "matern52"):
        return False
    per_gb = float(os.environ.get("RK_ALLOC_MS_PER_GB", "5"))
    if per_gb <= 0.0 or expected_products is None:
        return False
    n = X.shape[0]
    gb = 2.0 * n * n / 1e9
    cached_free = torch.cuda.memory_reserved(X.device) - torch.cuda.memory_allocated(X.device)
    alloc_ms = 0.0 if cached_free >= gb * 1e9 else per_gb * gb
    t_pack = alloc_ms + 1.0 * gb + expected_products * (gb / 6.9)           # ms
    t_free = 0.09e3 * (n / 2e5) ** 2 + expected_products * (n * n / 3.1e9)  # ms (extra sketch time + products)
    return t_free < t_pack


def _gram_center(kind: str, A1: torch.Tensor, A2: torch.Tensor, lengthscale, device=None):
    """Centre for the matrix-core formulation of matrix-free products over (subsets of) the rows
    of A1 and A2, or None (``HipBackend.gram_center``; one host read-back, so operators cache the
    answer - the point sets of an operator never change)."""
    if kind in ("laplace", "matern12") or A1.dtype != torch.float32 or A1.shape[1] > 128:
        return None
    be = get_backend()
    if not hasattr(be, "gram_center"):
        return None
    dev = device if device is not None else A2.device
    if torch.device(dev).type != "cuda":
        return None
    same = _same_points(A1, A2)
    A2d = A2.to(dev)
    A1d = A2d if same else A1.to(dev)
    ell = lengthscale.to(dev) if isinstance(lengthscale, torch.Tensor) else lengthscale
    return be.gram_center(A1d, A2d, ell)


_FREE_WIDE_COLS = 96     # matrix-free operators: beyond this many columns panels + GEMM beat the fused product
_FREE_PANEL_BYTES = 8 << 30  # their temporary panel (a matrix-free operator is chosen to keep allocations small)
_PANEL_BYTES = 16 << 30  # temporary row-major panel for wide products in sympack mode (>= 512 GEMM tiles)
_BLK_MATERIALIZE_BYTES = 1 << 30  # block oracles up to this size are materialised (SAP blocks)
_SYM_MAX_COLS = 16      # beyond this many columns a product goes through panels + MFMA


class _SymPackBlock:
    """Strip-packed K(X, X) (strips first, first + stride, ...) on one device."""

    def __init__(self, kind: str, X: torch.Tensor, config: KernelConfig, first: int = 0, stride: int = 1):
        self.kind, self.X, self.config = kind, X, config
        self.n, self.first, self.stride = X.shape[0], first, stride
        self.scale = float(getattr(config, "const_scaling", 1.0))
        self.store_dtype = _storage_dtype(X.dtype)
        self._Kp = None
        self._panel = None
        self._X_store = None

    @staticmethod
    def fits(n: int, stride: int, dtype, device) -> bool:
        es = torch.empty((), dtype=_storage_dtype(dtype)).element_size()
        need = n * n * es // (2 * stride) + _PANEL_BYTES
        if device.type != "cuda":
            return True
        free, _total = torch.cuda.mem_get_info(device)
        return need <= _HBM_FRACTION * free

    @property
    def Kp(self) -> torch.Tensor:
        if self._Kp is None:
            self._Kp = get_backend().sympack_materialize(
                self.kind, self.X, self.config.lengthscale, scale=self.scale,
                first=self.first, stride=self.stride, store_dtype=self.store_dtype,
            )
        return self._Kp

    def partial_matmat(self, x, diag_shift=0.0):
        """This rank's partial of K x (+ diag_shift x); complete when stride == 1."""
        return get_backend().symm_matmat(
            self.Kp, self.n, x, diag_shift=diag_shift, first=self.first, stride=self.stride,
            tag="kernel_shard_product", kmax=abs(self.scale) if self.scale != 0.0 else None,
        )

    def gemm_ready(self, x) -> bool:
        """Wide products straight from the resident strips (``rk_symm_gemm``): float32 storage, a 2-D block."""
        return (self.store_dtype == torch.float32 and x.ndim == 2 and x.dtype in (torch.float32, torch.float64)
                and self.scale != 0.0 and os.environ.get("RK_SYMGEMM", "1") != "0")

    def partial_matmat_wide(self, x):
        """This rank's partial of K x (all n rows) for many columns, from the packed strips in place on the
        16-bit matrix cores - no row-major panels, no second evaluation of the entries; complete when
        stride == 1.  float64 x over float32 storage: the product runs in float32 (as the panel route does)."""
        be = get_backend()
        grid = on_f16_grid(x)  # (a test matrix drawn on the float16 grid: one piece, two products per pair)
        xs = x if x.dtype == torch.float32 else x.float()
        outs = [be.symm_gemm(self.Kp, self.n, xs[:, c0 : c0 + 1024], kmax=abs(self.scale), first=self.first,
                             stride=self.stride, tag="kernel_sketch_product", rhs_f16=grid)
                for c0 in range(0, xs.shape[1], 1024)]
        out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
        return out.to(x.dtype)

    def rows_matmat_wide(self, x, lo: int, hi: int):
        """Rows [lo, hi) of K x for many columns: temporary row-major panels + GEMM (in the
        storage dtype: with float32 storage the sketch product runs on the fp32 matrix cores)."""
        be = get_backend()
        out_dtype = x.dtype
        X = self.X
        ell = self.config.lengthscale
        if self.store_dtype != x.dtype:
            if self._X_store is None:
                self._X_store = self.X.to(self.store_dtype)
            X, x = self._X_store, x.to(self.store_dtype)
            if isinstance(ell, torch.Tensor):
                ell = ell.to(self.store_dtype)
        es = x.element_size()
        budget = _PANEL_BYTES
        if x.is_cuda:
            budget = min(budget, int(0.5 * torch.cuda.mem_get_info(x.device)[0]))
        rpp = max(256, min(hi - lo, budget // (self.n * es)))
        # 128 x 128 GEMM tiles, 4 column tiles for a rank-512 sketch: a multiple of 8192 rows gives
        # a multiple of 256 tiles = whole rounds of the 256 CUs (no partial last round)
        if rpp >= 8192:
            rpp -= rpp % 8192
        if self._panel is None or self._panel.shape[0] < min(rpp, hi - lo):
            self._panel = torch.empty((min(rpp, hi - lo), self.n), dtype=x.dtype, device=x.device)
        out = torch.empty((hi - lo, x.shape[1]), dtype=x.dtype, device=x.device)
        for a in range(lo, hi, rpp):
            b = min(hi, a + rpp)
            panel = be.kernel_materialize(
                self.kind, X[a:b], X, ell, scale=self.scale, out=self._panel[: b - a],
            )
            out[a - lo : b - lo] = be.matmat(panel, x)
        self._panel = None  # the panel is only needed while a sketch is being built
        return out.to(out_dtype)

    def release(self):
        self._Kp = None
        self._panel = None


class _KernelBlock:
    """K(A1[rows], A2) on one device: lazily materialised, or matrix-free."""

    def __init__(self, kind: str, A1: torch.Tensor, A2: torch.Tensor, config: KernelConfig):
        self.kind, self.A1, self.A2, self.config = kind, A1, A2, config
        self.scale = float(getattr(config, "const_scaling", 1.0))
        self._K = None
        self._gc = _UNSET
        self._symmetric = A1 is A2 or (
            A1.data_ptr() == A2.data_ptr() and A1.shape == A2.shape and A1.stride() == A2.stride()
        )
        mode = _mode_override()
        if mode == "auto":
            self.materialised = _fits_in_hbm(A1.shape[0], A2.shape[0], A1.dtype, A1.device)
        else:
            self.materialised = mode == "materialize"

    @property
    def K(self) -> torch.Tensor:
        if self._K is None:
            self._K = get_backend().kernel_materialize(
                self.kind, self.A1, self.A2, self.config.lengthscale, scale=self.scale
            )
        return self._K

    @property
    def gram_center(self):
        if self._gc is _UNSET:
            self._gc = _gram_center(self.kind, self.A1, self.A2, self.config.lengthscale)
        return self._gc

    def matmat(self, x, diag_shift=0.0, row0=0, alpha=1.0, beta=0.0, C=None):
        be = get_backend()
        if self.materialised:
            return be.matmat(self.K, x, alpha=alpha, beta=beta, C=C, diag_shift=diag_shift, row0=row0,
                             tag="kernel_shard_product")
        cols = 1 if x.ndim == 1 else x.shape[1]
        if cols > _FREE_WIDE_COLS:
            # many columns (the Nystrom sketch): the fused product recomputes the kernel entries once per
            # 16 columns (36 ms each at n = 200 000); beyond ~100 columns temporary row-major panels and
            # the matrix-core GEMM are cheaper (0.19 s of entries + 0.7 ms per column)
            y = self._matmat_wide(x)
        else:
            y = be.kernel_matmat(self.kind, self.A1, self.A2, self.config.lengthscale, x, scale=self.scale,
                                 gram_center=self.gram_center)
        if diag_shift != 0.0:
            y = y + diag_shift * x[row0 : row0 + y.shape[0]]
        if alpha != 1.0 or beta != 0.0:
            y = alpha * y + (beta * C if beta != 0.0 else 0.0)
        return y

    def _matmat_wide_strips(self, x):
        """K(X, X) @ x for MANY columns without n x n storage and without the float32 GEMM: the packed triangle is
        written P cyclic strip subsets at a time into ONE temporary buffer (<= _FREE_PANEL_BYTES; subset f = strips
        f, f + P, ... - the layout a rank of a P-rank run holds) and each subset's partial of K x comes from the
        16-bit matrix cores (``rk_symm_gemm``); the partials add up.  Every entry is evaluated once (the row-major
        panels evaluate the full square) and the product runs at the sketch kernel's rate instead of the float32
        matrix cores': n = 200 000, r = 512: see DESIGN.md (0.77 s through panels).  Writing subset f + 1 on a second
        stream beside subset f's product (two buffers) was measured: 0.29 against 0.26 s for the assembly - the two
        kernels compete for the same CUs; not kept."""
        be = get_backend()
        n = self.A1.shape[0]
        budget = 2 * _FREE_PANEL_BYTES  # (16 GB: five subsets at n = 200 000; every subset's product ends in a ~6 ms tail)
        if x.is_cuda:
            budget = min(budget, int(0.5 * torch.cuda.mem_get_info(x.device)[0]))
        parts = max(1, -(-(n * n * 2 + n * 1024) // max(budget, 1)))  # packed float32 triangle = 2 n^2 bytes (+ diagonal blocks)
        parts = min(parts, max(1, -(-n // 256)))
        cap = max(be.sympack_elems(n, f, parts) for f in range(parts))
        buf = torch.empty(cap, dtype=torch.float32, device=x.device)
        grid = on_f16_grid(x)
        xs = x if x.dtype == torch.float32 else x.float()
        out = None
        for f in range(parts):
            Kp = be.sympack_materialize(self.kind, self.A1, self.config.lengthscale, scale=self.scale, first=f,
                                        stride=parts, out=buf, gram_center=self.gram_center)
            for c0 in range(0, xs.shape[1], 1024):
                part = be.symm_gemm(Kp, n, xs[:, c0 : c0 + 1024], kmax=abs(self.scale), first=f, stride=parts,
                                    tag="kernel_sketch_product", rhs_f16=grid)
                if out is None:
                    out = torch.zeros((n, xs.shape[1]), dtype=torch.float32, device=x.device)
                out[:, c0 : c0 + 1024] += part
        return out.to(x.dtype)

    def _matmat_wide(self, x):
        be = get_backend()
        m, n = self.A1.shape[0], self.A2.shape[0]
        if (self.A1.dtype == torch.float32 and x.is_cuda and x.ndim == 2 and x.dtype in (torch.float32, torch.float64)
                and self.A1.shape[1] <= 64 and self.kind in ("rbf", "matern32", "matern52") and self.scale != 0.0
                and hasattr(be, "kernel_sketch_matmat") and self.gram_center is not None
                and os.environ.get("RK_FREE_WIDE_FUSED", "1") != "0"):
            # the kernel entries are evaluated inside the product (flash-style): no storage at all, rectangular
            # shards included
            xs = x if x.dtype == torch.float32 else x.float()
            return be.kernel_sketch_matmat(self.kind, self.A1, self.A2, self.config.lengthscale, xs, scale=self.scale,
                                           gram_center=self.gram_center, tag="kernel_sketch_product",
                                           rhs_f16=on_f16_grid(x)).to(x.dtype)
        if (self._symmetric and m == n and self.A1.dtype == torch.float32 and x.is_cuda and x.ndim == 2
                and self.scale != 0.0 and hasattr(be, "symm_gemm") and hasattr(be, "sympack_elems")
                and os.environ.get("RK_FREE_WIDE_STRIPS", "1") != "0"):
            return self._matmat_wide_strips(x)
        es = x.element_size()
        budget = _FREE_PANEL_BYTES
        if x.is_cuda:
            budget = min(budget, int(0.5 * torch.cuda.mem_get_info(x.device)[0]))
        rpp = max(256, min(m, budget // (n * es)))
        if rpp >= 8192:
            rpp -= rpp % 8192  # whole rounds of 128 x 128 GEMM tiles (see _SymPackBlock.rows_matmat_wide)
        panel = torch.empty((min(rpp, m), n), dtype=x.dtype, device=x.device)
        out = torch.empty((m, x.shape[1]), dtype=x.dtype, device=x.device)
        for a in range(0, m, rpp):
            b = min(m, a + rpp)
            K = be.kernel_materialize(self.kind, self.A1[a:b], self.A2, self.config.lengthscale,
                                      scale=self.scale, out=panel[: b - a])
            out[a:b] = be.matmat(K, x)
        return out

    def rmatmat(self, x):
        be = get_backend()
        if self.materialised:
            if self._symmetric and self.A1.shape[0] == self.A2.shape[0]:
                return be.matmat(self.K, x)
            return be.rmatmat(self.K, x)
        # K(A1, A2)^T = K(A2, A1)
        return be.kernel_matmat(self.kind, self.A2, self.A1, self.config.lengthscale, x, scale=self.scale,
                                gram_center=self.gram_center)

    def release(self):
        self._K = None


def _check_data(A1: Any, A2: Any, kernel_config: Any, same_device: bool):
    _is_torch_tensor(A1, "A1")
    _is_torch_tensor(A2, "A2")
    if A1.ndim != 2:
        raise ValueError(f"A1 must be a 2D tensor, got {A1.ndim}D tensor.")
    if A2.ndim != 2:
        raise ValueError(f"A2 must be a 2D tensor, got {A2.ndim}D tensor.")
    if same_device and A1.device != A2.device:
        raise ValueError("A1 and A2 must be on the same device.")
    if A1.dtype != A2.dtype:
        raise ValueError("A1 and A2 must have the same dtype.")
    _is_kernel_config(kernel_config, "kernel_config")


class _KernelLinOp(TwoSidedLinOp, ScaleMixin):
    """Single-device kernel operator K(A1, A2) with row / block oracles."""

    def __init__(self, A1, A2, kernel_config: KernelConfig, _kernel_kind: str):
        _check_data(A1, A2, kernel_config, same_device=True)
        self._A1, self._A2, self._kernel_config, self._kind = A1, A2, kernel_config, _kernel_kind
        self._initialize_scaling(getattr(kernel_config, "const_scaling", 1.0))
        mode = _mode_override()
        self._sym = None
        if _same_points(A1, A2) and mode in ("auto", "sympack") and (
            mode == "sympack" or _SymPackBlock.fits(A1.shape[0], 1, A1.dtype, A1.device)
        ):
            self._sym = _SymPackBlock(_kernel_kind, A1, kernel_config)
            fwd = bwd = self._sym_matmat
            self._block = None
        else:
            self._block = _KernelBlock(_kernel_kind, A1, A2, kernel_config)
            fwd, bwd = self._block.matmat, self._block.rmatmat
        super().__init__(
            device=A1.device,
            shape=torch.Size((A1.shape[0], A2.shape[0])),
            matvec=fwd, rmatvec=bwd, matmat=fwd, rmatmat=bwd,
            dtype=A1.dtype,
        )

    A1 = property(lambda self: self._A1)
    A2 = property(lambda self: self._A2)
    kernel_config = property(lambda self: self._kernel_config)

    @property
    def storage_mode(self) -> str:
        if self._sym is not None:
            return "sympack" if self._sym.store_dtype == self.dtype else "sympack(float32 storage)"
        return "materialize" if self._block.materialised else "matrix_free"

    def _rk_plan(self, expected_products: float) -> None:
        """Hint from a solver about how many products are coming (LinSys.solve: iterations + logged residuals +
        the sketch): in auto mode, before anything has been materialised, the operator may decide to stay
        storage-free (``_prefer_matrix_free``)."""
        if (_mode_override() == "auto" and self._sym is not None and self._sym._Kp is None
                and _gram_center(self._kind, self._A1, self._A2, self._kernel_config.lengthscale) is not None
                and _prefer_matrix_free(self._kind, self._A1, expected_products)):
            self._sym = None
            self._block = _KernelBlock(self._kind, self._A1, self._A2, self._kernel_config)
            self._block.materialised = False
            self._matvec_fn = self._matmat_fn = self._block.matmat
            self._rmatvec_fn = self._rmatmat_fn = self._block.rmatmat

    def _sym_matmat(self, x, diag_shift=0.0):
        cols = 1 if x.ndim == 1 else x.shape[1]
        if cols <= _SYM_MAX_COLS:
            return self._sym.partial_matmat(x, diag_shift=diag_shift)
        if self._sym.gemm_ready(x):
            y = self._sym.partial_matmat_wide(x)  # (one device: stride 1, the partial is the product)
        else:
            y = self._sym.rows_matmat_wide(x, 0, self.shape[0])
        return y + diag_shift * x if diag_shift != 0.0 else y

    # fused "A @ X + shift * X" used by the solvers (saves one pass over X)
    def _rk_shifted_matmat(self, X, shift):
        if self._sym is not None:
            return self._sym_matmat(X, diag_shift=shift)
        return self._block.matmat(X, diag_shift=shift, row0=0)

    def _rk_columns(self, J: torch.Tensor) -> torch.Tensor:
        """K[:, J] as a dense n1 x |J| block, written directly from the closed form (n1 |J| pair evaluations):
        what a column-sampling sketch's ``A @ Omega`` is (sketches.ColSample)."""
        return get_backend().kernel_materialize(
            self._kind, self._A1, self._A2, self._kernel_config.lengthscale, scale=self._scaling,
            idx2=J.to(self.device))

    _rk_local_columns = _rk_columns

    def _get_kernel_linop(self, idx1=None, idx2=None) -> LinOp:
        """Matrix-free K(A1[idx1], A2[idx2]) (base.py:104-128); scaling applied once."""
        be, cfg = get_backend(), self._kernel_config
        i1 = None if idx1 is None else idx1.to(self.device)
        i2 = None if idx2 is None else idx2.to(self.device)
        rows = self._A1.shape[0] if idx1 is None else idx1.shape[0]
        cols = self._A2.shape[0] if idx2 is None else idx2.shape[0]
        if getattr(self, "_gc", _UNSET) is _UNSET:  # once per operator: subsets inherit the bound
            self._gc = _gram_center(self._kind, self._A1, self._A2, cfg.lengthscale)
        gc = self._gc

        def matvec(x):
            return be.kernel_matmat(
                self._kind, self._A1, self._A2, cfg.lengthscale, x, scale=self._scaling,
                idx1=i1, idx2=i2, gram_center=gc,
            )

        return LinOp(
            device=self.device, shape=torch.Size((rows, cols)), matvec=matvec, matmat=matvec,
            dtype=self.dtype,
        )

    def row_oracle(self, blk: torch.Tensor) -> LinOp:
        return self._get_kernel_linop(idx1=blk)

    def blk_oracle(self, blk: torch.Tensor) -> LinOp:
        return self._get_kernel_linop(idx1=blk, idx2=blk)


class _KernelShard(TwoSidedLinOp):
    """Row shard K(A1[rows_i], A2) resident on one device (the role of the reference's
    ``_CacheableKernelLinOp``, base.py:131-225, without per-call pickling or printing)."""

    def __init__(self, kind, A1_rows, A2, kernel_config, device):
        self._block = _KernelBlock(kind, A1_rows.to(device), A2.to(device), kernel_config.to(device))
        super().__init__(
            device=device,
            shape=torch.Size((A1_rows.shape[0], A2.shape[0])),
            matvec=self._block.matmat,
            rmatvec=self._block.rmatmat,
            matmat=self._block.matmat,
            rmatmat=self._block.rmatmat,
            dtype=A1_rows.dtype,
        )

    def _clear_cache(self):
        self._block.release()


def _oracle_only(_x):
    raise RuntimeError("this operator was built with use_full_kernel=False: only "
                       "row_oracle / blk_oracle are available")


class _UnavailableShard(TwoSidedLinOp):
    """Shard of an operator built with ``use_full_kernel=False`` (base.py:383-406): a
    ``TwoSidedLinOp`` placeholder, as in the reference; only the oracles are usable (the
    reference's placeholder returns ``None`` from every product and fails later in
    ``torch.cat`` - here the product raises a clear error)."""

    def __init__(self, device, shape, dtype):
        super().__init__(device, shape, _oracle_only, _oracle_only, _oracle_only, _oracle_only, dtype)


def _sorted_devices(devices) -> list[torch.device]:
    # a set has hash order (base.py:356); sort for a deterministic shard <-> device map
    return sorted(devices, key=lambda d: (d.type, -1 if d.index is None else d.index))


class _DistributedKernelLinOp(_DistributedTwoSidedLinOp, ScaleMixin):
    """Row-sharded kernel operator with distributed row / block oracles."""

    def __init__(self, A1, A2, kernel_config, devices, use_full_kernel, _kernel_kind, group=None):
        self._check_inputs(A1, A2, kernel_config, devices)
        self._kind = _kernel_kind
        self._A1, self._A2, self._kernel_config = A1, A2, kernel_config
        self._initialize_scaling(getattr(kernel_config, "const_scaling", 1.0))
        devs = _sorted_devices(devices)
        rank, size = _dist.world(group)
        spmd = _dist.spmd_active(group)
        # the reference's own multi-GPU call: one process, devices={cuda:0 .. cuda:p-1} (rlaopt/kernels/factory.py:34-41)
        self._inprocess_multi_device = (not spmd
                                        and len({(d.type, d.index) for d in devs if d.type == "cuda"}) > 1)
        if self._inprocess_multi_device:
            import warnings

            warnings.warn(
                f"rlaopt_amd: {len(devs)} devices in ONE process.  LinSys.solve runs such a system as one fresh worker "
                "process per device (rlaopt_amd/spmd.py: one rank per GPU over RCCL, strip-packed storage, state "
                "row-partitioned) and returns rank 0's result - unless a user callback or RK_MULTI_DEVICE=inprocess "
                "asks for the in-process form, whose shards are driven from this process's launch queues one after "
                "another, without the packed storage and without RCCL.  Neither form has run on more than one physical "
                "GPU yet.  Scripts that start one rank per GPU themselves (python -m torch.distributed.run "
                "--nproc-per-node P ...) pass this rank's device: the operator then shards itself over the group.",
                RuntimeWarning, stacklevel=3)
        parts = size if spmd else len(devs)
        n1, n2 = A1.shape[0], A2.shape[0]
        # base.py:297-302: torch.chunk partition of the rows of A1 (operator) and A2 (oracle)
        self._A1_bounds = _dist.row_chunks(n1, parts)
        self._A2_bounds = _dist.row_chunks(n2, parts)
        self._row_x2_caches = {}  # chunk index -> converted image of that A2 chunk (row-oracle products)
        self.A1_row_chunks = [torch.arange(lo, hi) for lo, hi in self._A1_bounds]
        self.A2_row_chunks = [torch.arange(lo, hi) for lo, hi in self._A2_bounds]
        if spmd:
            mine = self._my_device(devs, A1)
            self._shard_devices = [mine] * parts
        else:
            self._shard_devices = devs  # zip() below drops surplus devices like the reference
        self._kernel_config_devices = {d: kernel_config.to(d) for d in set(self._shard_devices)}

        def owned(i):
            return (not spmd) or i == rank

        # A2 column blocks for the row oracle stay resident on their device (base.py:305-307)
        self.A2_chunks = [
            A2[lo:hi].to(dev) if owned(i) else None
            for i, ((lo, hi), dev) in enumerate(zip(self._A2_bounds, self._shard_devices))
        ]
        # symmetric data + one shard per process: strip-packed triangle, strips dealt cyclically
        mode = _mode_override()
        nshards = len(self._A1_bounds)
        my_dev = self._shard_devices[rank if spmd else 0] if nshards else None
        self._sym = None
        if (use_full_kernel and _same_points(A1, A2) and mode in ("auto", "sympack")
                and (spmd or nshards == 1)
                and (mode == "sympack" or _SymPackBlock.fits(n1, size if spmd else 1, A1.dtype, my_dev))):
            self._sym = _SymPackBlock(
                _kernel_kind, A1.to(my_dev), kernel_config.to(my_dev),
                first=rank if spmd else 0, stride=size if spmd else 1,
            )
        shards = []
        for i, ((lo, hi), dev) in enumerate(zip(self._A1_bounds, self._shard_devices)):
            shape = torch.Size((hi - lo, n2))
            if not owned(i) or self._sym is not None:
                shards.append(_RemoteShard(dev, shape, A1.dtype))  # shape bookkeeping only
            elif use_full_kernel:
                shards.append(_KernelShard(_kernel_kind, A1[lo:hi], A2, kernel_config, dev))
            else:
                shards.append(_UnavailableShard(dev, shape, A1.dtype))
        super().__init__(
            shape=torch.Size((n1, n2)), A=shards, distribution_mode="row", is_new=True,
            group=group, spmd=spmd,
        )
        self.kernel_ops = shards
        self._use_full_kernel = use_full_kernel

    @staticmethod
    def _my_device(devs, A1):
        if len(devs) == 1:
            return devs[0]
        if torch.cuda.is_available():
            cur = torch.device("cuda", torch.cuda.current_device())
            if cur in devs:
                return cur
        if A1.device in devs:
            return A1.device
        raise ValueError("in SPMD mode `devices` must contain this rank's device")

    A1 = property(lambda self: self._A1)
    A2 = property(lambda self: self._A2)
    kernel_config = property(lambda self: self._kernel_config)

    @staticmethod
    def _check_inputs(A1, A2, kernel_config, devices):
        _check_data(A1, A2, kernel_config, same_device=False)
        _is_set(devices, "devices")
        if len(devices) == 0:
            raise ValueError("devices must be a non-empty set.")
        if not all(isinstance(d, torch.device) for d in devices):
            raise ValueError("All elements in devices must be torch.device instances.")

    # -- sharded fast path used by PCG / Nystrom (SPMD: state stays row-partitioned) -------
    @property
    def is_row_sharded(self) -> bool:
        return self._spmd

    @property
    def row_range(self) -> tuple[int, int]:
        """[lo, hi) of the rows of A1 owned by this rank."""
        ids = self._local_ids()
        return self._A1_bounds[ids[0]] if ids else (self.shape[0], self.shape[0])

    @property
    def local_device(self) -> torch.device:
        ids = self._local_ids()
        return self._shard_devices[ids[0] if ids else 0]

    @property
    def storage_mode(self) -> str:
        if self._sym is not None:
            return "sympack" if self._sym.store_dtype == self.dtype else "sympack(float32 storage)"
        ids = self._local_ids()
        shard = self._A[ids[0]] if ids else None
        if isinstance(shard, _KernelShard):
            return "materialize" if shard._block.materialised else "matrix_free"
        return "oracle_only"

    def _sym_full_matmat(self, X_full: torch.Tensor) -> torch.Tensor:
        """K @ X (all rows) from the packed strips: local partial, summed over ranks."""
        X_full = X_full.to(self._sym.X.device)
        part = self._sym.partial_matmat(X_full)
        if self._spmd:
            _dist.all_reduce_sum(part, self._group)
        return part

    def _matvec(self, w: torch.Tensor) -> torch.Tensor:
        if self._sym is None:
            return super()._matvec(w)
        cols = 1 if w.ndim == 1 else w.shape[1]
        if cols <= _SYM_MAX_COLS:
            return self._sym_full_matmat(w).to(w.device)
        lo, hi = self.row_range
        ws = w.to(self._sym.X.device)
        if self._sym.gemm_ready(ws):
            full = self._sym.partial_matmat_wide(ws)  # every row, this rank's strips
            if self._spmd:
                _dist.all_reduce_sum(full, self._group)
            return full.to(w.device)
        mine = self._sym.rows_matmat_wide(ws, lo, hi)
        if self._spmd:
            mine = _dist.all_gather_rows(mine, self.shape[0], self._group)
        return mine.to(w.device)

    def _rmatvec(self, w: torch.Tensor) -> torch.Tensor:
        return self._matvec(w) if self._sym is not None else super()._rmatvec(w)

    @property
    def T(self):
        return self if self._sym is not None else super().T

    def local_matmat(self, X_full: torch.Tensor, diag_shift: float = 0.0) -> torch.Tensor:
        """K[rows_i, :] @ X_full (+ diag_shift * X_full[rows_i]) -> local row block."""
        if self._sym is not None:
            lo, hi = self.row_range
            cols = 1 if X_full.ndim == 1 else X_full.shape[1]
            if cols <= _SYM_MAX_COLS:
                if not self._spmd:  # one shard: the kernel applies the shift itself
                    return self._sym.partial_matmat(X_full.to(self._sym.X.device), diag_shift=diag_shift)
                Y = self._sym_full_matmat(X_full)[lo:hi]
            elif self._sym.gemm_ready(X_full):
                Y = self._sym.partial_matmat_wide(X_full.to(self._sym.X.device))
                if self._spmd:  # the ranks' partials add up to the product; this rank keeps its rows
                    _dist.all_reduce_sum(Y, self._group)
                Y = Y[lo:hi]
            else:
                Y = self._sym.rows_matmat_wide(X_full.to(self._sym.X.device), lo, hi)
            return Y + diag_shift * X_full[lo:hi].to(Y.device) if diag_shift != 0.0 else Y
        ids = self._local_ids()
        if not ids:
            k = () if X_full.ndim == 1 else (X_full.shape[1],)
            return torch.zeros((0, *k), dtype=X_full.dtype, device=X_full.device)
        shard = self._A[ids[0]]
        lo, _hi = self._A1_bounds[ids[0]]
        if isinstance(shard, _KernelShard):
            return shard._block.matmat(X_full.to(shard.device), diag_shift=diag_shift, row0=lo)
        return shard @ X_full  # raises for use_full_kernel=False

    def _rk_local_columns(self, J: torch.Tensor) -> torch.Tensor:
        """K[rows_i, J] for this rank's rows (one rank per GPU) or every row (one process): dense, written
        directly from the closed form - the column-sampling sketch's image needs no product and no collective."""
        be = get_backend()
        if self._spmd:
            lo, hi = self.row_range
            dev = self.local_device
        else:
            lo, hi = 0, self.shape[0]
            dev = self._shard_devices[0]
        cfg = self._kernel_config_devices[dev]
        A1r = self._A1[lo:hi].to(dev)
        A2d = A1r if (lo, hi) == (0, self.shape[0]) and _same_points(self._A1, self._A2) else self._A2.to(dev)
        return be.kernel_materialize(self._kind, A1r, A2d, cfg.lengthscale, scale=self._scaling, idx2=J.to(dev))

    def _rk_columns(self, J: torch.Tensor) -> torch.Tensor:
        Y = self._rk_local_columns(J)
        return _dist.all_gather_rows(Y, self.shape[0], self._group) if self._spmd else Y

    def _rk_shifted_matmat(self, X, shift):
        if self._spmd or len(self._A) != 1:
            return None  # generic path (A @ X, then + shift * X)
        return self.local_matmat(X, diag_shift=shift).to(X.device)

    def _rk_spmd_matmat_gram(self, X_full: torch.Tensor, shift: float):
        """SPMD + strip-packed storage: ((K + shift I) X)[rows_i] and X^T (K + shift I) X from ONE
        all-reduce.  Every rank holds a full-length partial of K X, so its share of the Gram
        block is X^T partial_i; it rides behind the partial in the same (float64) buffer.  The
        shift term is added by rank 0 only, so the sum contains it exactly once.  Returns None
        when this layout does not apply (the caller then uses product + separate Gram)."""
        if self._sym is None or not self._spmd or X_full.ndim != 2 or X_full.shape[1] > _SYM_MAX_COLS:
            return None
        be = get_backend()
        n, k = X_full.shape
        lo, hi = self.row_range
        X_full = X_full.to(self._sym.X.device)
        part = self._sym.partial_matmat(X_full, diag_shift=shift if self._rank == 0 else 0.0)
        gram = be.rmatmat(X_full, part, out_f64=True)
        if os.environ.get("RK_SPMD_REDUCE_SCATTER", "1") != "0":
            # every rank keeps only its rows of the sum: a reduce-scatter of the partials in their own
            # dtype, the k x k Gram partial riding in every slot (half the wire bytes per element of an
            # all-reduce; a quarter for float32 partials, which used to travel in a float64 buffer)
            Y, PAP = _dist.reduce_scatter_rows(part, n, tail=gram, group=self._group)
            return Y, PAP
        both = torch.cat([part.reshape(-1).double(), gram.reshape(-1)])
        _dist.all_reduce_sum(both, self._group)
        Y = both[: n * k].view(n, k)[lo:hi].to(X_full.dtype)
        return Y, both[n * k :].view(k, k).clone()

    @property
    def row_range_all(self):
        return self._A1_bounds

    # -- oracles ---------------------------------------------------------------------------
    def _gram_center_on(self, dev):
        """Centre valid for every row of A1 and A2, resident on ``dev`` (cached per device)."""
        cache = self.__dict__.setdefault("_gc_dev", {})
        if dev not in cache:
            cache[dev] = _gram_center(self._kind, self._A1, self._A2,
                                      self._kernel_config_devices[dev].lengthscale, device=dev)
        return cache[dev]

    def row_oracle(self, blk: torch.Tensor) -> _DistributedLinOp:
        """K(A1[blk], A2): column-distributed - shard i multiplies K(A1[blk], A2[cols_i]) with
        x[cols_i]; partial results are summed (base.py:408-447, utils.py:10-29)."""
        be = get_backend()
        ops = []
        for i, ((lo, hi), dev) in enumerate(zip(self._A2_bounds, self._shard_devices)):
            shape = torch.Size((blk.shape[0], hi - lo))
            A2c = self.A2_chunks[i]
            if A2c is None:
                ops.append(_RemoteShard(dev, shape, self.dtype))
                continue
            cfg = self._kernel_config_devices[dev]
            A1b = self._A1[blk.to(self._A1.device)].to(dev)  # gathered once per oracle
            gc = self._gram_center_on(dev)

            # (every row oracle of this operator multiplies by the same A2 chunk: its converted image is
            # kept between calls - the chunk, the lengthscale and the centre live as long as the operator)
            cache = self._row_x2_caches.setdefault(i, {})

            def matvec(x, A1b=A1b, A2c=A2c, cfg=cfg, gc=gc, cache=cache):
                return be.kernel_matmat(self._kind, A1b, A2c, cfg.lengthscale, x, scale=self._scaling,
                                        gram_center=gc, x2_cache=cache)

            ops.append(LinOp(device=dev, shape=shape, matvec=matvec, matmat=matvec, dtype=self.dtype))
        return _DistributedLinOp(
            shape=torch.Size((blk.shape[0], self._A2.shape[0])), A=ops,
            distribution_mode="column", is_new=False, group=self._group, spmd=self._spmd,
        )

    def blk_oracle(self, blk: torch.Tensor) -> _DistributedLinOp:
        """K(A1[blk], A2[blk]): row-distributed over chunks of blk (base.py:449-505)."""
        be = get_backend()
        es = torch.empty((), dtype=self.dtype).element_size()
        if (self._spmd and 0 < blk.shape[0] ** 2 * es <= _BLK_MATERIALIZE_BYTES
                and os.environ.get("RK_SAP_REPLICATE_BLOCK", "1") != "0"):
            # One rank per GPU and a block that fits: every rank keeps the WHOLE |blk| x |blk| block
            # (a SAP block is ~100 MB: 90 us to write).  The block preconditioner and the step size of
            # a SAP step are then rank-local - no collective inside the Nystrom assembly or the power
            # iteration, and the device-resident power iteration applies - and identical on every rank
            # (same block, same broadcast sketch seed); only the row-oracle product stays distributed.
            dev = self._shard_devices[self._rank]
            cfg = self._kernel_config_devices[dev]
            Ab = self._A1[blk.to(self._A1.device)].to(dev)
            A2b = Ab if _same_points(self._A1, self._A2) else self._A2[blk.to(self._A2.device)].to(dev)
            Kb = be.kernel_materialize(self._kind, Ab, A2b, cfg.lengthscale, scale=self._scaling,
                                       gram_center=self._gram_center_on(dev))

            def full_matvec(x, Kb=Kb):
                return be.matmat(Kb, x)

            op = LinOp(device=dev, shape=torch.Size((blk.shape[0], blk.shape[0])), matvec=full_matvec,
                       matmat=full_matvec, dtype=self.dtype)
            op._rk_dense = Kb
            return op
        parts = len(self._shard_devices)
        bounds = _dist.row_chunks(blk.shape[0], parts)
        ops = []
        dense = [None] * len(bounds)
        for i, (lo, hi) in enumerate(bounds):
            dev = self._shard_devices[i]
            shape = torch.Size((hi - lo, blk.shape[0]))
            if self._spmd and i != self._rank:
                ops.append(_RemoteShard(dev, shape, self.dtype))
                continue
            cfg = self._kernel_config_devices[dev]
            A1c = self._A1[blk[lo:hi].to(self._A1.device)].to(dev)
            A2b = self._A2[blk.to(self._A2.device)].to(dev)
            if 0 < (hi - lo) * blk.shape[0] * A1c.element_size() <= _BLK_MATERIALIZE_BYTES:
                # a block this small is written out once (|blk|^2 pair evaluations) and every
                # later product streams it: one SAP step multiplies with K_bb ~17 times (7 sketch
                # passes of 16 columns, <= 10 power iterations) - measured 2 ms of a 17 ms step
                Kb = be.kernel_materialize(self._kind, A1c, A2b, cfg.lengthscale, scale=self._scaling,
                                           gram_center=self._gram_center_on(dev))
                dense[i] = Kb

                def matvec(x, Kb=Kb):
                    return be.matmat(Kb, x)
            else:
                gc = self._gram_center_on(dev)

                def matvec(x, A1c=A1c, A2b=A2b, cfg=cfg, gc=gc):
                    return be.kernel_matmat(self._kind, A1c, A2b, cfg.lengthscale, x, scale=self._scaling,
                                            gram_center=gc)

            ops.append(LinOp(device=dev, shape=shape, matvec=matvec, matmat=matvec, dtype=self.dtype))
        op = _DistributedLinOp(
            shape=torch.Size((blk.shape[0], blk.shape[0])), A=ops, distribution_mode="row",
            is_new=False, group=self._group, spmd=self._spmd,
        )
        if len(bounds) == 1 and dense[0] is not None:
            op._rk_dense = dense[0]  # the whole block as one resident tensor (fused SAP step size)
        return op

    def shutdown(self):
        """Release the materialised shards (base.py:507-520)."""
        for op in self.kernel_ops:
            if hasattr(op, "_clear_cache"):
                op._clear_cache()
        if self._sym is not None:
            self._sym.release()
        super().shutdown()
