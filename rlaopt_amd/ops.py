"""Tensor-level entry points of the HIP hot path (thin wrappers over the C ABI).

PyTorch is used for device storage and streams only; every n-length or n x n computation
goes through ``librk_hip.so``.  All functions require CUDA(=HIP) tensors and raise
otherwise - there is no CPU fallback in the product.

``get_backend()`` returns the object the host-side classes call.  The product default is
:class:`HipBackend`.  Tests that exercise the *host logic* (sharding, collectives, loop
cadence) on CPU install an explicit stand-in with ``use_backend(...)``; nothing in this
package ever does so by itself.
"""

from __future__ import annotations

import contextlib
import ctypes
import os

import torch

from . import _lib

_DT = {torch.float32: _lib.RK_F32, torch.float64: _lib.RK_F64}
MAX_SKINNY = 16  # widest right-hand side handled by the streaming kernels


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _req(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(
                "rlaopt_amd: HIP kernels need tensors on a cuda (ROCm) device; "
                f"got {type(t).__name__} on {getattr(t, 'device', '?')}. There is no CPU fallback."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError(f"dtype must be float32 or float64, got {t.dtype}") from None


def _rowmajor(t):
    """2-D tensor with unit stride along columns (copy if needed) -> (tensor, ld)."""
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, max(ld, t.shape[1])


def _as2d(v):
    return (v[:, None], True) if v.ndim == 1 else (v, False)


def f16_grid(x: torch.Tensor) -> torch.Tensor:
    """x rounded to the float16 grid (kept in x's dtype) and tagged: products that split their operands into
    float16 pieces then need ONE piece of it (``symm_gemm(rhs_f16=True)``).  The tag is a Python attribute of this
    tensor object: slices and copies lose it (and then simply take the general path)."""
    y = x.to(torch.float16).to(x.dtype)
    y._rk_f16_grid = True
    return y


def on_f16_grid(x) -> bool:
    return bool(getattr(x, "_rk_f16_grid", False))


class HipBackend:
    """The product compute backend: every method is one or two kernel launches."""

    name = "hip"

    def __init__(self):
        self._lib = _lib.load()
        self._ws = {}
        self._timing = None  # {op: [(start_event, end_event, algorithmic_bytes), ...]}

    # -- optional per-launch timing (HIP events on the launch stream; used by bench.py) ------
    def enable_timing(self, on=True):
        self._timing = {} if on else None

    def _timed(self, op, device, nbytes):
        """Context manager bracketing one launch with events on the stream it is queued on."""
        be = self

        class _T:
            def __enter__(self_inner):
                self_inner.a = None
                if be._timing is None or torch.cuda.is_current_stream_capturing():
                    return  # (events recorded inside a graph capture cannot be timed)
                self_inner.a = torch.cuda.Event(enable_timing=True)
                self_inner.b = torch.cuda.Event(enable_timing=True)
                self_inner.a.record(torch.cuda.current_stream(device))

            def __exit__(self_inner, *exc):
                if be._timing is None or self_inner.a is None:
                    return
                self_inner.b.record(torch.cuda.current_stream(device))
                be._timing.setdefault(op, []).append((self_inner.a, self_inner.b, nbytes))

        return _T()

    def timing_summary(self):
        """{op: {"launches", "avg_ms", "bytes_per_launch"}} - synchronises the device."""
        torch.cuda.synchronize()
        out = {}
        for op, evs in (self._timing or {}).items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[op] = {
                "launches": len(ms),
                "avg_ms": sum(ms) / len(ms),
                "bytes_per_launch": sum(nb for _, _, nb in evs) / len(evs),
            }
        return out

    # -- scratch -----------------------------------------------------------------------
    def _workspace(self, device, nbytes):
        if nbytes == 0:
            return None, 0
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._ws[key] = buf
        return buf, buf.numel()

    # -- kernel matrices ---------------------------------------------------------------
    def _kernel_common(self, kind, X1, X2, lengthscale, idx1, idx2):
        dev = _req(X1, X2, idx1, idx2)
        if X1.dtype != X2.dtype:
            raise ValueError("X1 and X2 must have the same dtype")
        if X1.shape[1] != X2.shape[1]:
            raise ValueError("X1 and X2 must have the same number of features")
        X1, ld1 = _rowmajor(X1)
        X2, ld2 = _rowmajor(X2)
        ell_vec = None
        ell = 1.0
        if isinstance(lengthscale, torch.Tensor):
            ell_vec = lengthscale.to(device=dev, dtype=X1.dtype).contiguous()
            if ell_vec.numel() != X1.shape[1]:
                raise ValueError("lengthscale tensor must have one entry per feature")
        else:
            ell = float(lengthscale)
        if idx1 is not None:
            idx1 = idx1.to(device=dev, dtype=torch.int64).contiguous()
        if idx2 is not None:
            idx2 = idx2.to(device=dev, dtype=torch.int64).contiguous()
        n1 = X1.shape[0] if idx1 is None else idx1.numel()
        n2 = X2.shape[0] if idx2 is None else idx2.numel()
        return dev, _lib.KIND_IDS[kind], X1, ld1, X2, ld2, ell, ell_vec, idx1, idx2, n1, n2

    def kernel_materialize(self, kind, X1, X2, lengthscale, scale=1.0, idx1=None, idx2=None, out=None, gram_center=None):
        """Dense row-major block scale * K(X1[idx1], X2[idx2]).  ``gram_center``: the tensor :meth:`gram_center` returned for
        point sets that contain these (float32, d <= 64, rbf / matern32 / matern52, within the norm budget of stored
        entries): the block is evaluated on the matrix cores (``rk_kernel_materialize_mc``)."""
        dev, kid, X1, ld1, X2, ld2, ell, ell_vec, idx1, idx2, n1, n2 = self._kernel_common(
            kind, X1, X2, lengthscale, idx1, idx2
        )
        if out is None:
            out = torch.empty((n1, n2), dtype=X1.dtype, device=dev)
        if (gram_center is not None and X1.dtype == torch.float32 and X1.shape[1] <= 64 and kind in ("rbf", "matern32", "matern52")
                and n1 > 0 and n2 > 0 and out.dtype == torch.float32 and out.stride(1) == 1 and float(scale) != 0.0
                and os.environ.get("RK_SYMPACK_MC", "1") != "0"
                and getattr(gram_center, "_rk_norm2", float("inf")) <= float(os.environ.get("RK_SYMPACK_MC_NORM2", "8"))):
            _req(gram_center)
            gc = gram_center.contiguous()
            nbytes = self._lib.rk_kernel_materialize_mc_ws_bytes(n1, n2)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = self._lib.rk_kernel_materialize_mc(
                kid, _ptr(X1), n1, ld1, _ptr(idx1), _ptr(X2), n2, ld2, _ptr(idx2), X1.shape[1], ell, _ptr(ell_vec), _ptr(gc),
                float(scale), _ptr(out), out.stride(0), _ptr(ws), nbytes, _stream(dev),
            )
            _lib.check(rc, "rk_kernel_materialize_mc")
            return out
        rc = self._lib.rk_kernel_materialize(
            _dt(X1), kid, _ptr(X1), n1, ld1, _ptr(idx1), _ptr(X2), n2, ld2, _ptr(idx2),
            X1.shape[1], ell, _ptr(ell_vec), float(scale), _ptr(out), out.stride(0), _stream(dev),
        )
        _lib.check(rc, "rk_kernel_materialize")
        return out

    def gram_center(self, X1, X2, lengthscale):
        """Centre c (length-d tensor) for the matrix-core formulation of the matrix-free float32
        product, or None when the data do not qualify: float32 only, d <= 128, and
        max |(x - c) / ell|^2 <= RK_GRAM_MAX_NORM2 over the rows of X1 and X2 (include/rk.h).
        One small device reduction and ONE host read-back: operators call it once per point set
        (gathered subsets inherit the bound) and hand the result to ``kernel_matmat``."""
        dev = _req(X1, X2)
        if X1.dtype != torch.float32 or X2.dtype != torch.float32 or X1.shape[1] > 128:
            return None
        ell_vec, ell = None, 1.0
        if isinstance(lengthscale, torch.Tensor):
            ell_vec = lengthscale.to(device=dev, dtype=X1.dtype).contiguous()
        else:
            ell = float(lengthscale)
        X2r, ld2 = _rowmajor(X2)
        d = X2r.shape[1]
        center = torch.empty(d, dtype=X1.dtype, device=dev)
        bound = torch.zeros(1, dtype=X1.dtype, device=dev)
        rc = self._lib.rk_kernel_center(_dt(X2r), _ptr(X2r), X2r.shape[0], ld2, d, _ptr(center), _stream(dev))
        _lib.check(rc, "rk_kernel_center")
        sets = [(X2r, ld2)]
        if not (X1 is X2 or (X1.data_ptr() == X2.data_ptr() and X1.shape == X2.shape)):
            sets.append(_rowmajor(X1))
        for X, ld in sets:
            rc = self._lib.rk_kernel_gram_bound(
                _dt(X), _ptr(X), X.shape[0], ld, d, ell, _ptr(ell_vec), _ptr(center), _ptr(bound), _stream(dev)
            )
            _lib.check(rc, "rk_kernel_gram_bound")
        b = float(bound)
        if b > _lib.GRAM_MAX_NORM2:
            return None
        center._rk_norm2 = b  # (max |(x - c) / ell|^2 found: consumers with a tighter budget read it)
        return center

    def kernel_matmat(self, kind, X1, X2, lengthscale, V, scale=1.0, idx1=None, idx2=None,
                      gram_center="auto", x2_cache=None):
        """scale * K(X1[idx1], X2[idx2]) @ V, matrix-free; wide V is processed in
        16-column groups (each group recomputes the kernel entries).  ``gram_center``: the
        tensor returned by :meth:`gram_center` for these point sets (matrix-core formulation),
        None (direct-difference kernels), or "auto" = compute it here (one host read-back per
        call: operators precompute it instead).  ``x2_cache``: a dict an operator keeps for ONE
        (X2, idx2, lengthscale, gram_center) combination it multiplies by repeatedly - the converted
        image of that point set (``rk_kernel_matmat_cached``) is kept in it between calls."""
        dev, kid, X1, ld1, X2, ld2, ell, ell_vec, idx1, idx2, n1, n2 = self._kernel_common(
            kind, X1, X2, lengthscale, idx1, idx2
        )
        if isinstance(gram_center, str):
            if gram_center != "auto":
                raise ValueError("gram_center must be a tensor, None or 'auto'")
            gram_center = self.gram_center(X1, X2, lengthscale) if kind not in ("laplace", "matern12") else None
        elif gram_center is not None:
            _req(gram_center)
            if gram_center.dtype != X1.dtype or gram_center.numel() != X1.shape[1]:
                raise ValueError("gram_center must have one entry per feature and the dtype of the data")
            gram_center = gram_center.contiguous()
        _req(V)
        V2, squeeze = _as2d(V)
        if V2.shape[0] != n2:
            raise ValueError(f"V has {V2.shape[0]} rows, operator has {n2} columns")
        if V2.dtype != X1.dtype:
            raise ValueError("V must have the dtype of the data")
        V2, ldv = _rowmajor(V2)
        k = V2.shape[1]
        out = torch.empty((n1, k), dtype=X1.dtype, device=dev)
        same = (X1.data_ptr() == X2.data_ptr() and ld1 == ld2 and n1 == n2 and gram_center is not None
                and (idx1 is idx2 or (idx1 is not None and idx2 is not None and idx1.data_ptr() == idx2.data_ptr())))
        if same and idx1 is not None:
            idx2 = idx1  # (one pointer: the C side recognises the symmetric case by pointer equality)
        for c0 in range(0, k, MAX_SKINNY):
            kc = min(MAX_SKINNY, k - c0)
            nbytes = self._lib.rk_kernel_matmat_ws_bytes(_dt(X1), n1, n2, kc)
            if same:  # symmetric evaluation (half the tiles) needs its tables
                nbytes = max(nbytes, self._lib.rk_kernel_matmat_sym_ws_bytes(_dt(X1), n1, kc))
            ws, wsn = self._workspace(dev, nbytes)
            es = V2.element_size()
            img = self._x2_image(x2_cache, dev, kid, X1, X2, ld2, idx2, n1, n2, ell, ell_vec, gram_center, kc, same,
                                 lengthscale=lengthscale)
            if img is not None:
                buf, ready, slot = img
                valid = ctypes.c_int(0)
                rc = self._lib.rk_kernel_matmat_cached(
                    _dt(X1), kid, _ptr(X1), n1, ld1, _ptr(idx1), _ptr(X2), n2, ld2, _ptr(idx2),
                    X1.shape[1], ell, _ptr(ell_vec), float(scale), V2.data_ptr() + c0 * es, kc, ldv,
                    out.data_ptr() + c0 * es, k, 0, _ptr(gram_center), _ptr(ws), wsn, _ptr(buf), buf.numel(),
                    int(ready), ctypes.byref(valid), _stream(dev),
                )
                _lib.check(rc, "rk_kernel_matmat_cached")
                slot["ready"] = bool(valid.value)
                continue
            rc = self._lib.rk_kernel_matmat(
                _dt(X1), kid, _ptr(X1), n1, ld1, _ptr(idx1), _ptr(X2), n2, ld2, _ptr(idx2),
                X1.shape[1], ell, _ptr(ell_vec), float(scale), V2.data_ptr() + c0 * es, kc, ldv,
                out.data_ptr() + c0 * es, k, 0, _ptr(gram_center), _ptr(ws), wsn, _stream(dev),
            )
            _lib.check(rc, "rk_kernel_matmat")
        return out[:, 0] if squeeze else out

    def _x2_image(self, cache, dev, kid, X1, X2, ld2, idx2, n1, n2, ell, ell_vec, gram_center, kc, same,
                  lengthscale=None):
        """(buffer, ready, slot) for ``rk_kernel_matmat_cached`` or None.  One image per stream (a call on
        another stream converts into its own buffer: nothing orders two streams' use of one).  An image
        is valid for exactly the operands that wrote it: the key carries their addresses AND their
        in-place modification counters, and the slot holds references to them, so that neither an
        in-place update nor an address the allocator hands out again can make a stale image look
        current."""
        if cache is None or gram_center is None or X1.dtype != torch.float32:
            return None
        nbytes = self._lib.rk_kernel_x2_image_bytes(_dt(X1), kid, n1, n2, X1.shape[1], kc, int(bool(same)))
        if nbytes == 0:
            return None
        ell_src = lengthscale if isinstance(lengthscale, torch.Tensor) else None
        held = (X2, idx2, ell_src, gram_center)
        key = (n2, ld2, X1.shape[1], float(ell), kid, dev.index) + tuple(
            (0, 0) if t is None else (t.data_ptr(), t._version) for t in held)
        slot = cache.setdefault("streams", {}).setdefault(_stream(dev), {})
        buf = slot.get("buf")
        if buf is None or buf.numel() < nbytes or buf.device != dev:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            slot.update(buf=buf, key=None, ready=False)
        if slot.get("key") != key:
            slot.update(key=key, ready=False, held=held)
        return buf, bool(slot.get("ready")), slot

    # -- symmetric packed kernel matrices ------------------------------------------------------
    def sympack_elems(self, n, first=0, stride=1, store_dtype=torch.float32):
        return int(self._lib.rk_sympack_elems(_DT[store_dtype], n, first, stride))

    def sympack_materialize(self, kind, X, lengthscale, scale=1.0, first=0, stride=1, store_dtype=None, out=None,
                            gram_center="auto"):
        """Strip-packed upper triangle of scale * K(X, X) (strips first, first+stride, ...).
        ``store_dtype=torch.float32`` with float64 X keeps the entries in float32.  ``out``: a 1-D buffer of the
        storage dtype with at least ``sympack_elems`` entries to write into (its head is returned).
        float32, d <= 64, rbf / matern32 / matern52 and a centre within the norm bound (``gram_center``: the tensor
        :meth:`gram_center` returned for X, None = do not, "auto" = compute it here): the entries are evaluated on
        the matrix cores (``rk_kernel_materialize_sympack_mc``) when max |(x - c) / ell|^2 <= RK_SYMPACK_MC_NORM2 (8);
        RK_SYMPACK_MC=0 keeps the vector-unit kernel."""
        dev, kid, X, ld, _X2, _ld2, ell, ell_vec, _i1, _i2, n, _n2 = self._kernel_common(
            kind, X, X, lengthscale, None, None
        )
        store_dtype = store_dtype or X.dtype
        elems = self._lib.rk_sympack_elems(_DT[store_dtype], n, first, stride)
        if out is not None:
            _req(out)
            if out.dtype != store_dtype or out.ndim != 1 or out.numel() < elems or not out.is_contiguous():
                raise ValueError("sympack_materialize: out must be a contiguous 1-D buffer of the storage dtype with room "
                                 "for the packed strips")
            Kp = out[: max(elems, 1)]
        else:
            Kp = torch.empty(max(elems, 1), dtype=store_dtype, device=dev)
        if (X.dtype == torch.float32 and store_dtype == torch.float32 and X.shape[1] <= 64 and kind in ("rbf", "matern32", "matern52")
                and gram_center is not None and float(scale) != 0.0 and os.environ.get("RK_SYMPACK_MC", "1") != "0"):
            gc = self.gram_center(X, X, lengthscale) if isinstance(gram_center, str) else gram_center
            # entry error of this evaluation: 2^-22 |x'||y'| in the dot product, i.e. up to ~7e-7 max|x'|^2 relative to the
            # kernel's scale: taken for STORED entries only while that stays near float32's own rounding
            lim = float(os.environ.get("RK_SYMPACK_MC_NORM2", "8"))
            if gc is not None and getattr(gc, "_rk_norm2", float("inf")) <= lim:
                _req(gc)
                gc = gc.contiguous()
                nbytes = self._lib.rk_kernel_materialize_sympack_mc_ws_bytes(n)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                rc = self._lib.rk_kernel_materialize_sympack_mc(
                    kid, _ptr(X), n, ld, X.shape[1], ell, _ptr(ell_vec), _ptr(gc), float(scale), first, stride, _ptr(Kp),
                    _ptr(ws), nbytes, _stream(dev),
                )
                _lib.check(rc, "rk_kernel_materialize_sympack_mc")
                return Kp
        rc = self._lib.rk_kernel_materialize_sympack(
            _dt(X), _DT[store_dtype], kid, _ptr(X), n, ld, X.shape[1], ell, _ptr(ell_vec),
            float(scale), first, stride, _ptr(Kp), _stream(dev),
        )
        _lib.check(rc, "rk_kernel_materialize_sympack")
        return Kp

    def symm_matmat(self, Kp, n, V, alpha=1.0, diag_shift=0.0, first=0, stride=1, tag=None, kmax=None):
        """alpha * (this rank's partial of) K V + diag_shift * V from the packed triangle; the
        arithmetic type is V's dtype, the storage type Kp's (float32 storage under float64 V).
        ``kmax``: a bound on the magnitude of the stored entries (the operator's const_scaling) - with it,
        float32 products of RK_SYMM_F16X2_MIN (3) .. 16 columns run on the 16-bit matrix cores
        (``rk_symm_matmat_f16x2``, block form)."""
        dev = _req(Kp, V)
        V2, squeeze = _as2d(V)
        if V2.shape[0] != n:
            raise ValueError("V must have n rows")
        if not (V2.dtype == Kp.dtype or (V2.dtype == torch.float64 and Kp.dtype == torch.float32)):
            raise ValueError("V must have the dtype of the packed matrix (or float64 over float32 storage)")
        V2, ldv = _rowmajor(V2)
        k = V2.shape[1]
        out = torch.empty((n, k), dtype=V2.dtype, device=dev)
        es = V2.element_size()
        # one or two columns run the block decomposition on the vector units (rk_symm_matmat -> symm_blk_reg_kernel: exact
        # float32 arithmetic, no split prologue; as fast as the matrix-core form at 80 GB - 11.6 ms - and faster on
        # small shards: 1.57 against 1.63 ms at 10 GB); from three columns on the 16-bit matrix cores take over
        if (kmax is not None and kmax > 0 and V2.dtype == torch.float32 and Kp.dtype == torch.float32
                and k >= int(os.environ.get("RK_SYMM_F16X2_MIN", "3")) and os.environ.get("RK_SYMM_F16X2", "1") != "0"):
            for c0 in range(0, k, 16):
                kc = min(16, k - c0)
                ws, wsn = self._workspace(dev, self._lib.rk_symm_matmat_f16x2_ws_bytes(n, kc, first, stride))
                algo = Kp.numel() * 4 + 2 * n * kc * 4
                with self._timed(tag or "rk_symm_matmat", dev, algo):
                    rc = self._lib.rk_symm_matmat_f16x2(
                        _ptr(Kp), n, first, stride, float(kmax), V2.data_ptr() + c0 * es, kc, ldv, float(alpha),
                        float(diag_shift), out.data_ptr() + c0 * es, k, _ptr(ws), wsn, _stream(dev))
                _lib.check(rc, "rk_symm_matmat_f16x2")
            return out[:, 0] if squeeze else out
        # 5 .. 16 right-hand sides go through the matrix-core strip kernel in ONE pass over the packed
        # triangle (storage and arithmetic type equal); otherwise groups of 4 (register kernel)
        wide_from = 4 if V2.dtype == torch.float64 else 5  # (the float64 register kernel is slow at k = 4)
        group = 16 if (k >= wide_from and V2.dtype == Kp.dtype and os.environ.get("RK_SYMM_WIDE", "1") != "0") else 4
        for c0 in range(0, k, group):
            kc = min(group, k - c0)
            ws, wsn = self._workspace(
                dev, self._lib.rk_symm_matmat_ws_bytes(_dt(V2), _dt(Kp), n, kc, first, stride)
            )
            algo = Kp.numel() * Kp.element_size() + 2 * n * kc * es  # packed triangle once + V + Y
            with self._timed(tag or "rk_symm_matmat", dev, algo):
                rc = self._lib.rk_symm_matmat(
                    _dt(V2), _dt(Kp), _ptr(Kp), n, first, stride, V2.data_ptr() + c0 * es, kc, ldv,
                    float(alpha), float(diag_shift), out.data_ptr() + c0 * es, k, _ptr(ws), wsn,
                    _stream(dev),
                )
            _lib.check(rc, "rk_symm_matmat")
        return out[:, 0] if squeeze else out

    def symm_gemm(self, Kp, n, V, kmax=1.0, alpha=1.0, first=0, stride=1, tag=None, rhs_f16=False):
        """alpha * (this rank's partial of) K V for MANY columns (the sketch product Y = K Omega), from
        the float32 packed triangle in place, on the 16-bit matrix cores (``rk_symm_gemm``).  ``kmax``
        bounds the magnitude of the stored entries (the operator's const_scaling).  ``rhs_f16``: V is on
        the float16 grid (``f16_grid``): one piece of V, two products per pair (``rk_symm_gemm_rhs16``)."""
        dev = _req(Kp, V)
        if Kp.dtype != torch.float32 or V.dtype != torch.float32 or V.ndim != 2 or V.shape[0] != n:
            raise ValueError("symm_gemm: float32 packed matrix and a float32 n x k block")
        V2, ldv = _rowmajor(V)
        k = V2.shape[1]
        out = torch.empty((n, k), dtype=torch.float32, device=dev)
        nbytes = self._lib.rk_symm_gemm_ws_bytes(n, k)
        # (its own buffer: the split image of V is hundreds of MB - not something to pin in the shared workspace)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with self._timed(tag or "rk_symm_gemm", dev, 2 * Kp.numel() * 4 + 2 * n * k * 4):
            fn = self._lib.rk_symm_gemm_rhs16 if rhs_f16 else self._lib.rk_symm_gemm
            rc = fn(_ptr(Kp), n, first, stride, float(kmax), _ptr(V2), k, ldv, float(alpha), _ptr(out), k, _ptr(ws),
                    nbytes, _stream(dev))
        _lib.check(rc, "rk_symm_gemm")
        return out

    # -- dense products ----------------------------------------------------------------
    def kernel_sketch_matmat(self, kind, X1, X2, lengthscale, V, scale=1.0, gram_center=None, alpha=1.0, tag=None,
                             rhs_f16=False):
        """alpha * scale * K(X1, X2) @ V for MANY columns with the kernel entries evaluated inside the product
        (``rk_kernel_sketch_matmat``: float32, d <= 64, rbf / matern32 / matern52, a ``gram_center`` from
        :meth:`gram_center` required) - the sketch product of an operator that stores nothing."""
        dev, kid, X1, ld1, X2, ld2, ell, ell_vec, _i1, _i2, n1, n2 = self._kernel_common(kind, X1, X2, lengthscale, None, None)
        if gram_center is None or X1.dtype != torch.float32 or V.dtype != torch.float32 or V.ndim != 2 or V.shape[0] != n2:
            raise ValueError("kernel_sketch_matmat: float32 data, a float32 n2 x k block and a gram_center")
        _req(V, gram_center)
        gram_center = gram_center.contiguous()
        V2, ldv = _rowmajor(V)
        k = V2.shape[1]
        out = torch.empty((n1, k), dtype=torch.float32, device=dev)
        for c0 in range(0, k, 1024):
            kc = min(1024, k - c0)
            nbytes = self._lib.rk_kernel_sketch_ws_bytes(n1, n2, kc)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)  # (its own buffer: hundreds of MB of split images)
            with self._timed(tag or "rk_kernel_sketch_matmat", dev, 0):
                rc = (self._lib.rk_kernel_sketch_matmat_rhs16 if rhs_f16 else self._lib.rk_kernel_sketch_matmat)(
                    kid, _ptr(X1), n1, ld1, _ptr(X2), n2, ld2, X1.shape[1], ell, _ptr(ell_vec), _ptr(gram_center),
                    float(scale), V2.data_ptr() + c0 * 4, kc, ldv, float(alpha), out.data_ptr() + c0 * 4, k, _ptr(ws),
                    nbytes, _stream(dev))
            _lib.check(rc, "rk_kernel_sketch_matmat")
        return out

    def matmat(self, A, V, alpha=1.0, beta=0.0, C=None, diag_shift=0.0, row0=0, tag=None):
        """alpha * A @ V + beta * C + diag_shift * V[row0 : row0 + m].  ``tag`` names the
        launch in the optional timing table."""
        dev = _req(A, V, C)
        V2, squeeze = _as2d(V)
        if A.shape[1] != V2.shape[0]:
            raise ValueError(f"shape mismatch: {tuple(A.shape)} @ {tuple(V2.shape)}")
        if A.dtype != V2.dtype:
            raise ValueError("A and V must have the same dtype")
        A, lda = _rowmajor(A)
        V2, ldv = _rowmajor(V2)
        m, n = A.shape
        k = V2.shape[1]
        C2 = None
        if C is not None:
            C2 = C[:, None] if C.ndim == 1 else C
            C2, ldc = _rowmajor(C2)
        out = torch.empty((m, k), dtype=A.dtype, device=dev)
        if k <= MAX_SKINNY:
            nbytes = self._lib.rk_matmat_skinny_ws_bytes(_dt(A), m, n, k)
            ws, wsn = self._workspace(dev, nbytes)
            algo = (m * n + n * k + m * k) * A.element_size()
            with self._timed(tag or "rk_matmat_skinny(untagged)", dev, algo):
                rc = self._lib.rk_matmat_skinny(
                    _dt(A), _ptr(A), m, n, lda, _ptr(V2), k, ldv, float(alpha), float(beta),
                    _ptr(C2), (ldc if C2 is not None else k), float(diag_shift), int(row0),
                    _ptr(out), k, _ptr(ws), wsn, _stream(dev),
                )
            _lib.check(rc, "rk_matmat_skinny")
        else:
            if beta != 0.0:
                out.copy_(C2)
            ws, wsn = self._workspace(dev, self._lib.rk_gemm_ws_bytes(_dt(A), m, k, n))
            rc = self._lib.rk_gemm(
                _dt(A), 0, 0, m, k, n, float(alpha), _ptr(A), lda, _ptr(V2), ldv, float(beta),
                _ptr(out), k, _ptr(ws), wsn, _stream(dev),
            )
            _lib.check(rc, "rk_gemm")
            if diag_shift != 0.0:
                out.add_(V2[row0 : row0 + m], alpha=float(diag_shift))
        return out[:, 0] if squeeze else out

    def rmatmat(self, A, V, alpha=1.0, beta=0.0, C=None, out_f64=False):
        """alpha * A.T @ V + beta * C  (A is m x n, V is m x k)."""
        dev = _req(A, V, C)
        V2, squeeze = _as2d(V)
        if A.shape[0] != V2.shape[0]:
            raise ValueError(f"shape mismatch: {tuple(A.shape)}.T @ {tuple(V2.shape)}")
        if A.dtype != V2.dtype:
            raise ValueError("A and V must have the same dtype")
        A, lda = _rowmajor(A)
        V2, ldv = _rowmajor(V2)
        m, n = A.shape
        k = V2.shape[1]
        C2 = None
        if C is not None:
            C2 = C[:, None] if C.ndim == 1 else C
            C2, ldc = _rowmajor(C2)
        odt = torch.float64 if out_f64 else A.dtype
        out = torch.empty((n, k), dtype=odt, device=dev)
        if out_f64 and n <= 4 and k <= 4 and beta == 0.0:
            # block inner product of two tall blocks: dedicated two-launch reduction
            ws, wsn = self._workspace(dev, self._lib.rk_gram_small_ws_bytes(m, n, k))
            rc = self._lib.rk_gram_small(
                _dt(A), _ptr(A), lda, _ptr(V2), ldv, m, n, k, _ptr(out), _ptr(ws), wsn, _stream(dev)
            )
            _lib.check(rc, "rk_gram_small")
            if alpha != 1.0:
                out.mul_(alpha)
        elif k <= MAX_SKINNY:
            nbytes = self._lib.rk_matmat_skinny_t_ws_bytes(_dt(A), m, n, k)
            ws, wsn = self._workspace(dev, nbytes)
            rc = self._lib.rk_matmat_skinny_t(
                _dt(A), _ptr(A), m, n, lda, _ptr(V2), k, ldv, float(alpha), float(beta),
                _ptr(C2), (ldc if C2 is not None else k), _ptr(out), k, int(out_f64),
                _ptr(ws), wsn, _stream(dev),
            )
            _lib.check(rc, "rk_matmat_skinny_t")
        else:
            if out_f64:
                if beta != 0.0:
                    raise ValueError("out_f64 with beta != 0 is not supported for wide V")
                out = self.gram_tn(A, V2)
                if alpha != 1.0:
                    out.mul_(alpha)
            else:
                if beta != 0.0:
                    out.copy_(C2)
                ws, wsn = self._workspace(dev, self._lib.rk_gemm_ws_bytes(_dt(A), n, k, m))
                rc = self._lib.rk_gemm(
                    _dt(A), 1, 0, n, k, m, float(alpha), _ptr(A), lda, _ptr(V2), ldv,
                    float(beta), _ptr(out), k, _ptr(ws), wsn, _stream(dev),
                )
                _lib.check(rc, "rk_gemm")
        return out[:, 0] if squeeze else out

    def gemm(self, A, B, transA=False, transB=False, alpha=1.0):
        """alpha * op(A) @ op(B) on the matrix cores."""
        dev = _req(A, B)
        if A.dtype != B.dtype:
            raise ValueError("A and B must have the same dtype")
        A, lda = _rowmajor(A)
        B, ldb = _rowmajor(B)
        m, kd = (A.shape[1], A.shape[0]) if transA else A.shape
        kb, n = (B.shape[1], B.shape[0]) if transB else B.shape
        if kd != kb:
            raise ValueError("inner dimensions differ")
        out = torch.empty((m, n), dtype=A.dtype, device=dev)
        ws, wsn = self._workspace(dev, self._lib.rk_gemm_ws_bytes(_dt(A), m, n, kd))
        rc = self._lib.rk_gemm(
            _dt(A), int(transA), int(transB), m, n, kd, float(alpha), _ptr(A), lda, _ptr(B), ldb,
            0.0, _ptr(out), n, _ptr(ws), wsn, _stream(dev),
        )
        _lib.check(rc, "rk_gemm")
        return out

    def gram_tn(self, A, B):
        """float64 A.T @ B for tall A (rows x m), B (rows x n), f64 accumulation."""
        dev = _req(A, B)
        if A.dtype != B.dtype or A.shape[0] != B.shape[0]:
            raise ValueError("A and B must share dtype and row count")
        A, lda = _rowmajor(A)
        B, ldb = _rowmajor(B)
        rows, m = A.shape
        n = B.shape[1]
        out = torch.empty((m, n), dtype=torch.float64, device=dev)
        ws, wsn = self._workspace(dev, self._lib.rk_gram_tn_ws_bytes(rows, m, n))
        rc = self._lib.rk_gram_tn(
            _dt(A), _ptr(A), lda, _ptr(B), ldb, rows, m, n, _ptr(out), n, _ptr(ws), wsn,
            _stream(dev),
        )
        _lib.check(rc, "rk_gram_tn")
        return out

    def spmd_pack(self, part, n, step, slots, tail=None):
        """(slots, step k + t) wire buffer of ``dist.reduce_scatter_rows``: slot r = [rows r step ..
        of the n x k partial, zero padded | tail] in the partial's dtype (``rk_spmd_pack``, one launch)."""
        dev = _req(part, tail)
        P2, ldp = _rowmajor(part)
        k = P2.shape[1]
        t = 0 if tail is None else tail.numel()
        if tail is not None and (tail.dtype != torch.float64 or not tail.is_contiguous()):
            tail = tail.double().contiguous()
        out = torch.empty((slots, step * k + t), dtype=P2.dtype, device=dev)
        rc = self._lib.rk_spmd_pack(_dt(P2), _ptr(P2), n, k, ldp, _ptr(tail), t, step, slots, _ptr(out), _stream(dev))
        _lib.check(rc, "rk_spmd_pack")
        return out

    def nystrom_inv_apply(self, U, M, rho, X):
        """(X - U (M (U^T X))) / rho in three launches (``rk_nystrom_inv_apply``); M: r x r or length-r float64."""
        dev = _req(U, M, X)
        X2, squeeze = _as2d(X)
        if X2.dtype != U.dtype or M.dtype != torch.float64 or X2.shape[0] != U.shape[0]:
            raise ValueError("nystrom_inv_apply: X of U's dtype and row count, M float64")
        U2, ldu = _rowmajor(U)
        X2, ldx = _rowmajor(X2)
        M = M.contiguous()
        n, r = U2.shape
        k = X2.shape[1]
        out = torch.empty((n, k), dtype=U.dtype, device=dev)
        ws, wsn = self._workspace(dev, self._lib.rk_nystrom_inv_apply_ws_bytes(_dt(U2), n, r, k))
        rc = self._lib.rk_nystrom_inv_apply(_dt(U2), _ptr(U2), n, r, ldu, _ptr(M), int(M.ndim == 1), float(rho),
                                            _ptr(X2), k, ldx, _ptr(out), k, _ptr(ws), wsn, _stream(dev))
        _lib.check(rc, "rk_nystrom_inv_apply")
        return out[:, 0] if squeeze else out

    # -- block PCG pieces --------------------------------------------------------------
    def pcg_update_wr(self, W, R, P, AP, alpha):
        dev = _req(W, R, P, AP, alpha)
        n, k = W.shape
        rc = self._lib.rk_pcg_update_wr(
            _dt(W), n, k, _ptr(W), W.stride(0), _ptr(R), R.stride(0), _ptr(P), P.stride(0),
            _ptr(AP), AP.stride(0), _ptr(alpha), _stream(dev),
        )
        _lib.check(rc, "rk_pcg_update_wr")

    def pcg_update_p(self, P, Z, beta):
        dev = _req(P, Z, beta)
        n, k = P.shape
        rc = self._lib.rk_pcg_update_p(
            _dt(P), n, k, _ptr(P), P.stride(0), _ptr(Z), Z.stride(0), _ptr(beta), _stream(dev)
        )
        _lib.check(rc, "rk_pcg_update_p")

    def axpby_cols(self, X, a, Y, b=None):
        """Y[:, c] = a[c] * X[:, c] + b[c] * Y[:, c] in place; a / b float64 device vectors or None."""
        dev = _req(X, Y, a, b)
        if X.shape != Y.shape or X.dtype != Y.dtype:
            raise ValueError("X and Y must match in shape and dtype")
        X2 = X[:, None] if X.ndim == 1 else X
        Y2 = Y[:, None] if Y.ndim == 1 else Y
        rc = self._lib.rk_axpby_cols(
            _dt(X2), X2.shape[0], X2.shape[1], _ptr(X2), X2.stride(0), _ptr(a), _ptr(Y2),
            Y2.stride(0), _ptr(b), _stream(dev),
        )
        _lib.check(rc, "rk_axpby_cols")
        return Y

    def lsqr_tail(self, G, V, W, Y, beta, rhobar, phibar):
        """The n-space half of an LSQR step in one launch (``rk_lsqr_tail``): V, W, Y (n x k, working
        dtype) and rhobar, phibar (float64 k-vectors) are updated in place; returns alpha (float64 k).
        None when the operands are not plain row-major device tensors of one dtype."""
        ok = (G.ndim == 2 and G.shape == V.shape == W.shape == Y.shape and G.dtype == V.dtype == W.dtype == Y.dtype
              and all(t.is_cuda and t.stride(1) == 1 for t in (G, V, W, Y))
              and all(t.dtype == torch.float64 and t.is_contiguous() and t.numel() == G.shape[1]
                      for t in (beta, rhobar, phibar)))
        if not ok:
            return None
        dev = _req(G, V, W, Y, beta, rhobar, phibar)
        n, k = G.shape
        alpha = torch.empty(k, dtype=torch.float64, device=dev)
        rc = self._lib.rk_lsqr_tail(
            _dt(G), n, k, _ptr(G), G.stride(0), _ptr(V), V.stride(0), _ptr(W), W.stride(0), _ptr(Y), Y.stride(0),
            _ptr(beta), _ptr(rhobar), _ptr(phibar), _ptr(alpha), _stream(dev),
        )
        _lib.check(rc, "rk_lsqr_tail")
        return alpha

    def normal_pass(self, A, Z, U, coef):
        """One pass over the tall A for a Golub-Kahan step: U <- A Z - U diag(coef) in place;
        returns (A^T U_new as float64 n x k, column sums of squares of U_new as float64 k), or
        None when the shape is outside what the fused kernel supports (the caller then does the
        two products).  ``coef``: float64 device vector of length k."""
        dev = _req(A, Z, U, coef)
        if A.ndim != 2 or Z.ndim != 2 or U.ndim != 2 or A.dtype != Z.dtype or A.dtype != U.dtype:
            return None
        m, n = A.shape
        k = Z.shape[1]
        Z, _ldz = _rowmajor(Z)  # (n x k: tiny; triangular solves hand back column-major results)
        if (Z.shape[0] != n or U.shape != (m, k) or A.stride(1) != 1
                or U.stride(1) != 1 or A.data_ptr() % 16 != 0 or coef.dtype != torch.float64
                or not self._lib.rk_normal_pass_supported(_dt(A), m, n, A.stride(0), k)):
            return None
        G = torch.empty((n, k), dtype=torch.float64, device=dev)
        ss = torch.empty(k, dtype=torch.float64, device=dev)
        ws, wsn = self._workspace(dev, self._lib.rk_normal_pass_ws_bytes(_dt(A), m, n, k))
        algo = (m * n + 2 * m * k + 2 * n * k) * A.element_size()
        with self._timed("rk_normal_pass", dev, algo):
            rc = self._lib.rk_normal_pass(
                _dt(A), _ptr(A), m, n, A.stride(0), _ptr(Z), k, Z.stride(0), _ptr(U), U.stride(0),
                _ptr(coef), _ptr(G), _ptr(ss), _ptr(ws), wsn, _stream(dev),
            )
        _lib.check(rc, "rk_normal_pass")
        return G, ss

    def power_iteration(self, K, reg, v0, U=None, M=None, rho=1.0, max_iters=10, rtol=1e-3):
        """lambda_max(Pinv (K + reg I)) by the reference's power iteration
        (spectral_estimators/spectral_norm.py:11-29), queued as one launch sequence without host
        synchronisation (``rk_power_iteration``).  Pinv = (I - U M U^T) / rho for Nystrom factors
        (M: r x r float64, or a length-r float64 diagonal), the identity when U is None.  Returns a
        float64 device tensor [eigenvalue, converged, iterations]; ``v0`` is overwritten."""
        dev = _req(K, v0, U, M)
        K, ldk = _rowmajor(K)
        m = K.shape[0]
        if K.shape[1] != m or v0.shape != (m,) or v0.dtype != K.dtype or not v0.is_contiguous():
            raise ValueError("power_iteration: K must be square and v0 a contiguous vector of its dtype")
        r, ldu, mdiag = 0, 1, 0
        if U is not None:
            U, ldu = _rowmajor(U)
            r = U.shape[1]
            if U.shape[0] != m or U.dtype != K.dtype or M is None or M.dtype != torch.float64:
                raise ValueError("power_iteration: U must be m x r in K's dtype and M float64")
            M = M.contiguous()
            mdiag = int(M.ndim == 1)
            if tuple(M.shape) not in ((r,), (r, r)):
                raise ValueError("power_iteration: M must be r x r or length r")
        state = torch.empty(3, dtype=torch.float64, device=dev)
        ws, wsn = self._workspace(dev, self._lib.rk_power_iteration_ws_bytes(_dt(K), m, r))
        rc = self._lib.rk_power_iteration(
            _dt(K), _ptr(K), m, ldk, float(reg), _ptr(U), r, ldu, _ptr(M), mdiag, float(rho), _ptr(v0),
            int(max_iters), float(rtol), _ptr(state), _ptr(ws), wsn, _stream(dev),
        )
        _lib.check(rc, "rk_power_iteration")
        return state

    def sap_update(self, W, V, Y, blk, D, eta, accel, beta=0.0, gamma=0.0, alpha=0.0):
        """In-place SAP iterate update (``rk_sap_update``); ``eta`` a float or a 0-dim device
        tensor of the iterates' dtype.  Returns False when the layout is not supported."""
        dev = _req(W, V, Y, blk, D)
        n, k = W.shape
        ok = W.is_contiguous() and D.shape == (blk.numel(), k) and D.stride(1) == 1 and blk.dtype == torch.int64
        if accel:
            ok = ok and V.is_contiguous() and Y.is_contiguous() and V.shape == W.shape == Y.shape
        if not ok or D.dtype != W.dtype:
            return False
        eta_dev, eta_host = None, 0.0
        if isinstance(eta, torch.Tensor):
            _req(eta)
            eta_dev = eta.reshape(1).to(W.dtype)
        else:
            eta_host = float(eta)
        rc = self._lib.rk_sap_update(
            _dt(W), n, k, _ptr(W), _ptr(V), _ptr(Y), k, _ptr(blk.contiguous()), blk.numel(), _ptr(D),
            D.stride(0), _ptr(eta_dev), eta_host, int(bool(accel)), float(beta), float(gamma), float(alpha),
            _stream(dev),
        )
        _lib.check(rc, "rk_sap_update")
        return True

    def small_solve(self, A, B):
        """float64 device solve A X = B, A (k x k), B (k x nrhs), k, nrhs <= 16."""
        dev = _req(A, B)
        if A.dtype != torch.float64 or B.dtype != torch.float64:
            raise ValueError("small_solve works on float64 coefficient matrices")
        A = A.contiguous()
        B = B.contiguous()
        X = torch.empty_like(B)
        rc = self._lib.rk_small_solve(A.shape[0], B.shape[1], _ptr(A), _ptr(B), _ptr(X), _stream(dev))
        _lib.check(rc, "rk_small_solve")
        return X

    def sparse_sign_apply(self, A, col, sgn, row_ptr, scale):
        """scale * Omega @ A for a row-sorted sparse-sign embedding (see include/rk.h)."""
        dev = _req(A, col, sgn, row_ptr)
        A, lda = _rowmajor(A)
        m, d = A.shape
        s = row_ptr.numel() - 1
        if col.dtype != torch.int32 or sgn.dtype != torch.int8 or row_ptr.dtype != torch.int64:
            raise ValueError("col must be int32, sgn int8, row_ptr int64")
        Y = torch.empty((s, d), dtype=A.dtype, device=dev)
        rc = self._lib.rk_sparse_sign_apply(
            _dt(A), _ptr(A), m, d, lda, _ptr(col.contiguous()), _ptr(sgn.contiguous()),
            _ptr(row_ptr.contiguous()), s, float(scale), _ptr(Y), d, _stream(dev),
        )
        _lib.check(rc, "rk_sparse_sign_apply")
        return Y

    def randn(self, rows, cols, dtype, device, seed, offset=0, scale=1.0):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("rlaopt_amd: randn runs on a cuda (ROCm) device only")
        out = torch.empty((rows, cols), dtype=dtype, device=device)
        rc = self._lib.rk_randn(
            _DT[dtype], _ptr(out), rows, cols, cols, int(seed) & (2**64 - 1),
            int(offset) & (2**64 - 1), float(scale), _stream(device),
        )
        _lib.check(rc, "rk_randn")
        return out

    def sample_distinct(self, n, k, device, seed):
        """k distinct uniform draws from [0, n) (int64, on ``device``): ``rk_sample_distinct``."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("rlaopt_amd: sample_distinct runs on a cuda (ROCm) device only")
        out = torch.empty((k,), dtype=torch.int64, device=device)
        rc = self._lib.rk_sample_distinct(int(n), int(k), int(seed) & (2**64 - 1), _ptr(out), _stream(device))
        _lib.check(rc, "rk_sample_distinct")
        return out


def _device_of(args, kwargs):
    for a in args:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    for a in kwargs.values():
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    dev = kwargs.get("device")
    for a in (dev, *args):
        if isinstance(a, (torch.device, str)):
            try:
                a = torch.device(a)
            except (RuntimeError, TypeError):
                continue
            if a.type == "cuda":
                return a
    return None


def _guard_device(fn):
    """Make the operands' device current for the duration of the call: the C side launches on the
    stream it is given but sizes persistent grids / sets function attributes for the CURRENT
    device, and a single process may drive several GPUs (``devices={cuda:0, cuda:1}``)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = _device_of(args, kwargs)
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)

    return wrapped


for _name in ("gram_center", "kernel_materialize", "kernel_matmat", "kernel_sketch_matmat", "sympack_materialize", "symm_matmat",
              "matmat", "rmatmat", "gemm", "gram_tn", "pcg_update_wr", "pcg_update_p", "axpby_cols",
              "normal_pass", "lsqr_tail", "small_solve", "sparse_sign_apply", "randn", "sample_distinct", "power_iteration",
              "sap_update"):
    setattr(HipBackend, _name, _guard_device(getattr(HipBackend, _name)))
del _name

_backend = None


def get_backend():
    """The active compute backend (HipBackend unless a test installed a stand-in)."""
    global _backend
    if _backend is None:
        _backend = HipBackend()  # raises loudly when the extension is missing
    return _backend


@contextlib.contextmanager
def use_backend(backend):
    """Install ``backend`` for the duration of a ``with`` block (test infrastructure: lets
    CPU tests drive the host logic with the oracle; never used by the package itself)."""
    global _backend
    prev = _backend
    _backend = backend
    try:
        yield backend
    finally:
        _backend = prev
