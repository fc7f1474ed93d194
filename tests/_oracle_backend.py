"""CPU stand-in for ``rlaopt_amd.ops.HipBackend`` built on the oracle (TEST ONLY).

It lets the CPU suite drive the package's HOST logic - operator sharding, collectives,
Nystrom small-matrix algebra, PCG/SAP control flow, LinSys cadence - without a GPU.  It is
installed explicitly with ``rlaopt_amd.ops.use_backend(OracleBackend())``; the package never
falls back to it.
"""

import torch

from oracle import kernels as ok


def _two_d(v):
    return (v[:, None], True) if v.ndim == 1 else (v, False)


class OracleBackend:
    name = "oracle-cpu"

    def kernel_materialize(self, kind, X1, X2, lengthscale, scale=1.0, idx1=None, idx2=None, out=None, gram_center=None):
        A1 = X1 if idx1 is None else X1[idx1]
        A2 = X2 if idx2 is None else X2[idx2]
        return ok.kernel_matrix(kind, A1, A2, lengthscale, scale)

    def kernel_matmat(self, kind, X1, X2, lengthscale, V, scale=1.0, idx1=None, idx2=None, gram_center=None,
                      x2_cache=None):
        return ok.kernel_matmat(kind, X1, X2, lengthscale, V, scale, idx1=idx1, idx2=idx2)

    # sympack stand-in: keeps the dense matrix restricted to this rank's strips (upper part)
    def sympack_materialize(self, kind, X, lengthscale, scale=1.0, first=0, stride=1, store_dtype=None, out=None,
                            gram_center=None):
        n = X.shape[0]
        store_dtype = store_dtype or X.dtype
        W = 256 if store_dtype == torch.float32 else 128
        K = ok.kernel_matrix(kind, X, X, lengthscale, scale).to(store_dtype)
        keep = torch.zeros_like(K)
        for s in range(first, -(-n // W), stride):
            lo, hi = s * W, min(n, (s + 1) * W)
            keep[:hi, lo:hi] = K[:hi, lo:hi]
        return keep.reshape(-1)

    def symm_matmat(self, Kp, n, V, alpha=1.0, diag_shift=0.0, first=0, stride=1, tag=None, kmax=None):
        W = 256 if Kp.dtype == torch.float32 else 128
        V2, sq = _two_d(V)
        U = Kp.reshape(n, n).to(V2.dtype)
        Y = U @ V2  # row contributions of every stored entry
        for s in range(first, -(-n // W), stride):  # column contributions from above the diagonal block
            lo, hi = s * W, min(n, (s + 1) * W)
            Y[lo:hi] += U[:lo, lo:hi].T @ V2[:lo]
        Y = alpha * Y + diag_shift * V2
        return Y[:, 0] if sq else Y

    def spmd_pack(self, part, n, step, slots, tail=None):
        k = part.shape[1]
        t = 0 if tail is None else tail.numel()
        out = torch.zeros((slots, step * k + t), dtype=part.dtype)
        rows = torch.zeros((slots * step, k), dtype=part.dtype)
        rows[:n] = part
        out[:, : step * k] = rows.view(slots, step * k)
        if t:
            out[:, step * k :] = tail.reshape(-1).to(part.dtype)
        return out

    def symm_gemm(self, Kp, n, V, kmax=1.0, alpha=1.0, first=0, stride=1, tag=None, rhs_f16=False):
        if rhs_f16:  # (rk_symm_gemm_rhs16 rounds V to the float16 grid)
            V = V.to(torch.float16).to(V.dtype)
        return self.symm_matmat(Kp, n, V, alpha=alpha, first=first, stride=stride)

    def matmat(self, A, V, alpha=1.0, beta=0.0, C=None, diag_shift=0.0, row0=0, tag=None):
        V2, sq = _two_d(V)
        Y = alpha * (A @ V2)
        if beta != 0.0:
            Y = Y + beta * (C[:, None] if C.ndim == 1 else C)
        if diag_shift != 0.0:
            Y = Y + diag_shift * V2[row0 : row0 + A.shape[0]]
        return Y[:, 0] if sq else Y

    def rmatmat(self, A, V, alpha=1.0, beta=0.0, C=None, out_f64=False):
        V2, sq = _two_d(V)
        if out_f64:
            Y = alpha * (A.double().T @ V2.double())
        else:
            Y = alpha * (A.T @ V2)
        if beta != 0.0:
            Y = Y + beta * (C[:, None] if C.ndim == 1 else C)
        return Y[:, 0] if sq else Y

    def gemm(self, A, B, transA=False, transB=False, alpha=1.0):
        return alpha * ((A.T if transA else A) @ (B.T if transB else B))

    def gram_tn(self, A, B):
        return A.double().T @ B.double()

    def pcg_update_wr(self, W, R, P, AP, alpha):
        a = alpha.to(W.dtype)
        W += P @ a
        R -= AP @ a

    def pcg_update_p(self, P, Z, beta):
        P.copy_(Z + P @ beta.to(P.dtype))

    def axpby_cols(self, X, a, Y, b=None):
        av = 1.0 if a is None else a.to(X.dtype)
        new = X * av if b is None else X * av + Y * b.to(X.dtype)
        Y.copy_(new)
        return Y

    def normal_pass(self, A, Z, U, coef):
        Unew = A @ Z - U * coef.to(U.dtype)
        U.copy_(Unew)
        return A.double().T @ Unew.double(), (Unew.double() ** 2).sum(0)

    def small_solve(self, A, B):
        return torch.linalg.solve(A, B)

    def sparse_sign_apply(self, A, col, sgn, row_ptr, scale):
        s = row_ptr.numel() - 1
        Y = torch.zeros((s, A.shape[1]), dtype=A.dtype)
        rows = torch.repeat_interleave(torch.arange(s), row_ptr[1:] - row_ptr[:-1])
        Y.index_add_(0, rows, A[col.long()] * sgn.to(A.dtype)[:, None])
        return scale * Y

    def randn(self, rows, cols, dtype, device, seed, offset=0, scale=1.0):
        g = torch.Generator().manual_seed((int(seed) * 1000003 + int(offset)) % (2**63))
        return scale * torch.randn(rows, cols, dtype=dtype, generator=g).to(device)


def install_for_worker():
    """``init`` hook of ``rlaopt_amd.spmd.run`` in the CPU tests: every worker process computes with the oracle."""
    import rlaopt_amd.ops as ops

    ops._backend = OracleBackend()
