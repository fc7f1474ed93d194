"""GPU parity of every C-ABI entry point against the oracle / a float64 torch reference.
Runs on the MI355X box only (``-m gpu``)."""

import pytest
import torch

from conftest import DTYPES, TOL, load_golden
from oracle import kernels as ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from rlaopt_amd.ops import get_backend

    return get_backend()


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


LIM = {torch.float32: 2e-5, torch.float64: 1e-12}


def _ls64(ls):
    return ls.double() if isinstance(ls, torch.Tensor) else ls


@pytest.mark.parametrize("kind", ok.KINDS)
@pytest.mark.parametrize("ell", ["scalar", "ard"])
@pytest.mark.parametrize("shape", ["small", "ragged"])
def test_materialize_matches_reference_closed_form(be, prec, kind, ell, shape):
    g = load_golden(f"kernels_{prec}")
    X1, X2 = g[f"{shape}_X1"].to(dev()), g[f"{shape}_X2"].to(dev())
    ls = 1.0 if ell == "scalar" else g[f"{shape}_ard"].to(dev())
    K = be.kernel_materialize(kind, X1, X2, ls, scale=2.0)
    torch.testing.assert_close(K.cpu(), g[f"{shape}_{kind}_{ell}"], **TOL[DTYPES[prec]])


@pytest.mark.parametrize("kind", ok.KINDS)
@pytest.mark.parametrize(
    "n1,n2,d,k",
    [(10, 5, 3, 1), (130, 257, 7, 3), (300, 1000, 40, 10), (65, 64, 33, 16), (70, 90, 5, 37),
     # d > 40: the packed 64 x 128 float32 kernel (rows 4-, 8- and 16-byte aligned; one, two and
     # three 64-feature chunks; ragged tiles; every KB instantiation; k > 16 in host chunks)
     (70, 300, 41, 1), (129, 1000, 50, 10), (64, 128, 64, 16), (200, 515, 64, 2), (90, 260, 100, 4),
     (65, 129, 130, 8), (33, 2100, 52, 37)],
)
def test_kernel_matmat_and_materialize_vs_oracle(be, prec, kind, n1, n2, d, k):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(n1 * 7 + n2)
    X1 = torch.randn(n1, d, dtype=dt, generator=gen) / d**0.5
    X2 = torch.randn(n2, d, dtype=dt, generator=gen) / d**0.5
    V = torch.randn(n2, k, dtype=dt, generator=gen)
    ard = torch.rand(d, dtype=dt, generator=gen) + 0.5
    for ls in (0.7, ard):
        want = ok.kernel_matmat(kind, X1.double(), X2.double(), _ls64(ls), V.double(), 1.5)
        lsd = ls.to(dev()) if isinstance(ls, torch.Tensor) else ls
        got = be.kernel_matmat(kind, X1.to(dev()), X2.to(dev()), lsd, V.to(dev()), scale=1.5)
        assert rel_err(got.cpu(), want) < LIM[dt]
        Kd = be.kernel_materialize(kind, X1.to(dev()), X2.to(dev()), lsd, scale=1.5)
        Kw = ok.kernel_matrix(kind, X1.double(), X2.double(), _ls64(ls), 1.5)
        assert rel_err(Kd.cpu(), Kw) < LIM[dt]
    # 1-D right-hand side
    got = be.kernel_matmat(kind, X1.to(dev()), X2.to(dev()), 0.7, V[:, 0].contiguous().to(dev()))
    want = ok.kernel_matmat(kind, X1.double(), X2.double(), 0.7, V[:, 0].double())
    assert got.ndim == 1 and rel_err(got.cpu(), want) < LIM[dt]


def test_kernel_index_oracles(be, prec):
    """row oracle K(X[blk], X) and block oracle K(X[blk], X[blk]) gathers (base.py:104-128)."""
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(3)
    _index_oracle_case(be, dt, gen, 500, 9, 4)
    _index_oracle_case(be, dt, gen, 700, 50, 10)  # packed wide kernel, 8-byte aligned rows
    _index_oracle_case(be, dt, gen, 400, 64, 3)   # ... 16-byte aligned rows (coalesced staging)


def _index_oracle_case(be, dt, gen, n, d, k):
    X = torch.randn(n, d, dtype=dt, generator=gen) / 3
    blk = torch.randperm(n, generator=gen)[:77]
    V = torch.randn(n, k, dtype=dt, generator=gen)
    Vb = torch.randn(77, k, dtype=dt, generator=gen)
    Xd, bd = X.to(dev()), blk.to(dev())
    for kind in ("rbf", "matern52"):
        got = be.kernel_matmat(kind, Xd, Xd, 1.0, V.to(dev()), idx1=bd)
        want = ok.kernel_matmat(kind, X.double(), X.double(), 1.0, V.double(), idx1=blk)
        assert rel_err(got.cpu(), want) < LIM[dt]
        got = be.kernel_matmat(kind, Xd, Xd, 1.0, Vb.to(dev()), idx1=bd, idx2=bd)
        want = ok.kernel_matmat(kind, X.double(), X.double(), 1.0, Vb.double(), idx1=blk, idx2=blk)
        assert rel_err(got.cpu(), want) < LIM[dt]
        Kb = be.kernel_materialize(kind, Xd, Xd, 1.0, idx1=bd, idx2=bd)
        assert rel_err(Kb.cpu(), ok.kernel_matrix(kind, X[blk].double(), X[blk].double(), 1.0)) < LIM[dt]


@pytest.mark.parametrize("m,n", [(1, 1), (7, 5), (64, 256), (100, 1000), (257, 4099), (1000, 9000), (3000, 70000)])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 10, 16])
def test_matmat_skinny(be, prec, m, n, k):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(m + n + k)
    A = torch.randn(m, n, dtype=dt, generator=gen)
    V = torch.randn(n, k, dtype=dt, generator=gen)
    C = torch.randn(m, k, dtype=dt, generator=gen)
    Ad = A.to(dev())
    got = be.matmat(Ad, V.to(dev()), alpha=-0.5, beta=2.0, C=C.to(dev()))
    want = -0.5 * (A.double() @ V.double()) + 2.0 * C.double()
    assert rel_err(got.cpu(), want) < LIM[dt] * 5
    got = be.matmat(Ad, V.to(dev()))
    assert rel_err(got.cpu(), A.double() @ V.double()) < LIM[dt] * 5
    # transpose product, float64 output
    U = torch.randn(m, k, dtype=dt, generator=gen)
    got = be.rmatmat(Ad, U.to(dev()), out_f64=True)
    assert got.dtype == torch.float64
    assert rel_err(got.cpu(), A.double().T @ U.double()) < LIM[dt] * 5
    got = be.rmatmat(Ad, U.to(dev()), alpha=2.0)
    assert got.dtype == dt and rel_err(got.cpu(), 2 * A.double().T @ U.double()) < LIM[dt] * 5


def test_matmat_skinny_diag_shift_and_unaligned(be, prec):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(5)
    n, m, row0, k = 999, 301, 200, 3
    Afull = torch.randn(n, n + 1, dtype=dt, generator=gen)  # odd leading dimension
    A = Afull[row0 : row0 + m, :n]
    V = torch.randn(n, k, dtype=dt, generator=gen)
    Ad = Afull.to(dev())[row0 : row0 + m, :n]  # genuinely strided, unaligned rows on the device
    got = be.matmat(Ad, V.to(dev()), diag_shift=0.25, row0=row0)
    want = A.double() @ V.double() + 0.25 * V[row0 : row0 + m].double()
    assert rel_err(got.cpu(), want) < LIM[dt] * 5
    got = be.matmat(Ad, V[:, 0].contiguous().to(dev()))
    assert got.ndim == 1 and rel_err(got.cpu(), A.double() @ V[:, 0].double()) < LIM[dt] * 5
    got = be.rmatmat(Ad, torch.ones(m, 2, dtype=dt, device=dev()))
    assert rel_err(got.cpu(), A.double().T @ torch.ones(m, 2, dtype=torch.float64)) < LIM[dt] * 5


@pytest.mark.parametrize("m,n,kd", [(128, 128, 16), (200, 96, 50), (1000, 512, 777), (130, 260, 1030), (33, 17, 5)])
@pytest.mark.parametrize("tA", [False, True])
@pytest.mark.parametrize("tB", [False, True])
def test_gemm_mfma(be, prec, m, n, kd, tA, tB):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(m * 3 + n + kd)
    # asymmetric integer-valued operands catch swapped lane maps exactly
    A = torch.randint(-3, 4, (kd, m) if tA else (m, kd), generator=gen).to(dt)
    B = torch.randint(-3, 4, (n, kd) if tB else (kd, n), generator=gen).to(dt)
    got = be.gemm(A.to(dev()), B.to(dev()), transA=tA, transB=tB)
    want = (A.T if tA else A).double() @ (B.T if tB else B).double()
    assert torch.equal(got.cpu().double(), want)
    A = torch.randn(A.shape, dtype=dt, generator=gen)
    B = torch.randn(B.shape, dtype=dt, generator=gen)
    got = be.gemm(A.to(dev()), B.to(dev()), transA=tA, transB=tB, alpha=0.5)
    want = 0.5 * (A.T if tA else A).double() @ (B.T if tB else B).double()
    assert rel_err(got.cpu(), want) < LIM[dt] * 5


def test_matmat_wide_goes_through_gemm(be, prec):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(9)
    A = torch.randn(300, 420, dtype=dt, generator=gen)
    V = torch.randn(420, 40, dtype=dt, generator=gen)
    U = torch.randn(300, 40, dtype=dt, generator=gen)
    assert rel_err(be.matmat(A.to(dev()), V.to(dev())).cpu(), A.double() @ V.double()) < LIM[dt] * 5
    assert rel_err(be.rmatmat(A.to(dev()), U.to(dev())).cpu(), A.double().T @ U.double()) < LIM[dt] * 5
    got = be.rmatmat(A.to(dev()), U.to(dev()), out_f64=True)
    assert got.dtype == torch.float64 and rel_err(got.cpu(), A.double().T @ U.double()) < 1e-13


@pytest.mark.parametrize("rows,m,n", [(100, 8, 8), (5000, 64, 48), (20000, 200, 200), (777, 130, 5)])
def test_gram_tn_float64_accumulation(be, prec, rows, m, n):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(rows + m)
    A = torch.randn(rows, m, dtype=dt, generator=gen)
    B = torch.randn(rows, n, dtype=dt, generator=gen)
    got = be.gram_tn(A.to(dev()), B.to(dev()))
    assert got.dtype == torch.float64
    assert rel_err(got.cpu(), A.double().T @ B.double()) < 1e-13  # f64 accumulation of exact inputs


@pytest.mark.parametrize("k", [1, 2, 3, 7, 16])
def test_pcg_updates_and_small_solve(be, prec, k):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(k)
    n = 1234
    W, R, P, AP, Z = (torch.randn(n, k, dtype=dt, generator=gen) for _ in range(5))
    al = torch.randn(k, k, dtype=torch.float64, generator=gen)
    Wd, Rd, Pd = W.to(dev()), R.to(dev()), P.to(dev())
    be.pcg_update_wr(Wd, Rd, Pd, AP.to(dev()), al.to(dev()))
    assert rel_err(Wd.cpu(), W.double() + P.double() @ al) < LIM[dt]
    assert rel_err(Rd.cpu(), R.double() - AP.double() @ al) < LIM[dt]
    be.pcg_update_p(Pd, Z.to(dev()), al.to(dev()))
    assert rel_err(Pd.cpu(), Z.double() + P.double() @ al) < LIM[dt]
    M = torch.randn(k, k, dtype=torch.float64, generator=gen)
    M = M @ M.T + 0.1 * torch.eye(k, dtype=torch.float64)
    M[0], M[k - 1] = M[k - 1].clone(), M[0].clone()  # force pivoting
    Bm = torch.randn(k, min(k + 1, 16), dtype=torch.float64, generator=gen)
    X = be.small_solve(M.to(dev()), Bm.to(dev()))
    assert rel_err(X.cpu(), torch.linalg.solve(M, Bm)) < 1e-10


def test_randn_statistics_and_reproducibility(be, prec):
    dt = DTYPES[prec]
    a = be.randn(1000, 513, dt, dev(), seed=123, offset=0, scale=2.0)
    b = be.randn(1000, 513, dt, dev(), seed=123, offset=0, scale=2.0)
    c = be.randn(1000, 513, dt, dev(), seed=124, offset=0, scale=2.0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = a.double().cpu() / 2.0
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.std()) - 1) < 5e-3
    assert abs(float((x**4).mean()) - 3) < 0.05 and torch.isfinite(x).all()
    assert abs(float((x[0] * x[1]).mean())) < 0.2


def test_bad_arguments_raise(be):
    A = torch.zeros(4, 4, device=dev())
    with pytest.raises(ValueError):
        be.matmat(A, torch.zeros(5, 1, device=dev()))
    with pytest.raises(RuntimeError):
        be.matmat(A.cpu(), torch.zeros(4, 1))
    with pytest.raises(ValueError):
        be.kernel_materialize("rbf", A, A, -1.0)
    with pytest.raises(ValueError):
        be.matmat(A.to(torch.float16), torch.zeros(4, 1, device=dev(), dtype=torch.float16))


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
@pytest.mark.parametrize("n", [1, 63, 256, 257, 700, 1024, 1025, 3000])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 10, 16])
def test_sympack_product_matches_dense(be, prec, kind, n, k):
    """Strip-packed symmetric storage: y = K v from half the bytes equals the dense product, for
    every strip distribution (first, stride) - partial sums of the ranks add up to K v."""
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(n + k)
    d = 5
    X = torch.randn(n, d, dtype=dt, generator=gen) / d**0.5
    V = torch.randn(n, k, dtype=dt, generator=gen)
    K = ok.kernel_matrix(kind, X.double(), X.double(), 0.8, 1.5)
    want = K @ V.double()
    Xd, Vd = X.to(dev()), V.to(dev())
    for stride in (1, 2, 3):
        total = torch.zeros(n, k, dtype=torch.float64)
        for first in range(stride):
            Kp = be.sympack_materialize(kind, Xd, 0.8, scale=1.5, first=first, stride=stride)
            part = be.symm_matmat(Kp, n, Vd, first=first, stride=stride)
            total += part.cpu().double()
        assert rel_err(total, want) < LIM[dt] * 5, (stride,)
    Kp = be.sympack_materialize(kind, Xd, 0.8, scale=1.5)
    got = be.symm_matmat(Kp, n, Vd, alpha=-2.0, diag_shift=0.25)
    assert rel_err(got.cpu(), -2.0 * want + 0.25 * V.double()) < LIM[dt] * 5
    got1 = be.symm_matmat(Kp, n, Vd[:, 0].contiguous())
    assert got1.ndim == 1 and rel_err(got1.cpu(), want[:, 0]) < LIM[dt] * 5
    # reproducible bit for bit
    assert torch.equal(be.symm_matmat(Kp, n, Vd), be.symm_matmat(Kp, n, Vd))
    # per-feature (ARD) lengthscale
    ard = (torch.rand(d, dtype=dt, generator=gen) + 0.5)
    Kp = be.sympack_materialize(kind, Xd, ard.to(dev()), scale=1.0)
    want = ok.kernel_matrix(kind, X.double(), X.double(), ard.double(), 1.0) @ V.double()
    assert rel_err(be.symm_matmat(Kp, n, Vd).cpu(), want) < LIM[dt] * 5


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("n,d,stride,ard", [(1, 3, 1, False), (255, 9, 1, False), (256, 64, 2, True), (257, 33, 1, False),
                                          (1300, 50, 3, False), (4097, 64, 1, True), (5000, 7, 5, False)])
def test_sympack_written_from_the_matrix_cores_matches_the_vector_unit_kernel(be, monkeypatch, kind, n, d, stride, ard):
    """rk_kernel_materialize_sympack_mc: the packed triangle evaluated on the matrix cores (split points, three exact
    products per block, the kernel function on the accumulators) against the vector-unit kernel entry by entry (same
    layout, same zeros past column n) and against the float64 oracle; every strip distribution; data with a common
    offset (the centre removes it); reproducible bit for bit; off-bound data and RK_SYMPACK_MC=0 take the old kernel."""
    dt = torch.float32
    g = torch.Generator().manual_seed(3 * n + d)
    X = torch.randn(n, d, dtype=dt, generator=g) / d**0.5 + 2.0
    ell = (0.7 + 0.6 * torch.rand(d, dtype=dt, generator=g)) if ard else 0.9
    Xd = X.to(dev())
    elld = ell.to(dev()) if ard else ell
    gc = be.gram_center(Xd, Xd, elld)
    assert gc is not None
    monkeypatch.setenv("RK_SYMPACK_MC_NORM2", "32")  # (every data set inside the norm bound, whatever its spread)
    # entry error: 2^-22 |x'||y'| in the dot product -> ~7e-7 max|x'|^2 of the kernel's scale, + float32 rounding
    tol = (7e-7 * gc._rk_norm2 + 1e-6) * 1.7
    Kd = ok.kernel_matrix(kind, X.double(), X.double(), ell.double() if ard else ell, 1.7)
    for first in range(stride):
        if be.sympack_elems(n, first, stride) == 0:
            continue  # (this rank holds no strip)
        new = be.sympack_materialize(kind, Xd, elld, scale=1.7, first=first, stride=stride, gram_center=gc)
        assert torch.equal(new, be.sympack_materialize(kind, Xd, elld, scale=1.7, first=first, stride=stride))  # ("auto")
        monkeypatch.setenv("RK_SYMPACK_MC", "0")
        old = be.sympack_materialize(kind, Xd, elld, scale=1.7, first=first, stride=stride)
        monkeypatch.delenv("RK_SYMPACK_MC")
        assert new.shape == old.shape
        assert torch.equal(new == 0, old == 0) or float((new - old).abs().max()) < 1e-5  # (the zero padding sits in the same places)
        assert float((new - old).abs().max()) < tol, (first, float((new - old).abs().max()), gc._rk_norm2)
        # against the float64 matrix, strip by strip
        off = 0
        for s in range(first, -(-n // 256), stride):
            rows, lo, hi = min(n, (s + 1) * 256), s * 256, min(n, (s + 1) * 256)
            blk = new[off : off + rows * 256].view(rows, 256).cpu().double()
            assert float((blk[:, : hi - lo] - Kd[:rows, lo:hi]).abs().max()) < tol
            assert float(blk[:, hi - lo :].abs().max()) == 0.0 if hi - lo < 256 else True
            off += rows * 256
    # data outside the norm bound: no centre, the vector-unit kernel
    if n >= 255:
        far = Xd * 40.0
        assert be.gram_center(far, far, 0.9) is None
        Kp = be.sympack_materialize(kind, far, 0.9)
        mid = Xd * 4.0  # inside the bound of the matrix-free product, outside the budget for stored entries
        gm = be.gram_center(mid, mid, 0.9)
        monkeypatch.setenv("RK_SYMPACK_MC_NORM2", "8")
        Km = be.sympack_materialize(kind, mid, 0.9) if gm is not None and gm._rk_norm2 > 8 else None
        monkeypatch.setenv("RK_SYMPACK_MC", "0")
        assert torch.equal(Kp, be.sympack_materialize(kind, far, 0.9))
        if Km is not None:
            assert torch.equal(Km, be.sympack_materialize(kind, mid, 0.9))


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("n1,n2,d,gather,ard", [(1, 1, 3, False, False), (300, 300, 9, False, True), (5000, 5000, 50, True, False),
                                                (777, 1301, 33, False, False), (1500, 515, 64, True, True), (129, 2051, 5, True, False)])
def test_row_major_block_written_from_the_matrix_cores(be, monkeypatch, kind, n1, n2, d, gather, ard):
    """rk_kernel_materialize_mc: the dense row-major block evaluated on the matrix cores (the ASkotch blocks K(X[blk], X[blk]),
    row-major shards) against the vector-unit kernel and the float64 oracle: square and rectangular, gathered rows, widths
    that are not multiples of 4 (element stores) and views with a wider leading dimension."""
    dt = torch.float32
    g = torch.Generator().manual_seed(n1 + 7 * n2 + d)
    N = max(n1, n2) * (3 if gather else 1)
    X = torch.randn(N, d, dtype=dt, generator=g) / d**0.5 - 1.5
    ell = (0.7 + 0.6 * torch.rand(d, dtype=dt, generator=g)) if ard else 1.1
    Xd = X.to(dev())
    elld = ell.to(dev()) if ard else ell
    gc = be.gram_center(Xd, Xd, elld)
    assert gc is not None
    monkeypatch.setenv("RK_SYMPACK_MC_NORM2", "32")
    tol = (7e-7 * gc._rk_norm2 + 1e-6) * 0.6
    if gather:
        i1 = torch.randperm(N, generator=g)[:n1]
        i2 = i1 if n1 == n2 else torch.randperm(N, generator=g)[:n2]
        X1, X2 = Xd, Xd
        new = be.kernel_materialize(kind, Xd, Xd, elld, scale=0.6, idx1=i1.to(dev()), idx2=i2.to(dev()), gram_center=gc)
        old = be.kernel_materialize(kind, Xd, Xd, elld, scale=0.6, idx1=i1.to(dev()), idx2=i2.to(dev()))
        want = ok.kernel_matrix(kind, X[i1].double(), X[i2].double(), ell.double() if ard else ell, 0.6)
    else:
        X1, X2 = Xd[:n1], (Xd[:n2] if n1 == n2 else Xd[N - n2 :])
        new = be.kernel_materialize(kind, X1, X2, elld, scale=0.6, gram_center=gc)
        old = be.kernel_materialize(kind, X1, X2, elld, scale=0.6)
        want = ok.kernel_matrix(kind, X1.cpu().double(), X2.cpu().double(), ell.double() if ard else ell, 0.6)
    assert new.shape == (n1, n2)
    assert float((new - old).abs().max()) < tol, float((new - old).abs().max())
    assert float((new.cpu().double() - want).abs().max()) < tol
    assert torch.equal(new, be.kernel_materialize(kind, X1, X2, elld, scale=0.6, gram_center=gc,
                                                   **({"idx1": i1.to(dev()), "idx2": i2.to(dev())} if gather else {})))
    # into a view with a wider leading dimension; the columns beside it stay untouched
    wide = torch.full((n1, n2 + 5), 7.0, dtype=dt, device=dev())
    be.kernel_materialize(kind, X1, X2, elld, scale=0.6, gram_center=gc, out=wide[:, 2 : 2 + n2],
                          **({"idx1": i1.to(dev()), "idx2": i2.to(dev())} if gather else {}))
    assert torch.equal(wide[:, 2 : 2 + n2], new) and float((wide[:, :2] - 7.0).abs().max()) == 0.0
    assert float((wide[:, 2 + n2 :] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("n", [1, 255, 4097, 300000])
@pytest.mark.parametrize("k1,k2", [(1, 1), (2, 3), (4, 4), (3, 1)])
def test_gram_small_block_inner_products(be, prec, n, k1, k2):
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(n + k1)
    X = torch.randn(n, k1, dtype=dt, generator=gen)
    Y = torch.randn(n, k2, dtype=dt, generator=gen)
    got = be.rmatmat(X.to(dev()), Y.to(dev()), out_f64=True)
    assert got.dtype == torch.float64 and got.shape == (k1, k2)
    want = X.double().T @ Y.double()
    assert float((got.cpu() - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max())) * n**0.5
    assert torch.equal(got, be.rmatmat(X.to(dev()), Y.to(dev()), out_f64=True))  # reproducible
    # strided views (columns of a wider array)
    Z = torch.randn(n, 6, dtype=dt, generator=gen).to(dev())
    got = be.rmatmat(Z[:, 1:3], Z[:, 2:5], out_f64=True)
    want = Z[:, 1:3].double().T @ Z[:, 2:5].double()
    assert float((got - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max())) * n**0.5


@pytest.mark.parametrize("m,n,k,lda_pad", [(3, 8, 1, 0), (1000, 512, 1, 0), (4099, 128, 2, 0), (777, 100, 1, 4),
                                           (5000, 256, 2, 8), (260, 2, 1, 0), (9000, 384, 1, 0),
                                           # three 16-byte steps per lane run the four-step instantiation: its
                                           # last column chunk starts at the end of the row (lda == n)
                                           (777, 768, 1, 0), (2051, 768, 2, 0), (1029, 640, 1, 0), (515, 320, 2, 0)])
def test_normal_pass_one_sweep(be, prec, m, n, k, lda_pad):
    """rk_normal_pass: U <- A Z - U diag(coef), G = A^T U_new, ss = |U_new|^2 from one pass."""
    dt = DTYPES[prec]
    vw = 4 if dt == torch.float32 else 2
    if n % vw or (n * (4 if dt == torch.float32 else 8)) > 4096:
        pytest.skip("shape outside the fused kernel (covered by test_normal_pass_unsupported_shapes)")
    gen = torch.Generator().manual_seed(m + n + k)
    Afull = torch.randn(m, n + lda_pad, dtype=dt, generator=gen)
    A = Afull[:, :n]
    Z = torch.randn(n, k, dtype=dt, generator=gen)
    U = torch.randn(m, k, dtype=dt, generator=gen)
    coef = torch.rand(k, dtype=torch.float64, generator=gen) + 0.5
    Ad, Ud = Afull.to(dev())[:, :n], U.clone().to(dev())
    out = be.normal_pass(Ad, Z.to(dev()), Ud, coef.to(dev()))
    assert out is not None
    G, ss = out
    want_U = A.double() @ Z.double() - U.double() * coef
    want_G = A.double().T @ want_U
    assert G.dtype == torch.float64 and ss.dtype == torch.float64
    assert rel_err(Ud.cpu(), want_U) < LIM[dt] * 5
    assert rel_err(G.cpu(), want_G) < LIM[dt] * 20
    assert rel_err(ss.cpu(), (want_U**2).sum(0)) < LIM[dt] * 20
    # bitwise reproducible (fixed-order partial sums)
    Ud2 = U.clone().to(dev())
    G2, ss2 = be.normal_pass(Ad, Z.to(dev()), Ud2, coef.to(dev()))
    assert torch.equal(G, G2) and torch.equal(ss, ss2) and torch.equal(Ud, Ud2)


def test_normal_pass_unsupported_shapes(be):
    """Shapes outside the fused kernel return None (the caller then does the two products)."""
    A = torch.randn(100, 1026, dtype=torch.float64, device=dev())  # row longer than 4 KiB
    Z = torch.randn(1026, 1, dtype=torch.float64, device=dev())
    U = torch.randn(100, 1, dtype=torch.float64, device=dev())
    c = torch.ones(1, dtype=torch.float64, device=dev())
    assert be.normal_pass(A, Z, U, c) is None
    A3 = torch.randn(100, 64, dtype=torch.float32, device=dev())  # k = 3
    assert be.normal_pass(A3, torch.randn(64, 3, device=dev()), torch.randn(100, 3, device=dev()),
                          torch.ones(3, dtype=torch.float64, device=dev())) is None
    A5 = torch.randn(100, 66, dtype=torch.float32, device=dev())  # n not a multiple of 4
    assert be.normal_pass(A5, torch.randn(66, 1, device=dev()), torch.randn(100, 1, device=dev()), c) is None


@pytest.mark.parametrize("n1,n2,d,k", [(129, 1000, 50, 10), (300, 515, 64, 1), (64, 128, 33, 16), (500, 2100, 8, 4),
                                       (257, 700, 100, 1), (130, 4500, 128, 3), (200, 333, 65, 10), (300, 400, 97, 2)])
def test_kernel_matmat_matrix_core_formulation(be, monkeypatch, n1, n2, d, k):
    """The matrix-core formulation (float32 kernels smooth in the squared distance, d <= 64, data
    inside the norm bound: S = X2' X1'^T on the fp32 MFMA, |x'|^2 + |y'|^2 - 2 S, kappa on the
    accumulators, Y^T += V^T P^T) against the oracle and against the direct-difference kernels
    (RK_KMAT_MFMA=0), with and without index gathers, scalar and per-feature lengthscales."""
    dt = torch.float32
    gen = torch.Generator().manual_seed(n1 + n2 + d)
    X1 = torch.randn(n1, d, dtype=dt, generator=gen) / d**0.5
    X2 = torch.randn(n2, d, dtype=dt, generator=gen) / d**0.5
    V = torch.randn(n2, k, dtype=dt, generator=gen)
    blk = torch.randperm(n2, generator=gen)[: min(n1, 97)]
    blk2 = torch.randperm(n2, generator=gen)[: max(130, n2 // 2)]
    ard = 0.6 + 0.4 * torch.rand(d, dtype=dt, generator=gen)
    for kind, ls in (("rbf", 0.8), ("matern32", 0.8), ("rbf", ard), ("matern52", ard)):
        ls64 = _ls64(ls)
        lsd = ls.to(dev()) if isinstance(ls, torch.Tensor) else ls
        want = ok.kernel_matmat(kind, X1.double(), X2.double(), ls64, V.double(), 1.5)
        want_g = ok.kernel_matmat(kind, X2.double(), X2.double(), ls64, V.double(), 1.5, idx1=blk)
        want_gg = ok.kernel_matmat(kind, X2.double(), X2.double(), ls64, V[: blk2.numel()].double(), 1.5,
                                   idx1=blk, idx2=blk2)
        assert be.gram_center(X1.to(dev()), X2.to(dev()), lsd) is not None
        monkeypatch.setenv("RK_KMAT_MFMA", "0")
        base = be.kernel_matmat(kind, X1.to(dev()), X2.to(dev()), lsd, V.to(dev()), scale=1.5)
        monkeypatch.delenv("RK_KMAT_MFMA")
        got = be.kernel_matmat(kind, X1.to(dev()), X2.to(dev()), lsd, V.to(dev()), scale=1.5)
        got_g = be.kernel_matmat(kind, X2.to(dev()), X2.to(dev()), lsd, V.to(dev()), scale=1.5, idx1=blk.to(dev()))
        got_gg = be.kernel_matmat(kind, X2.to(dev()), X2.to(dev()), lsd, V[: blk2.numel()].to(dev()), scale=1.5,
                                  idx1=blk.to(dev()), idx2=blk2.to(dev()))
        assert rel_err(got.cpu(), want) < LIM[dt]
        assert rel_err(got_g.cpu(), want_g) < LIM[dt]
        assert rel_err(got_gg.cpu(), want_gg) < LIM[dt]
        assert rel_err(got.cpu(), base.cpu().double()) < LIM[dt]
        assert not torch.equal(got, base)  # (a different kernel really ran)


def _columns(be, kind, X1d, X2d, ls, cols, idx1=None, idx2=None, **kw):
    """Columns ``cols`` of K(X1[idx1], X2[idx2]) through the matrix-free product (unit vectors)."""
    n2 = X2d.shape[0] if idx2 is None else idx2.numel()
    E = torch.zeros(n2, len(cols), dtype=X1d.dtype, device=X1d.device)
    E[torch.as_tensor(cols, device=X1d.device), torch.arange(len(cols), device=X1d.device)] = 1.0
    return be.kernel_matmat(kind, X1d, X2d, ls, E, idx1=idx1, idx2=idx2, **kw)


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("ell", ["scalar", "ard"])
@pytest.mark.parametrize("data", ["unit", "offset", "unit_d100"])
def test_matrix_core_formulation_entrywise_on_the_diagonal(be, monkeypatch, kind, ell, data):
    """ENTRY-WISE agreement of the matrix-core formulation with the direct-difference evaluation
    (``kernel_materialize``: what pykeops computes) for X1 is X2 - the diagonal K(x, x) = 1, where
    |x|^2 + |y|^2 - 2 x.y cancels completely - on centred unit-norm data and on un-centred data
    (mean 10, std 1, lengthscale 8: without the centring of the panels the expansion is off by
    1e-2), with scalar and per-feature lengthscales, with and without index gathers."""
    dt = torch.float32
    n, d, k = 700, (100 if data == "unit_d100" else 64), 16
    gen = torch.Generator().manual_seed(11)
    if data != "offset":
        X = torch.randn(n, d, dtype=dt, generator=gen) / d**0.5
        base_ell = 1.0
    else:
        X = 10.0 + torch.randn(n, d, dtype=dt, generator=gen)
        base_ell = 8.0
    ls = base_ell if ell == "scalar" else (base_ell * (0.75 + 0.5 * torch.rand(d, dtype=dt, generator=gen))).to(dev())
    Xd = X.to(dev())
    gc = be.gram_center(Xd, Xd, ls)
    assert gc is not None  # the data qualify: the matrix-core kernel is the one under test
    K = be.kernel_materialize(kind, Xd, Xd, ls)
    cols = list(range(0, n, n // k))[:k]
    got = _columns(be, kind, Xd, Xd, ls, cols, gram_center=gc)
    monkeypatch.setenv("RK_KMAT_MFMA", "0")
    direct = _columns(be, kind, Xd, Xd, ls, cols, gram_center=gc)
    monkeypatch.delenv("RK_KMAT_MFMA")
    assert not torch.equal(got, direct)  # two different kernels ran
    torch.testing.assert_close(direct, K[:, cols], rtol=0, atol=2e-6)
    torch.testing.assert_close(got, K[:, cols], rtol=0, atol=2e-5)  # reference tolerance: 1e-4
    diag = got[cols, torch.arange(len(cols))]
    torch.testing.assert_close(diag, torch.ones_like(diag), rtol=0, atol=2e-5)
    # gathered: block oracle K(X[blk], X[blk]) (kernels/base.py:104-128), diagonal included
    blk = torch.randperm(n, generator=gen)[:300].to(dev())
    Kb = K[blk][:, blk]
    cb = list(range(0, 300, 300 // k))[:k]
    got_b = _columns(be, kind, Xd, Xd, ls, cb, idx1=blk, idx2=blk, gram_center=gc)
    torch.testing.assert_close(got_b, Kb[:, cb], rtol=0, atol=2e-5)


def test_matrix_core_formulation_is_declined_when_unsafe(be, monkeypatch):
    """Matern-1/2 (not smooth in the squared distance), data whose centred norms exceed
    RK_GRAM_MAX_NORM2, and a missing centre all run the direct-difference kernels: results are
    bitwise those of RK_KMAT_MFMA=0."""
    from rlaopt_amd import _lib

    dt = torch.float32
    gen = torch.Generator().manual_seed(5)
    n, d = 400, 32
    X = (torch.randn(n, d, dtype=dt, generator=gen) / d**0.5).to(dev())
    V = torch.randn(n, 3, dtype=dt, generator=gen).to(dev())
    wide = (torch.randn(n, d, dtype=dt, generator=gen) * 3.0).to(dev())  # |x|^2 ~ 9 d = 288 > 32
    assert be.gram_center(X, X, 1.0) is not None
    assert be.gram_center(wide, wide, 1.0) is None
    assert be.gram_center(wide, wide, 10.0) is not None  # the bound is on |x / ell|^2
    assert be.gram_center(X, wide, 1.0) is None  # both point sets count
    assert be.gram_center(X.double(), X.double(), 1.0) is None
    bad = X.clone()
    bad[7, 3] = float("nan")
    assert be.gram_center(bad, bad, 1.0) is None
    assert _lib.GRAM_MAX_NORM2 == 32.0
    gc = be.gram_center(X, X, 1.0)
    cases = [("matern12", X, gc), ("rbf", X, None), ("rbf", wide, "auto"), ("laplace", X, gc)]
    for kind, P, c in cases:
        got = be.kernel_matmat(kind, P, P, 1.0, V, gram_center=c)
        monkeypatch.setenv("RK_KMAT_MFMA", "0")
        want = be.kernel_matmat(kind, P, P, 1.0, V, gram_center=c)
        monkeypatch.delenv("RK_KMAT_MFMA")
        assert torch.equal(got, want), kind
    with pytest.raises(ValueError):
        be.kernel_matmat("rbf", X, X, 1.0, V, gram_center=gc[:5])


def test_gram_center_and_bound_values(be):
    """rk_kernel_center / rk_kernel_gram_bound against torch: sampled mean (stride n // 4096) and
    the exact maximum of the centred, scaled squared norms."""
    gen = torch.Generator().manual_seed(3)
    for n, d in ((1000, 7), (10000, 64), (5, 3)):
        X = (2.0 + torch.randn(n, d, dtype=torch.float32, generator=gen)).to(dev())
        ell = (0.5 + torch.rand(d, dtype=torch.float32, generator=gen)).to(dev())
        stride = n // 4096 if n > 4096 else 1
        mean = X[::stride].double().mean(0)
        bound_ref = float((((X.double() - mean) / ell.double()) ** 2).sum(1).max())
        lib = be._lib
        center = torch.empty(d, dtype=torch.float32, device=dev())
        bound = torch.zeros(1, dtype=torch.float32, device=dev())
        assert lib.rk_kernel_center(0, X.data_ptr(), n, d, d, center.data_ptr(), None) == 0
        assert lib.rk_kernel_gram_bound(0, X.data_ptr(), n, d, d, 1.0, ell.data_ptr(), center.data_ptr(),
                                        bound.data_ptr(), None) == 0
        torch.cuda.synchronize()
        torch.testing.assert_close(center.double().cpu(), mean.cpu(), rtol=1e-5, atol=1e-6)
        assert abs(float(bound) - bound_ref) <= 1e-4 * bound_ref


@pytest.mark.parametrize("precond", ["nystrom_dense", "nystrom_diag", "identity"])
def test_power_iteration_matches_the_reference_loop(be, prec, precond):
    """rk_power_iteration (whole loop on the device, stopping rule frozen in device state) against
    the reference-shaped host loop (spectral_estimators/spectral_norm.py:11-29) on the same operator
    Pinv (K + reg I): same eigenvalue estimate, same number of iterations."""
    from rlaopt_amd.linops import LinOp
    from rlaopt_amd.spectral_estimators import randomized_powering

    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(17)
    m, r, d = 700, 24, 5
    X = torch.randn(m, d, dtype=torch.float64, generator=gen) / d**0.5
    K = torch.exp(-0.5 * torch.cdist(X, X) ** 2).to(dt).to(dev())
    U = torch.linalg.qr(torch.randn(m, r, dtype=torch.float64, generator=gen)).Q.to(dt).to(dev())
    reg, rho = 1e-2, 0.3
    v0 = torch.randn(m, dtype=torch.float64, generator=gen).to(dt).to(dev())
    if precond == "nystrom_dense":
        A = torch.randn(r, r, dtype=torch.float64, generator=gen)
        M = (torch.linalg.inv(A @ A.T + r * torch.eye(r, dtype=torch.float64)) * r * 0.7).to(dev())
    elif precond == "nystrom_diag":
        M = torch.rand(r, dtype=torch.float64, generator=gen).to(dev())
    else:
        M, U, rho = None, None, 1.0

    def pinv(y):
        if U is None:
            return y
        t = U.T.double() @ y.double()
        t = M @ t if M.ndim == 2 else M * t
        return ((y.double() - U.double() @ t) / rho).to(dt)

    S = LinOp(device=dev(), shape=torch.Size((m, m)), matvec=lambda v: pinv(K @ v + reg * v), dtype=dt)
    for max_iters, rtol in ((10, 1e-3), (3, 1e-3), (40, 1e-7)):
        want, _ = randomized_powering(S, max_iters=max_iters, rtol=rtol, v0=v0.clone())
        # iterations the reference loop used
        v, sig, its, err = v0 / v0.norm(), 0.0, 0, float("inf")
        while its < max_iters and err > rtol * sig:
            z = S @ v
            s_new = float(torch.dot(v, z))
            v = z / z.norm()
            err, sig, its = abs(s_new - sig), s_new, its + 1
        state = be.power_iteration(K, reg, v0.clone(), U=U, M=M, rho=rho, max_iters=max_iters, rtol=rtol)
        got, done, used = (float(x) for x in state)
        tol = 2e-5 if dt == torch.float32 else 1e-11
        assert abs(got - float(want)) <= tol * abs(float(want)), (precond, max_iters, got, float(want))
        if rtol > 1e-6:  # (at rtol 1e-7 the float32 stopping decision is within rounding of the threshold)
            assert int(used) == its, (precond, max_iters, used, its)
            assert bool(done) == (err <= rtol * sig)


@pytest.mark.parametrize("accel", [True, False])
@pytest.mark.parametrize("eta_on_device", [True, False])
def test_sap_update_matches_the_elementwise_formulas(be, prec, accel, eta_on_device):
    """rk_sap_update against the reference's update lines (solvers/sap.py:139-175) written with
    torch element-wise / index operations."""
    dt = DTYPES[prec]
    gen = torch.Generator().manual_seed(23)
    n, k, nb = 1000, 3, 77
    W, V, Y = (torch.randn(n, k, dtype=dt, generator=gen).to(dev()) for _ in range(3))
    blk = torch.randperm(n, generator=gen)[:nb].to(dev())
    d = torch.randn(nb, k, dtype=dt, generator=gen).to(dev())
    eta, beta, gamma, alpha = 0.37, 0.9, 3.0, 0.25
    W0, V0, Y0 = W.clone(), V.clone(), Y.clone()
    if accel:
        Wm = Y0.clone()
        Wm[blk] -= eta * d
        Vm = beta * V0 + (1 - beta) * Y0
        Vm[blk] -= (eta * gamma) * d
        want = (Wm, Vm, alpha * Vm + (1 - alpha) * Wm)
    else:
        Wm = W0.clone()
        Wm[blk] -= eta * d
        want = (Wm, V0, Y0)
    e = torch.tensor(eta, dtype=dt, device=dev()) if eta_on_device else eta
    assert be.sap_update(W, V if accel else None, Y if accel else None, blk, d, e, accel, beta, gamma, alpha)
    tol = dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-14, atol=1e-14)
    for got, ref in zip((W, V, Y), want):
        torch.testing.assert_close(got, ref, **tol)
    assert not be.sap_update(W.T, V, Y, blk, d, e, accel, beta, gamma, alpha)  # non-contiguous: declined


@pytest.mark.parametrize("dt,m,n,k", [
    (torch.float32, 33, 256, 10), (torch.float32, 1000, 260, 11), (torch.float32, 31, 5000, 16),
    (torch.float32, 257, 33000, 13), (torch.float32, 4096, 4096, 12), (torch.float32, 129, 64 * 30 + 4, 9),
    (torch.float32, 17, 64, 16), (torch.float32, 300, 68, 9),
    (torch.float64, 33, 256, 10), (torch.float64, 1000, 130, 11), (torch.float64, 31, 5000, 16),
    (torch.float64, 257, 33002, 13), (torch.float64, 2048, 4096, 9), (torch.float64, 65, 34, 12),
])
def test_matmat_skinny_matrix_core_kernel(be, monkeypatch, dt, m, n, k):
    """9 <= k <= 16, n a multiple of the 16-byte vector: the matrix-core kernel (operands straight from
    global memory, re-packed V) against float64 on the host and against the LDS-shared VALU kernel
    (RK_SKINNY_MFMA=0), with the fused alpha / beta / diag_shift epilogue and a padded leading
    dimension; ragged last steps, fewer rows than a wave owns, one step only."""
    gen = torch.Generator().manual_seed(m + n + k)
    Afull = torch.randn(m, n + 8, dtype=dt, generator=gen)
    A = Afull[:, :n]
    V = torch.randn(n, k, dtype=dt, generator=gen)
    C = torch.randn(m, k, dtype=dt, generator=gen)
    Ad = Afull.to(dev())[:, :n]  # lda = n + 8
    shift = 0.5 if m <= n else 0.0
    monkeypatch.setenv("RK_SKINNY_MFMA", "2")  # (by default only matrices of 16 MB and more take it)
    got = be.matmat(Ad, V.to(dev()), alpha=-1.5, beta=0.25, C=C.to(dev()), diag_shift=shift, row0=0)
    want = -1.5 * (A.double() @ V.double()) + 0.25 * C.double()
    if shift:
        want = want + shift * V[:m].double()
    assert rel_err(got.cpu(), want) < LIM[dt] * 5
    monkeypatch.setenv("RK_SKINNY_MFMA", "0")
    base = be.matmat(Ad, V.to(dev()), alpha=-1.5, beta=0.25, C=C.to(dev()), diag_shift=shift, row0=0)
    monkeypatch.setenv("RK_SKINNY_MFMA", "2")
    assert rel_err(got.cpu(), base.cpu().double()) < LIM[dt] * 5
    assert not torch.equal(got, base)  # two different kernels ran
    assert torch.equal(got, be.matmat(Ad, V.to(dev()), alpha=-1.5, beta=0.25, C=C.to(dev()), diag_shift=shift, row0=0))
    monkeypatch.delenv("RK_SKINNY_MFMA")
    auto = be.matmat(Ad, V.to(dev()), alpha=-1.5, beta=0.25, C=C.to(dev()), diag_shift=shift, row0=0)
    big = m * n * Ad.element_size() >= 16 << 20
    assert torch.equal(auto, got if big else base)  # by size: the matrix-core kernel from 16 MB on


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
@pytest.mark.parametrize("n,d", [(256, 8), (700, 64), (1000, 33), (2500, 50), (2049, 64)])
@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_kernel_matmat_symmetric_evaluation(be, monkeypatch, kind, n, d, k):
    """K(X, X) @ V matrix-free with the symmetric evaluation (only the tiles on and above the diagonal,
    each used for y[i] += P v[j] and y[j] += P^T v[i]) against the oracle and against the full
    evaluation (RK_KMAT_SYM=0); also through one shared index list (the block oracle K(X[blk], X[blk]))."""
    dt = torch.float32
    monkeypatch.setenv("RK_KMAT_SYM_MINN", "1")  # (the product defaults: n >= 65 536, k <= 2)
    monkeypatch.setenv("RK_KMAT_SYM_MAXK", "4")
    gen = torch.Generator().manual_seed(n + d + k)
    X = torch.randn(n, d, dtype=dt, generator=gen) / d**0.5
    V = torch.randn(n, k, dtype=dt, generator=gen)
    Xd, Vd = X.to(dev()), V.to(dev())
    gc = be.gram_center(Xd, Xd, 0.9)
    assert gc is not None
    want = ok.kernel_matmat(kind, X.double(), X.double(), 0.9, V.double(), 1.25)
    got = be.kernel_matmat(kind, Xd, Xd, 0.9, Vd, scale=1.25, gram_center=gc)
    monkeypatch.setenv("RK_KMAT_SYM", "0")
    full = be.kernel_matmat(kind, Xd, Xd, 0.9, Vd, scale=1.25, gram_center=gc)
    monkeypatch.delenv("RK_KMAT_SYM")
    assert rel_err(got.cpu(), want) < LIM[dt]
    assert rel_err(got.cpu(), full.cpu().double()) < LIM[dt]
    assert torch.equal(got, full) == (d <= 32)  # rows longer than 32 features: the symmetric evaluation ran
    assert torch.equal(got, be.kernel_matmat(kind, Xd, Xd, 0.9, Vd, scale=1.25, gram_center=gc))  # reproducible
    if n >= 700:
        blk = torch.randperm(n, generator=gen)[: n // 2 + 3].to(dev())
        Vb = Vd[: blk.numel()].contiguous()
        want_b = ok.kernel_matmat(kind, X.double(), X.double(), 0.9, V[: blk.numel()].double(), 1.25,
                                  idx1=blk.cpu(), idx2=blk.cpu())
        got_b = be.kernel_matmat(kind, Xd, Xd, 0.9, Vb, scale=1.25, idx1=blk, idx2=blk, gram_center=gc)
        assert rel_err(got_b.cpu(), want_b) < LIM[dt]


def test_matrix_free_operator_wide_product_uses_panels(be, monkeypatch, prec):
    """RK_KERNEL_MODE=matrix_free: a product with many columns (the Nystrom sketch) goes through
    temporary row-major panels + the matrix-core GEMM, a narrow one through the fused kernel; both
    equal the dense product.  A tiny panel budget forces several panels."""
    import rlaopt_amd.kernels.base as kb
    from rlaopt_amd.kernels import KernelConfig, RBFLinOp

    dt = DTYPES[prec]
    monkeypatch.setenv("RK_KERNEL_MODE", "matrix_free")
    monkeypatch.setattr(kb, "_FREE_PANEL_BYTES", 300 * 700 * 4)  # ~300 rows per row-major panel, two strip subsets
    gen = torch.Generator().manual_seed(4)
    n, d = 700, 9
    X = (torch.randn(n, d, dtype=dt, generator=gen) / d**0.5)
    Xd = X.to(dev())
    A = RBFLinOp(Xd, Xd, KernelConfig(lengthscale=1.1, const_scaling=0.7))
    assert A.storage_mode == "matrix_free"
    K = ok.kernel_matrix("rbf", X.double(), X.double(), 1.1, 0.7)
    for cols in (3, 130):
        M = torch.randn(n, cols, dtype=dt, generator=gen)
        got = A @ M.to(dev())
        assert rel_err(got.cpu(), K @ M.double()) < LIM[dt] * 5, cols
    if dt == torch.float32:
        # float32, smooth kernel, d <= 64: the wide product above evaluated the entries INSIDE the product
        # (rk_kernel_sketch_matmat).  Without it (RK_FREE_WIDE_FUSED=0), symmetric data go through temporary cyclic strip
        # subsets of the packed triangle + rk_symm_gemm (two subsets under this budget); the row-major panel route
        # gives the same product
        fused_calls = []
        real_f = be.kernel_sketch_matmat
        monkeypatch.setattr(be, "kernel_sketch_matmat", lambda *a, **k: fused_calls.append(1) or real_f(*a, **k))
        M = torch.randn(n, 130, dtype=dt, generator=gen)
        got_f = A @ M.to(dev())
        assert fused_calls == [1]
        assert rel_err(got_f.cpu(), K @ M.double()) < LIM[dt] * 5
        monkeypatch.setenv("RK_FREE_WIDE_FUSED", "0")
        calls = []
        real = be.symm_gemm
        monkeypatch.setattr(be, "symm_gemm", lambda *a, **k: calls.append(k.get("stride")) or real(*a, **k))
        M = torch.randn(n, 130, dtype=dt, generator=gen)
        got = A @ M.to(dev())
        assert len(calls) >= 2 and all(c == len(calls) for c in calls), calls
        monkeypatch.setenv("RK_FREE_WIDE_STRIPS", "0")
        panels = A @ M.to(dev())
        assert rel_err(got.cpu(), K @ M.double()) < LIM[dt] * 5
        assert rel_err(panels.cpu(), K @ M.double()) < LIM[dt] * 5


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("n1,n2,d,k,ard", [(700, 700, 9, 130, False), (777, 1301, 33, 40, True), (1500, 515, 64, 300, False),
                                           (129, 31, 5, 17, False), (2048, 2048, 64, 512, True)])
def test_kernel_sketch_matmat_evaluates_inside_the_product(be, kind, n1, n2, d, k, ard):
    """rk_kernel_sketch_matmat: alpha * scale * K(X1, X2) @ V with the entries evaluated tile by tile inside the wide
    product (X2 tile x X1 rows on the 16-bit matrix cores, the kernel function on the accumulators, split, straight
    into the operand image) - against the float64 oracle: square and rectangular point sets, sizes that are not
    multiples of the 32-point steps or the 128-row blocks, scalar and per-feature lengthscales, un-centred data, one
    and two column tiles, columns of very different magnitude, bitwise reproducibility."""
    dt = torch.float32
    g = torch.Generator().manual_seed(n1 + 7 * n2 + d)
    X1 = torch.randn(n1, d, dtype=dt, generator=g) / d**0.5 + 3.0   # (a common offset: the centre removes it)
    X2 = X1 if n1 == n2 else torch.randn(n2, d, dtype=dt, generator=g) / d**0.5 + 3.0
    ell = (0.8 + 0.4 * torch.rand(d, dtype=dt, generator=g)) if ard else 1.1
    V = torch.randn(n2, k, dtype=dt, generator=g) * torch.logspace(-4, 3, k, dtype=dt)
    X1d = X1.to(dev())
    X2d = X1d if n1 == n2 else X2.to(dev())
    elld = ell.to(dev()) if ard else ell
    gc = be.gram_center(X1d, X2d, elld)
    assert gc is not None
    want = -0.5 * (ok.kernel_matrix(kind, X1.double(), X2.double(), ell.double() if ard else ell, 0.7) @ V.double())
    got = be.kernel_sketch_matmat(kind, X1d, X2d, elld, V.to(dev()), scale=0.7, gram_center=gc, alpha=-0.5)
    again = be.kernel_sketch_matmat(kind, X1d, X2d, elld, V.to(dev()), scale=0.7, gram_center=gc, alpha=-0.5)
    assert torch.equal(got, again)
    err = (got.cpu().double() - want).norm(dim=0) / want.norm(dim=0)
    assert float(err.max()) < 1e-5, (float(err.max()), int(err.argmax()))
    with pytest.raises(ValueError):
        be.kernel_sketch_matmat("laplace", X1d, X2d, elld, V.to(dev()), gram_center=gc)
    with pytest.raises(ValueError):
        be.kernel_sketch_matmat(kind, X1d, X2d, elld, V.to(dev()), gram_center=None)
    # a V on the float16 grid (rk_kernel_sketch_matmat_rhs16: one piece, two products per pair): the general entry adds
    # exact zeros for the missing piece in the same order - equal bit for bit
    from rlaopt_amd.ops import f16_grid

    Vg = f16_grid(V / k**0.5).to(dev())
    one = be.kernel_sketch_matmat(kind, X1d, X2d, elld, Vg, scale=0.7, gram_center=gc, rhs_f16=True)
    assert torch.equal(one, be.kernel_sketch_matmat(kind, X1d, X2d, elld, Vg, scale=0.7, gram_center=gc))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 17, 1000, 4097, 500_000])
def test_sample_distinct_is_a_permutation_head(be, n):
    """rk_sample_distinct: k = n gives every index exactly once (a bijection of [0, n)); a shorter draw
    is the head of the longer one; the draw depends on the seed and on nothing else."""
    full = be.sample_distinct(n, n, dev(), seed=12345)
    assert full.dtype == torch.int64 and full.shape == (n,) and full.is_cuda
    assert torch.equal(torch.sort(full).values.cpu(), torch.arange(n))
    k = max(1, n // 100)
    head = be.sample_distinct(n, k, dev(), seed=12345)
    assert torch.equal(head, full[:k])
    assert torch.equal(head, be.sample_distinct(n, k, dev(), seed=12345))
    if n >= 1000:
        other = be.sample_distinct(n, n, dev(), seed=12346)
        assert not torch.equal(other, full)
        assert float((other == full).float().mean()) < 0.01  # unrelated permutations
        # not the identity and no drift: a permutation's values are spread evenly over its positions
        q = full[: n // 4].double()
        assert abs(float(q.mean()) / (n - 1) - 0.5) < 0.02
    with pytest.raises((RuntimeError, ValueError)):
        be.sample_distinct(n, n + 1, dev(), seed=1)


def test_sample_distinct_is_uniform_over_seeds(be):
    """Every index of [0, n) is equally likely to be in the drawn subset, and at each position of it
    (chi-square over 4 000 seeds, n = 23, k = 6: 22 degrees of freedom, 99.9 % quantile 48.3)."""
    n, k, seeds = 23, 6, 4000
    draws = torch.stack([be.sample_distinct(n, k, dev(), seed=1000 + 7 * s) for s in range(seeds)]).cpu()
    counts = torch.bincount(draws.reshape(-1), minlength=n).double()
    expect = seeds * k / n
    assert float(((counts - expect) ** 2 / expect).sum()) < 60.0
    first = torch.bincount(draws[:, 0], minlength=n).double()
    assert float(((first - seeds / n) ** 2 / (seeds / n)).sum()) < 60.0
    # pairs: position 0 and position 1 are not correlated beyond distinctness
    pair = torch.bincount(draws[:, 0] * n + draws[:, 1], minlength=n * n).double().view(n, n)
    assert float(pair.diagonal().sum()) == 0
    off = pair[~torch.eye(n, dtype=torch.bool)]
    e2 = seeds / (n * (n - 1))
    assert float(((off - e2) ** 2 / e2).sum()) < 506 + 5 * (2 * 506) ** 0.5  # 505 dof, mean + 5 sigma


def test_sap_block_draw_uses_the_device_permutation(be, monkeypatch):
    """SAP._get_blk on a GPU system: blk_sz distinct indices in one launch, the same block for the same
    torch seed; RK_SAP_DRAW=randperm restores the sort-based draw."""
    from rlaopt_amd.kernels import KernelConfig, RBFLinOp
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import IdentityConfig
    from rlaopt_amd.solvers.sap import SAP

    n = 3000
    g = torch.Generator().manual_seed(1)
    X = torch.randn(n, 4, generator=g).to(dev())
    B = torch.randn(n, 1, generator=g).to(dev())
    A = RBFLinOp(X, X, KernelConfig(lengthscale=1.0))
    system = LinSys(A, B, reg=1.0, A_row_oracle=A.row_oracle, A_blk_oracle=A.blk_oracle)
    s = SAP(system, torch.zeros_like(B), IdentityConfig(), dev(), 300, False, None, 5)
    torch.manual_seed(4)
    b1 = s._get_blk()
    torch.manual_seed(4)
    b2 = s._get_blk()
    assert b1.is_cuda and b1.dtype == torch.int64 and b1.shape == (300,) and torch.equal(b1, b2)
    assert torch.unique(b1).numel() == 300 and int(b1.min()) >= 0 and int(b1.max()) < n
    monkeypatch.setenv("RK_SAP_DRAW", "randperm")
    torch.manual_seed(4)
    b3 = s._get_blk()
    assert torch.unique(b3).numel() == 300 and not torch.equal(b1, b3)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,k", [(1, 1), (512, 1), (513, 2), (5000, 3)])
def test_lsqr_tail_matches_the_separate_updates(be, dt, n, k):
    """rk_lsqr_tail (v, alpha, plane rotation, y and w updates of an LSQR step in one launch) against the
    same recurrences written out in float64 on the host; views with a padded leading dimension."""
    g = torch.Generator().manual_seed(n + k)
    mk = lambda: torch.randn(n, k + 2, dtype=dt, generator=g)  # noqa: E731
    G0, V0, W0, Y0 = mk(), mk(), mk(), mk()
    beta = torch.rand(k, dtype=torch.float64, generator=g) + 0.5
    rhobar = torch.randn(k, dtype=torch.float64, generator=g)
    phibar = torch.randn(k, dtype=torch.float64, generator=g)
    Gd, Vd, Wd, Yd = (t.to(dev())[:, :k] for t in (G0, V0, W0, Y0))
    rb, pb = rhobar.to(dev()), phibar.to(dev())
    alpha = be.lsqr_tail(Gd, Vd, Wd, Yd, beta.to(dev()), rb, pb)
    assert alpha is not None and alpha.dtype == torch.float64 and alpha.shape == (k,)
    G, V, W, Y = (t[:, :k].double() for t in (G0, V0, W0, Y0))
    v = (G - beta * V).to(dt).double()
    a = torch.linalg.norm(v, dim=0)
    v = (v / a).to(dt).double()
    rho = torch.sqrt(rhobar**2 + beta**2)
    cs, sn = rhobar / rho, beta / rho
    theta, phi = sn * a, cs * phibar
    tol = dict(rtol=2e-6, atol=2e-6) if dt == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(alpha.cpu(), a, rtol=tol["rtol"], atol=0)
    torch.testing.assert_close(Vd.cpu().double(), v, **tol)
    torch.testing.assert_close(Yd.cpu().double(), Y + (phi / rho) * W, **tol)
    torch.testing.assert_close(Wd.cpu().double(), v - (theta / rho) * W, **tol)
    torch.testing.assert_close(rb.cpu(), -cs * a, rtol=tol["rtol"], atol=0)
    torch.testing.assert_close(pb.cpu(), sn * phibar, rtol=tol["rtol"], atol=0)
    if n > 1 and k > 1:  # column-major G: declined
        assert be.lsqr_tail(Gd.T.contiguous().T, Vd, Wd, Yd, beta.to(dev()), rb, pb) is None


@pytest.mark.parametrize("kind,d", [("rbf", 50), ("matern32", 64), ("rbf", 100)])
def test_kernel_matmat_keeps_the_x2_image_between_calls(be, monkeypatch, kind, d):
    """rk_kernel_matmat_cached: the converted image of the X2 side is written on the first call and reused
    afterwards (new X1 rows, new V each time) - bitwise the results of the plain entry point; other
    operands, another formulation or another stream convert again / keep no image."""
    g = torch.Generator().manual_seed(d)
    n2, k = 3000, 10
    X = (torch.randn(n2, d, generator=g) / d**0.5).to(dev())
    gc = be.gram_center(X, X, 1.0)
    assert gc is not None
    cache = {}

    def slot(c, stream=None):
        stream = torch.cuda.current_stream() if stream is None else stream
        return c.get("streams", {}).get(stream.cuda_stream, {})

    for step in range(3):
        blk = torch.randperm(n2, generator=g)[:200].to(dev())
        V = torch.randn(n2, k, generator=g).to(dev())
        A1 = X[blk].contiguous()
        plain = be.kernel_matmat(kind, A1, X, 1.0, V, gram_center=gc)
        kept = be.kernel_matmat(kind, A1, X, 1.0, V, gram_center=gc, x2_cache=cache)
        assert torch.equal(plain, kept)
        assert slot(cache)["ready"] is True and slot(cache)["buf"].numel() > 0
        if step == 1:
            buf_ptr = slot(cache)["buf"].data_ptr()
    assert slot(cache)["buf"].data_ptr() == buf_ptr  # (one buffer per stream for the lifetime of the cache)
    # the points change IN PLACE (same address): the image is converted again
    Xm = X.clone()
    cm = {}
    V = torch.randn(n2, k, generator=g).to(dev())
    be.kernel_matmat(kind, X[:100].contiguous(), Xm, 1.0, V, gram_center=gc, x2_cache=cm)
    assert slot(cm)["ready"] is True
    Xm.mul_(0.75)
    got = be.kernel_matmat(kind, X[:100].contiguous(), Xm, 1.0, V, gram_center=gc, x2_cache=cm)
    assert torch.equal(got, be.kernel_matmat(kind, X[:100].contiguous(), Xm, 1.0, V, gram_center=gc))
    # other operands: converted again, correct again
    X2 = (X * 0.5).contiguous()
    got = be.kernel_matmat(kind, X[:100].contiguous(), X2, 1.0, V, gram_center=gc, x2_cache=cache)
    assert torch.equal(got, be.kernel_matmat(kind, X[:100].contiguous(), X2, 1.0, V, gram_center=gc))
    # another stream: not ordered behind the write - it converts into a buffer of its own
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got2 = be.kernel_matmat(kind, X[:100].contiguous(), X2, 1.0, V, gram_center=gc, x2_cache=cache)
    side.synchronize()
    assert torch.equal(got2, got)
    assert slot(cache, side)["buf"].data_ptr() != slot(cache)["buf"].data_ptr()
    # a formulation without images: nothing is kept
    monkeypatch.setenv("RK_KMAT_BF16X3", "0")
    c2 = {}
    be.kernel_matmat(kind, X[:100].contiguous(), X2, 1.0, V, gram_center=gc, x2_cache=c2)
    assert slot(c2).get("ready") is False
    monkeypatch.delenv("RK_KMAT_BF16X3")
    got3 = be.kernel_matmat(kind, X[:100].contiguous(), X2, 1.0, V, gram_center=gc, x2_cache=c2)
    assert slot(c2)["ready"] is True and torch.equal(got3, got)
    # symmetric operands and float64 keep no image
    c3 = {}
    be.kernel_matmat(kind, X, X, 1.0, V[:, :2].contiguous(), gram_center=gc, x2_cache=c3)
    assert "streams" not in c3


@pytest.mark.parametrize("n,k,first,stride", [(1000, 40, 0, 1), (2304, 100, 0, 1), (3000, 256, 0, 1), (5000, 512, 0, 1),
                                             (4100, 130, 1, 3), (2600, 64, 0, 2), (2600, 64, 1, 2), (700, 17, 0, 1),
                                             (1500, 600, 0, 1), (1100, 1024, 0, 1), (2100, 700, 2, 5), (90, 33, 0, 1)])
@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
def test_symm_gemm_wide_product_from_the_packed_strips(be, monkeypatch, n, k, first, stride, split):
    """rk_symm_gemm: Y = K V for many columns straight from the float32 packed triangle (both piece formats),
    against the float64 product with the dense kernel matrix; strips dealt cyclically: the partials of the
    ranks add up to the product; const_scaling, a V with a wide dynamic range (per-column scales 1e-6 .. 1e3:
    the power-of-two scaling must keep the float16 pieces exact) and bitwise reproducibility."""
    if split == "bf16x3":
        monkeypatch.setenv("RK_SYMGEMM_SPLIT", "bf16x3")
    dt = torch.float32
    d = 9
    g = torch.Generator().manual_seed(n + k)
    X = torch.randn(n, d, dtype=dt, generator=g) / d**0.5
    scale = 3.5
    V = torch.randn(n, k, dtype=dt, generator=g) * torch.logspace(-6, 3, k, dtype=dt)
    Kd = ok.kernel_matrix("rbf", X.double(), X.double(), 1.0, scale)
    want = Kd @ V.double()
    Xd, Vd = X.to(dev()), V.to(dev())
    if stride == 1:
        Kp = be.sympack_materialize("rbf", Xd, 1.0, scale=scale)
        got = be.symm_gemm(Kp, n, Vd, kmax=scale)
        assert torch.equal(got, be.symm_gemm(Kp, n, Vd, kmax=scale))  # reproducible bit for bit
        got = got.cpu().double()
    else:
        parts = []
        for f in range(stride):
            Kp = be.sympack_materialize("rbf", Xd, 1.0, scale=scale, first=f, stride=stride)
            parts.append(be.symm_gemm(Kp, n, Vd, kmax=scale, first=f, stride=stride).cpu().double())
        got = sum(parts)
        if first:  # (the parametrisation only varies which rank's partial is also checked to be non-trivial)
            assert float(parts[first].abs().max()) > 0
    # column-wise relative error: the level of a float32 product (the float32 matrix itself carries 6e-8 per
    # entry; the matrix cores' float32 accumulation truncates: ~1e-8 per accumulation step of a chain)
    err = (got - want).norm(dim=0) / want.norm(dim=0)
    assert float(err.max()) < 5e-6, (float(err.max()), int(err.argmax()))
    # alpha, and a strided V (columns of a wider array)
    if stride == 1:
        wide = torch.randn(n, k + 3, dtype=dt, generator=g).to(dev())
        got2 = be.symm_gemm(Kp, n, wide[:, 2 : 2 + k], kmax=scale, alpha=-0.5).cpu().double()
        want2 = -0.5 * (Kd @ wide[:, 2 : 2 + k].cpu().double())
        assert float((got2 - want2).norm() / want2.norm()) < 5e-6


@pytest.mark.parametrize("n,k,first,stride", [(1000, 40, 0, 1), (3000, 256, 0, 1), (5000, 512, 0, 1), (4100, 130, 1, 3),
                                             (700, 17, 0, 1), (1100, 1024, 0, 1), (2100, 700, 2, 5), (90, 33, 0, 1)])
@pytest.mark.parametrize("impl", ["ph", "lockstep"])
def test_symm_gemm_with_a_block_on_the_float16_grid(be, monkeypatch, n, k, first, stride, impl):
    """rk_symm_gemm_rhs16: a V on the float16 grid has ONE piece - two products per pair.  On such a V the general
    entry adds exact zeros for the missing piece, in the same order: the two results are equal bit for bit; the
    float64 product bounds both.  A V off the grid is rounded to it (per column, after the kernel's scaling).  (RK_SYMGEMM_IMPL only selects the kernel of the
    general entry: both of them against the one-piece kernel.)"""
    from rlaopt_amd.ops import f16_grid, on_f16_grid

    monkeypatch.setenv("RK_SYMGEMM_IMPL", impl)
    dt = torch.float32
    g = torch.Generator().manual_seed(7 * n + k)
    X = torch.randn(n, 6, dtype=dt, generator=g) / 6**0.5
    scale = 0.75
    raw = torch.randn(n, k, dtype=dt, generator=g) * torch.logspace(-5, 2, k, dtype=dt) / k**0.5
    V = f16_grid(raw)
    assert on_f16_grid(V) and not on_f16_grid(raw) and torch.equal(V, V.half().float())
    Kd = ok.kernel_matrix("rbf", X.double(), X.double(), 1.0, scale)
    Xd = X.to(dev())
    Kp = be.sympack_materialize("rbf", Xd, 1.0, scale=scale, first=first, stride=stride)
    one = be.symm_gemm(Kp, n, V.to(dev()), kmax=scale, first=first, stride=stride, rhs_f16=True)
    two = be.symm_gemm(Kp, n, V.to(dev()), kmax=scale, first=first, stride=stride)
    assert torch.equal(one, two)
    # off the grid: rounded to float16 AFTER the kernel's power-of-two scaling of the column (largest entry below 2^14:
    # a column of small numbers keeps 11 bits where a plain .half() would make subnormals of them)
    ex = torch.frexp(raw.abs().amax(dim=0))[1]
    up = torch.ldexp(torch.ones(k), 14 - ex)
    rounded = (raw * up).half().float() / up
    got = be.symm_gemm(Kp, n, raw.to(dev()), kmax=scale, first=first, stride=stride, rhs_f16=True)
    assert torch.equal(got, be.symm_gemm(Kp, n, rounded.to(dev()), kmax=scale, first=first, stride=stride))
    if stride == 1:
        want = Kd @ V.double()
        err = (one.cpu().double() - want).norm(dim=0) / want.norm(dim=0)
        assert float(err.max()) < 5e-6, (float(err.max()), int(err.argmax()))


@pytest.mark.parametrize("name,kind", [("RBF", "rbf"), ("Laplace", "laplace"), ("Matern12", "matern12"),
                                       ("Matern32", "matern32"), ("Matern52", "matern52")])
def test_integration_md_stub_runs_against_the_oracle(be, prec, name, kind):
    """INTEGRATION.md section B, executed verbatim: the ctypes stub a maintainer would add to the reference
    (rlaopt/kernels/_rk.py) drives librk_hip.so directly - plain products, 1-D x, row and block oracles with
    in-kernel index gathers, per-feature (ARD) lengthscales, const_scaling, the matrix-core opt-in - and
    agrees with the oracle at the reference's tolerances."""
    from test_abi import load_integration_stub

    km = load_integration_stub()["kernel_matmat"]
    dt = DTYPES[prec]
    g = torch.Generator().manual_seed(11)
    n1, n2, d, k = 333, 1201, 7, 5
    A1 = (torch.randn(n1, d, dtype=dt, generator=g) / d**0.5).to(dev())
    A2 = (torch.randn(n2, d, dtype=dt, generator=g) / d**0.5).to(dev())
    x = torch.randn(n2, k, dtype=dt, generator=g).to(dev())
    ard = (torch.rand(d, dtype=dt, generator=g) + 0.5).to(dev())
    lim = LIM[dt] * 5
    for ell in (0.8, ard):
        want = ok.kernel_matmat(kind, A1.cpu().double(), A2.cpu().double(), ell.cpu().double() if isinstance(ell, torch.Tensor) else ell,
                                x.cpu().double(), 1.7)
        assert rel_err(km(name, A1, A2, ell, 1.7, x).cpu(), want) < lim
        assert rel_err(km(name, A1, A2, ell, 1.7, x[:, 0].contiguous()).cpu(), want[:, 0]) < lim
    # row oracle K(A1[blk], A2) and block oracle K(A2[blk], A2[blk]): device int64 index lists
    blk = torch.randperm(n1, generator=g)[:50].to(dev())
    want = ok.kernel_matmat(kind, A1.cpu().double(), A2.cpu().double(), 1.0, x.cpu().double(), 1.0, idx1=blk.cpu())
    assert rel_err(km(name, A1, A2, 1.0, 1.0, x, idx1=blk).cpu(), want) < lim
    blk2 = torch.randperm(n2, generator=g)[:64].to(dev())
    w = torch.randn(64, 2, dtype=dt, generator=g).to(dev())
    want = ok.kernel_matmat(kind, A2.cpu().double(), A2.cpu().double(), ard.cpu().double(), w.cpu().double(), 1.0,
                            idx1=blk2.cpu(), idx2=blk2.cpu())
    assert rel_err(km(name, A2, A2, ard, 1.0, w, idx1=blk2, idx2=blk2).cpu(), want) < lim
    # the transpose product of the table: K(A1, A2)^T u = K(A2, A1) u
    u = torch.randn(n1, 3, dtype=dt, generator=g).to(dev())
    want = ok.kernel_rmatmat(kind, A1.cpu().double(), A2.cpu().double(), 1.0, u.cpu().double(), 1.0)
    assert rel_err(km(name, A2, A1, 1.0, 1.0, u).cpu(), want) < lim
    # opting in to the matrix-core formulation with the operator's centre (float32, smooth kernels)
    if dt == torch.float32 and kind in ("rbf", "matern32", "matern52"):
        gc = be.gram_center(A1, A2, 1.0)
        assert gc is not None
        want = ok.kernel_matmat(kind, A1.cpu().double(), A2.cpu().double(), 1.0, x.cpu().double(), 1.0)
        assert rel_err(km(name, A1, A2, 1.0, 1.0, x, gram_center=gc).cpu(), want) < lim
    # errors surface as the exceptions the stub documents
    with pytest.raises(ValueError):  # more than 16 columns per call: rejected by the C side, status -1
        km(name, A1, A2, 1.0, 1.0, torch.zeros(n2, 17, dtype=dt, device=dev()))


@pytest.mark.parametrize("n,k,first,stride", [(300, 5, 0, 1), (1000, 16, 0, 1), (2304, 7, 0, 1), (3000, 10, 0, 1),
                                             (5000, 16, 0, 1), (4100, 9, 1, 3), (2600, 12, 0, 2), (1500, 1, 0, 1),
                                             (33, 6, 0, 1), (1025, 4, 0, 1), (2049, 3, 0, 1), (6200, 11, 0, 1),
                                             (9000, 16, 2, 5), (7000, 8, 0, 8), (1, 1, 0, 1), (31, 3, 0, 1), (257, 2, 0, 1),
                                             (1024, 16, 0, 1), (1280, 1, 1, 2), (4097, 4, 3, 4)])
def test_symm_matmat_f16x2_matrix_core_strip_product(be, n, k, first, stride):
    """rk_symm_matmat_f16x2: the packed symmetric product for up to 16 right-hand sides on the 16-bit matrix
    cores (two float16 pieces per factor, row reads and transposing reads of one LDS image) against the float64
    product: per-column error at float32 level for columns of very different magnitude, alpha and the
    "+ diag_shift V" term, cyclic strip distributions (partials add up), bitwise reproducibility, and agreement
    with the float32 kernels it replaces."""
    dt = torch.float32
    d = 6
    g = torch.Generator().manual_seed(3 * n + k)
    X = torch.randn(n, d, dtype=dt, generator=g) / d**0.5
    scale = 0.75
    V = torch.randn(n, k, dtype=dt, generator=g) * torch.logspace(-5, 2, k, dtype=dt)
    Kd = ok.kernel_matrix("matern32", X.double(), X.double(), 0.9, scale)
    Xd, Vd = X.to(dev()), V.to(dev())
    want = -1.5 * (Kd @ V.double()) + 0.25 * V.double()
    parts = []
    for f in range(stride):
        Kp = be.sympack_materialize("matern32", Xd, 0.9, scale=scale, first=f, stride=stride)
        import os as _os
        _os.environ["RK_SYMM_F16X2_MIN"] = "1"  # (one or two columns would take the vector-unit block kernel)
        got = be.symm_matmat(Kp, n, Vd, alpha=-1.5, diag_shift=0.25 if f == 0 else 0.0, first=f, stride=stride,
                             kmax=scale)
        again = be.symm_matmat(Kp, n, Vd, alpha=-1.5, diag_shift=0.25 if f == 0 else 0.0, first=f, stride=stride,
                               kmax=scale)
        assert torch.equal(got, again)  # fixed-order tables: reproducible bit for bit
        # the round-3 strip form of the same product (RK_SYMM_H2_IMPL=strip) stays available as A/B partner
        _os.environ["RK_SYMM_H2_IMPL"] = "strip"
        try:
            strip = be.symm_matmat(Kp, n, Vd, alpha=-1.5, diag_shift=0.25 if f == 0 else 0.0, first=f, stride=stride,
                                   kmax=scale)
        finally:
            del _os.environ["RK_SYMM_H2_IMPL"]
            del _os.environ["RK_SYMM_F16X2_MIN"]
        if stride == 1:
            assert float(((got - strip).double().norm(dim=0) / strip.double().norm(dim=0)).max()) < 3e-6
        parts.append(got.cpu().double())
        if stride == 1:  # the float32 kernels it replaces (register kernel / float32 matrix cores)
            old = be.symm_matmat(Kp, n, Vd, alpha=-1.5, diag_shift=0.25)
            assert float(((got - old).double().norm(dim=0) / old.double().norm(dim=0)).max()) < 5e-6
    got = sum(parts)
    err = (got - want).norm(dim=0) / want.norm(dim=0)
    assert float(err.max()) < 3e-6, (float(err.max()), int(err.argmax()))
    if first:
        assert float(parts[first].abs().max()) > 0


@pytest.mark.parametrize("n,r,k", [(1000, 32, 1), (5000, 100, 10), (777, 17, 16), (64, 5, 3)])
def test_nystrom_inv_apply_composite(be, prec, n, r, k):
    """rk_nystrom_inv_apply: (X - U (M (U^T X))) / rho in three launches, with the dense r x r factor of the
    float32 branch and the diagonal factor of the float64 branch (nystrom.py:112-132), 1-D and strided inputs."""
    dt = DTYPES[prec]
    g = torch.Generator().manual_seed(n + r)
    U = torch.linalg.qr(torch.randn(n, r, dtype=torch.float64, generator=g))[0].to(dt)
    X = torch.randn(n, k, dtype=dt, generator=g)
    A = torch.randn(r, r, dtype=torch.float64, generator=g)
    M = A @ A.T / r + torch.eye(r, dtype=torch.float64)
    S = torch.rand(r, dtype=torch.float64, generator=g) + 0.1
    rho = 0.37
    Ud, Xd = U.to(dev()), X.to(dev())
    lim = LIM[dt] * 5
    want = (X.double() - U.double() @ (M @ (U.double().T @ X.double()))) / rho
    assert rel_err(be.nystrom_inv_apply(Ud, M.to(dev()), rho, Xd).cpu(), want) < lim
    Md = S / (S + rho)
    want = (X.double() - U.double() @ (Md[:, None] * (U.double().T @ X.double()))) / rho
    assert rel_err(be.nystrom_inv_apply(Ud, Md.to(dev()), rho, Xd).cpu(), want) < lim
    got = be.nystrom_inv_apply(Ud, Md.to(dev()), rho, Xd[:, 0].contiguous())
    assert got.shape == (n,) and rel_err(got.cpu(), want[:, 0]) < lim
    wide = torch.randn(n, k + 2, dtype=dt, generator=g).to(dev())
    want = (wide[:, 1 : 1 + k].cpu().double() - U.double() @ (M @ (U.double().T @ wide[:, 1 : 1 + k].cpu().double()))) / rho
    assert rel_err(be.nystrom_inv_apply(Ud, M.to(dev()), rho, wide[:, 1 : 1 + k]).cpu(), want) < lim
