"""Pin the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.

Tolerances are the reference suite's own (fp32 1e-4, fp64 1e-8) except where the oracle
follows the reference call-for-call on the CPU, where agreement is required to be much
tighter (noted per test).
"""

import pytest
import torch

from conftest import DTYPES, TOL, load_golden
from oracle import kernels as ok
from oracle import linsys as ol
from oracle import nystrom as on
from oracle import pcg as opcg
from oracle import sap as osap
from oracle import sketches as osk
from oracle import skpre as oskpre


# ---------------------------------------------------------------- kernels (rows a1-a4)
@pytest.mark.parametrize("kind", ok.KINDS)
@pytest.mark.parametrize("shape", ["small", "ragged"])
@pytest.mark.parametrize("ell", ["scalar", "ard"])
def test_kernel_entries_match_reference_closed_form(prec, kind, shape, ell):
    g = load_golden(f"kernels_{prec}")
    X1, X2 = g[f"{shape}_X1"], g[f"{shape}_X2"]
    lengthscale = 1.0 if ell == "scalar" else g[f"{shape}_ard"]
    K = ok.kernel_matrix(kind, X1, X2, lengthscale, const_scaling=2.0)
    torch.testing.assert_close(K, g[f"{shape}_{kind}_{ell}"], **TOL[DTYPES[prec]])


@pytest.mark.parametrize("kind", ok.KINDS)
def test_kernel_products_and_oracles(prec, kind):
    """K@v, K@M, K.T@v, row oracle, block oracle against the golden dense K
    (the checks of tests/kernels/test_standard.py:186-326)."""
    g = load_golden(f"kernels_{prec}")
    dt = DTYPES[prec]
    X1, X2, ard = g["ragged_X1"], g["ragged_X2"], g["ragged_ard"]
    K = g[f"ragged_{kind}_ard"]
    gen = torch.Generator().manual_seed(1)
    v = torch.randn(X2.shape[0], dtype=dt, generator=gen)
    M = torch.randn(X2.shape[0], 3, dtype=dt, generator=gen)
    u = torch.randn(X1.shape[0], 2, dtype=dt, generator=gen)
    blk = torch.tensor([5, 0, 17, 9])
    tol = TOL[dt]
    torch.testing.assert_close(ok.kernel_matmat(kind, X1, X2, ard, v, 2.0), K @ v, **tol)
    torch.testing.assert_close(ok.kernel_matmat(kind, X1, X2, ard, M, 2.0), K @ M, **tol)
    torch.testing.assert_close(ok.kernel_rmatmat(kind, X1, X2, ard, u, 2.0), K.T @ u, **tol)
    torch.testing.assert_close(
        ok.kernel_matmat(kind, X1, X2, ard, M, 2.0, idx1=blk), K[blk] @ M, **tol
    )
    # block oracle needs a square index set valid for both inputs
    Ksq = ok.kernel_matrix(kind, X1, X1, ard, 2.0)
    w = torch.randn(4, 2, dtype=dt, generator=gen)
    torch.testing.assert_close(
        ok.kernel_matmat(kind, X1, X1, ard, w, 2.0, idx1=blk, idx2=blk), Ksq[blk][:, blk] @ w, **tol
    )
    # distributed forms reduce to the same products for any shard count
    for parts in (1, 2, 3, 5):
        torch.testing.assert_close(ok.distributed_matmat(kind, X1, X2, ard, M, parts, 2.0), K @ M, **tol)
        torch.testing.assert_close(ok.distributed_rmatmat(kind, X1, X2, ard, u, parts, 2.0), K.T @ u, **tol)
        torch.testing.assert_close(
            ok.distributed_row_oracle_matmat(kind, X1, X2, ard, blk, M, parts, 2.0), K[blk] @ M, **tol
        )
        torch.testing.assert_close(
            ok.distributed_blk_oracle_matmat(kind, X1, X1, ard, blk, w, parts, 2.0),
            Ksq[blk][:, blk] @ w, **tol,
        )


def test_row_chunks_follow_torch_chunk():
    for n in (1, 5, 7, 16, 25000, 200000):
        for p in (1, 2, 3, 4, 8):
            want = [(int(c[0]), int(c[-1]) + 1) for c in torch.chunk(torch.arange(n), p)]
            assert ok.row_chunks(n, p) == want
    assert len(ok.row_chunks(5, 4)) == 3  # fewer chunks than devices


# ------------------------------------------------------------------- sketches (row a7)
@pytest.mark.parametrize("name", ["gauss", "ortho", "sparse"])
@pytest.mark.parametrize("mode", ["left", "right"])
def test_sketch_matrices_bit_identical_to_reference(prec, name, mode):
    g = load_golden("sketches")
    torch.manual_seed(7)
    Om = osk.get_sketch_matrix(name, mode, 12, 40, DTYPES[prec])
    assert Om.shape == ((12, 40) if mode == "left" else (40, 12))
    torch.testing.assert_close(Om.contiguous(), g[f"{name}_{mode}_{prec}"], rtol=0, atol=0)


# ------------------------------------------------------------------- Nystrom (a8, a9)
@pytest.mark.parametrize("case", ["a50", "a300"])
@pytest.mark.parametrize("sketch", ["gauss", "ortho", "sparse"])
@pytest.mark.parametrize("mode", ["adaptive", "non_adaptive"])
def test_nystrom_matches_reference(prec, case, sketch, mode):
    g = load_golden(f"nystrom_{prec}")
    dt = DTYPES[prec]
    A, x = g[f"{case}_A"], g[f"{case}_x"]
    key = f"{case}_{sketch}_{mode}"
    rho0, base = (1.0, 0.5) if case == "a50" else (1e-6, 1e-6)
    st = on.build(A, g[f"{key}_Omega"], rho0)
    on.update_damping(st, base, adaptive=(mode == "adaptive"))
    tol = TOL[dt]
    torch.testing.assert_close(st.S, g[f"{key}_S"], **tol)
    assert abs(st.rho - float(g[f"{key}_rho"])) <= tol["atol"] + tol["rtol"] * abs(float(g[f"{key}_rho"]))
    # U is defined up to column signs / rotations inside clusters: compare the operators
    torch.testing.assert_close(on.matmul(st, x), g[f"{key}_Px"], **tol)
    ref = g[f"{key}_Pinvx"]
    got = on.inverse_matmul(st, x)
    scale = ref.abs().max()
    assert (got - ref).abs().max() <= tol["rtol"] * scale * 10 + tol["atol"]
    eye = torch.eye(st.U.shape[1], dtype=dt)
    torch.testing.assert_close(st.U.T @ st.U, eye, **tol)


# ---------------------------------------------------------------------- SkPre (a10)
def test_skpre_matches_reference(prec):
    g = load_golden(f"skpre_{prec}")
    dt = DTYPES[prec]
    L = oskpre.build(g["A"], g["Omega"], 1e-3)
    tol = TOL[dt] if dt == torch.float64 else dict(rtol=1e-2, atol=1e-4)  # test_skpre.py:65-72
    torch.testing.assert_close(L, g["L"], **tol)
    torch.testing.assert_close(oskpre.matmul(L, g["x"]), g["Px"], **tol)
    torch.testing.assert_close(oskpre.inverse_matmul(L, g["x"]), g["Pinvx"], **tol)


# ------------------------------------------------------------- PCG + LinSys (a11, a13)
@pytest.mark.parametrize("case", ["dense", "krr"])
@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("sketch", ["gauss", "ortho"])
def test_pcg_history_matches_reference(prec, case, k, sketch):
    g = load_golden(f"pcg_{prec}")
    dt = DTYPES[prec]
    key = f"{case}_k{k}_{sketch}"
    if case == "dense":
        A = g["dense_A"]
    else:
        A = ok.kernel_matrix("rbf", g["krr_X"], g["krr_X"], 1.0)
    B, reg, rtol = g[f"{key}_B"], float(g[f"{key}_reg"]), float(g[f"{key}_rtol"])
    st = on.build(A, g[f"{key}_Omega"], reg)
    on.update_damping(st, reg, adaptive=True)
    apply_A = lambda X: A @ X  # noqa: E731
    solver = opcg.BlockPCG(apply_A, lambda R: on.inverse_matmul(st, R), B, reg, torch.zeros_like(B))
    W, log = ol.run(solver, apply_A, B, reg, 300, 0.0, rtol, callback_freq=5)
    its = sorted(log)
    assert its == g[f"{key}_iters"].tolist()  # same iteration count, same log cadence
    rel = torch.stack([log[i]["rel_res"] for i in its])
    want = g[f"{key}_rel_res"]
    # residual histories: compare in log space where they are above the noise floor
    floor = 1e-10 if dt == torch.float64 else 1e-5
    sel = want > floor
    assert torch.allclose(rel[sel].log10(), want[sel].log10(), atol=0.05)
    torch.testing.assert_close(W, g[f"{key}_W"], **(TOL[dt] if dt == torch.float64 else dict(rtol=1e-3, atol=1e-3)))


# ----------------------------------------------------------------------- SAP (a12)
@pytest.mark.parametrize("pname", ["nystrom", "identity"])
@pytest.mark.parametrize("accel", [True, False])
def test_sap_trajectory_matches_reference(prec, pname, accel):
    g = load_golden(f"sap_{prec}")
    dt = DTYPES[prec]
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        key = f"{pname}_{'accel' if accel else 'plain'}"
        X, B, reg = g["X"], g["B"], float(g["reg"])
        K = ok.kernel_matrix("rbf", X, X, 1.0)
        n, k = B.shape
        precond = ("identity",) if pname == "identity" else ("nystrom", 16, reg, "gauss", True)
        solver = osap.SAP(
            n, lambda blk: K[blk], lambda blk: K[blk][:, blk], B, reg, torch.zeros_like(B),
            precond, blk_sz=48, accel=accel, mu=reg, nu=float(g["nu"]), power_iters=10,
        )
        torch.manual_seed(71)  # same seed as the generator script: same draws on CPU
        W, log = ol.run(solver, lambda V: K @ V, B, reg, 40, 0.0, 1e-12, callback_freq=10)
        assert sorted(log) == g[f"{key}_iters"].tolist()
        tol = TOL[dt] if dt == torch.float64 else dict(rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(W, g[f"{key}_W"], **tol)
        rel = torch.stack([log[i]["rel_res"] for i in sorted(log)])
        torch.testing.assert_close(rel, g[f"{key}_rel_res"], **tol)
    finally:
        torch.set_default_dtype(old)


# ---- the C restatement (oracle/rk_oracle.c): the checker of every full-size GPU assertion and the CPU baseline ----
@pytest.mark.parametrize("kind", ok.KINDS)
@pytest.mark.parametrize("shape", ["small", "ragged"])
@pytest.mark.parametrize("ell", ["scalar", "ard"])
def test_c_oracle_entries_match_reference_closed_form(prec, kind, shape, ell):
    """``cbind.kernel_matmat`` with V = I returns the kernel matrix itself: every entry against the
    golden matrices generated by the reference test-suite's closed forms (tests/kernels/utils.py:4-60),
    5 kinds x scalar / per-feature lengthscale x float32 / float64, reference tolerances."""
    from oracle import cbind

    g = load_golden(f"kernels_{prec}")
    X1, X2 = g[f"{shape}_X1"], g[f"{shape}_X2"]
    lengthscale = 1.0 if ell == "scalar" else g[f"{shape}_ard"]
    eye = torch.eye(X2.shape[0], dtype=DTYPES[prec])
    K = cbind.kernel_matmat(kind, X1, X2, lengthscale, eye, 2.0)
    torch.testing.assert_close(K, g[f"{shape}_{kind}_{ell}"], **TOL[DTYPES[prec]])
    torch.testing.assert_close(cbind.kernel_matrix(kind, X1, X2, lengthscale, 2.0), g[f"{shape}_{kind}_{ell}"],
                               **TOL[DTYPES[prec]])


@pytest.mark.parametrize("kind", ok.KINDS)
def test_c_oracle_products_match_python_oracle(prec, kind):
    """Products (vector and block right-hand sides, ragged shapes that do not divide the OpenMP chunking)
    against the golden dense K and against ``oracle/kernels.py`` on a larger seeded problem."""
    from oracle import cbind

    g = load_golden(f"kernels_{prec}")
    dt = DTYPES[prec]
    tol = TOL[dt]
    X1, X2, ard = g["ragged_X1"], g["ragged_X2"], g["ragged_ard"]
    K = g[f"ragged_{kind}_ard"]
    gen = torch.Generator().manual_seed(3)
    v = torch.randn(X2.shape[0], dtype=dt, generator=gen)
    M = torch.randn(X2.shape[0], 3, dtype=dt, generator=gen)
    got = cbind.kernel_matmat(kind, X1, X2, ard, v, 2.0)
    assert got.shape == (X1.shape[0],)
    torch.testing.assert_close(got, K @ v, **tol)
    torch.testing.assert_close(cbind.kernel_matmat(kind, X1, X2, ard, M, 2.0), K @ M, **tol)
    n1, n2, d, k = 301, 1013, 19, 5
    A1 = torch.randn(n1, d, dtype=dt, generator=gen) / d**0.5
    A2 = torch.randn(n2, d, dtype=dt, generator=gen) / d**0.5
    V = torch.randn(n2, k, dtype=dt, generator=gen)
    for ell in (0.7, torch.rand(d, dtype=dt, generator=gen) + 0.5):
        want = ok.kernel_matmat(kind, A1.double(), A2.double(), ell.double() if isinstance(ell, torch.Tensor) else ell,
                                V.double(), 1.5)
        got = cbind.kernel_matmat(kind, A1, A2, ell, V, 1.5)
        assert got.dtype == dt
        lim = 1e-5 if dt == torch.float32 else 1e-12
        assert float((got.double() - want).norm() / want.norm()) < lim
    with pytest.raises(ValueError):
        cbind.kernel_matmat(kind, A1[:0], A2, 1.0, V)
