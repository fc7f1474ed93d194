"""Seeded random shapes through the GPU kernels added in round 2 - the bf16 matrix-core product in
all its instantiations (1 / 2 / 4 k-steps, VALU or matrix-core second product, gathered operands,
symmetric evaluation), the matrix-core streaming kernels for many right-hand sides (packed symmetric
and row-major) - each against the float64 oracle.  Shapes are drawn around the tile edges (64 / 128
points, 32 / 64 / 128 features, 256 columns) where padding and masking decide."""

import random

import pytest
import torch

from oracle import kernels as ok

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def be():
    from rlaopt_amd.ops import get_backend

    return get_backend()


def _near(rng, centers, spread):
    c = rng.choice(centers)
    return max(1, c + rng.randint(-spread, spread))


def rel_err(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


@pytest.mark.parametrize("seed", range(24))
def test_matrix_free_product_random_shapes(be, monkeypatch, seed):
    rng = random.Random(1000 + seed)
    monkeypatch.setenv("RK_KMAT_SYM_MINN", "1")
    n1 = _near(rng, [1, 64, 128, 129, 300, 1000], 3)
    n2 = _near(rng, [128, 192, 256, 700, 2049], 5)
    if n2 < 128:
        n2 = 128
    d = _near(rng, [1, 8, 32, 33, 64, 65, 100, 128], 1)
    d = min(d, 128)
    k = rng.choice([1, 2, 3, 4, 5, 10, 16, 17])
    kind = rng.choice(["rbf", "matern32", "matern52"])
    ard = rng.random() < 0.3
    gen = torch.Generator().manual_seed(seed)
    offset = 3.0 if rng.random() < 0.3 else 0.0  # un-centred data: the panels are centred in the kernel
    X2 = offset + torch.randn(n2, d, generator=gen) / d**0.5
    same = rng.random() < 0.4
    X1 = X2 if same else offset + torch.randn(n1, d, generator=gen) / d**0.5
    if same:
        n1 = n2
    ls = (0.7 + 0.6 * torch.rand(d, generator=gen)) if ard else 0.9
    V = torch.randn(n2, k, generator=gen)
    idx1 = idx2 = None
    if rng.random() < 0.4:
        idx1 = torch.randperm(X1.shape[0], generator=gen)[: max(1, X1.shape[0] // 2)]
        if same and rng.random() < 0.5 and idx1.numel() >= 128:  # the block oracle: one shared gather
            idx2 = idx1
            V = V[: idx1.numel()]
    X1d, X2d = X1.to(DEV), (X1.to(DEV) if same else X2.to(DEV))
    if same:
        X2d = X1d
    lsd = ls.to(DEV) if ard else ls
    i1 = None if idx1 is None else idx1.to(DEV)
    i2 = None if idx2 is None else (i1 if idx2 is idx1 else idx2.to(DEV))
    got = be.kernel_matmat(kind, X1d, X2d, lsd, V.to(DEV), scale=1.3, idx1=i1, idx2=i2)
    want = ok.kernel_matmat(kind, X1.double(), X2.double(), ls.double() if ard else ls, V.double(), 1.3,
                            idx1=idx1, idx2=idx2)
    assert got.shape == want.shape
    assert rel_err(got, want) < 3e-5, (n1, n2, d, k, kind, ard, same, idx1 is not None, idx2 is not None)


@pytest.mark.parametrize("seed", range(12))
def test_streaming_products_many_columns_random_shapes(be, monkeypatch, seed):
    monkeypatch.setenv("RK_SKINNY_MFMA", "2")  # the matrix-core row-major kernel also for these small shapes
    rng = random.Random(2000 + seed)
    gen = torch.Generator().manual_seed(seed)
    # packed symmetric, 5 .. 16 columns through the matrix-core strip kernel
    n = _near(rng, [255, 256, 257, 1024, 1025, 2300], 2)
    k = rng.randint(5, 16)
    X = torch.randn(n, 6, generator=gen) / 6**0.5
    V = torch.randn(n, k, generator=gen)
    K = ok.kernel_matrix("rbf", X.double(), X.double(), 1.0, 1.0)
    stride = rng.choice([1, 2, 3])
    total = torch.zeros(n, k, dtype=torch.float64)
    for first in range(stride):
        Kp = be.sympack_materialize("rbf", X.to(DEV), 1.0, first=first, stride=stride)
        total += be.symm_matmat(Kp, n, V.to(DEV), first=first, stride=stride).double().cpu()
    assert rel_err(total, K @ V.double()) < 1e-4 / 2, (n, k, stride)
    # row-major, 10 .. 16 columns through the matrix-core kernel (n % 4 == 0, padded leading dimension)
    m = _near(rng, [1, 31, 32, 33, 500], 1)
    nn = 4 * _near(rng, [64, 65, 200, 1000], 2)
    kk = rng.randint(10, 16)
    pad = rng.choice([0, 4, 12])
    Afull = torch.randn(m, nn + pad, generator=gen)
    A = Afull[:, :nn]
    W = torch.randn(nn, kk, generator=gen)
    C = torch.randn(m, kk, generator=gen)
    got = be.matmat(Afull.to(DEV)[:, :nn], W.to(DEV), alpha=0.5, beta=-1.0, C=C.to(DEV))
    assert rel_err(got, 0.5 * A.double() @ W.double() - C.double()) < 1e-4 / 2, (m, nn, kk, pad)
    # the same kernel in float64 (7 .. 16 columns, n % 2 == 0), several row blocks and column segments
    m = _near(rng, [63, 64, 65, 300, 700], 2)
    nn = 2 * _near(rng, [16, 33, 2000, 9000, 15000], 3)
    kk = rng.randint(7, 16)
    pad = rng.choice([0, 2, 6])
    Afull = torch.randn(m, nn + pad, dtype=torch.float64, generator=gen)
    W = torch.randn(nn, kk, dtype=torch.float64, generator=gen)
    got = be.matmat(Afull.to(DEV)[:, :nn], W.to(DEV), diag_shift=0.25 if m <= nn else 0.0)
    want = Afull[:, :nn] @ W + (0.25 * W[:m] if m <= nn else 0.0)
    assert rel_err(got, want) < 1e-13, (m, nn, kk, pad)


@pytest.mark.parametrize("seed", range(16))
def test_block_symmetric_product_and_fused_sketch_random_shapes(be, monkeypatch, seed):
    """Round 4: the block form of the packed symmetric product (any width 1 .. 16, any cyclic strip distribution: the
    ranks' partials add up) and the sketch product with the entries evaluated inside it (rectangular point sets, any
    width up to 600), at sizes around the 256-column strips, the 1024-column groups, the 32-row tiles, the 128-row
    blocks - each against the float64 oracle."""
    rng = random.Random(4000 + seed)
    gen = torch.Generator().manual_seed(seed)
    monkeypatch.setenv("RK_SYMM_F16X2_MIN", "1")  # (one or two columns would take the vector-unit block kernel: below)
    n = _near(rng, [1, 33, 256, 257, 1024, 1025, 2048, 3100, 5121], 2)
    d = rng.choice([3, 16, 40, 64])
    k = rng.choice([1, 2, 3, 5, 8, 11, 16])
    stride = rng.choice([1, 1, 2, 3, 5])
    kind = rng.choice(["rbf", "matern32", "matern52"])
    scale = rng.choice([1.0, 0.3, 4.0])
    X = torch.randn(n, d, generator=gen) / d**0.5
    V = torch.randn(n, k, generator=gen) * torch.logspace(-3, 2, k)
    Xd, Vd = X.to(DEV), V.to(DEV)
    Kd = ok.kernel_matrix(kind, X.double(), X.double(), 1.1, scale)
    want = Kd @ V.double() + 0.5 * V.double()
    total = torch.zeros(n, k, dtype=torch.float64)
    for f in range(stride):
        Kp = be.sympack_materialize(kind, Xd, 1.1, scale=scale, first=f, stride=stride)
        total += be.symm_matmat(Kp, n, Vd, diag_shift=0.5 if f == 0 else 0.0, first=f, stride=stride,
                                kmax=scale).cpu().double()
    err = (total - want).norm(dim=0) / want.norm(dim=0)
    assert float(err.max()) < 5e-6, (n, d, k, stride, kind, float(err.max()))
    # the vector-unit block kernel (exact float32 arithmetic; what one or two columns run by default), and the same with
    # float64 vectors over the float32 strips
    monkeypatch.delenv("RK_SYMM_F16X2_MIN")
    k2 = min(k, 2)
    for vdt in (torch.float32, torch.float64):
        tot2 = torch.zeros(n, k2, dtype=torch.float64)
        Vv = V[:, :k2].to(vdt).contiguous().to(DEV)
        for f in range(stride):
            Kp = be.sympack_materialize(kind, Xd, 1.1, scale=scale, first=f, stride=stride)
            tot2 += be.symm_matmat(Kp, n, Vv, diag_shift=0.5 if f == 0 else 0.0, first=f, stride=stride,
                                   kmax=scale).cpu().double()
        err2 = (tot2 - want[:, :k2]).norm(dim=0) / want[:, :k2].norm(dim=0)
        # (float64 vectors: what remains is the float32 rounding of the STORED entries, evaluated in float32 from float32 data)
        assert float(err2.max()) < (5e-6 if vdt == torch.float32 else 2e-6), (n, d, k2, stride, kind, vdt, float(err2.max()))
    # the fused sketch product on the same points against a second, rectangular set
    n2 = _near(rng, [31, 129, 700, 1500], 3)
    r = rng.choice([17, 64, 130, 257, 600])
    X2 = torch.randn(n2, d, generator=gen) / d**0.5
    Om = torch.randn(n2, r, generator=gen)
    X2d = X2.to(DEV)
    gc = be.gram_center(Xd, X2d, 1.1)
    assert gc is not None
    got = be.kernel_sketch_matmat(kind, Xd, X2d, 1.1, Om.to(DEV), scale=scale, gram_center=gc)
    want2 = ok.kernel_matrix(kind, X.double(), X2.double(), 1.1, scale) @ Om.double()
    assert rel_err(got, want2) < 1e-5, (n, n2, d, r, kind)
    # the sketch product from the packed strips (any cyclic distribution): phase kernel = lockstep kernel bit for bit, and
    # for a block on the float16 grid the one-piece kernels (stored strips and fused evaluation) = the general ones
    from rlaopt_amd.ops import f16_grid

    Vw = torch.randn(n, r, generator=gen)
    Vg = f16_grid(Vw)
    tot = torch.zeros(n, r, dtype=torch.float64)
    for f in range(stride):
        Kp = be.sympack_materialize(kind, Xd, 1.1, scale=scale, first=f, stride=stride)
        monkeypatch.setenv("RK_SYMGEMM_IMPL", "lockstep")
        lock = be.symm_gemm(Kp, n, Vw.to(DEV), kmax=scale, first=f, stride=stride)
        monkeypatch.setenv("RK_SYMGEMM_IMPL", "ph")
        ph = be.symm_gemm(Kp, n, Vw.to(DEV), kmax=scale, first=f, stride=stride)
        assert torch.equal(lock, ph), (n, r, f, stride)
        assert torch.equal(be.symm_gemm(Kp, n, Vg.to(DEV), kmax=scale, first=f, stride=stride, rhs_f16=True),
                           be.symm_gemm(Kp, n, Vg.to(DEV), kmax=scale, first=f, stride=stride)), (n, r, f, stride)
        tot += ph.cpu().double()
    wantw = Kd @ Vw.double()
    assert float(((tot - wantw).norm(dim=0) / wantw.norm(dim=0)).max()) < 5e-6, (n, r, stride, kind)
    Og = f16_grid(Om).to(DEV)
    assert torch.equal(be.kernel_sketch_matmat(kind, Xd, X2d, 1.1, Og, scale=scale, gram_center=gc, rhs_f16=True),
                       be.kernel_sketch_matmat(kind, Xd, X2d, 1.1, Og, scale=scale, gram_center=gc))
