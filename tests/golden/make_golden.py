#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE itself.

Run in the build container only (``python tests/golden/make_golden.py``): it imports
the reference solver stack unmodified from /root/reference and writes small ``.npz``
files (inputs + reference outputs) next to this script.  Neither this script nor the
reference travels to the GPU box; only the ``.npz`` data does.

How the import works: ``rlaopt/__init__.py`` imports the unbuilt
sparse-only extension ``rlaopt._C`` and ``rlaopt/utils/logger.py`` imports ``wandb``;
neither is touched on the path exercised here, so two inert empty modules are placed in
``sys.modules`` first.  ``rlaopt.kernels`` cannot be imported (pykeops is absent), so
kernel fixtures come from the reference test-suite's dependency-free closed-form oracle
``tests/kernels/utils.py`` and the KRR solver fixtures wrap a dense closed-form K in the
reference's own ``SymmetricLinOp`` / ``LinOp``.

Random draws made *inside* reference code (sketch matrices, SAP blocks, power-iteration
start vectors) are recorded by wrapping ``torch.randn`` / ``torch.multinomial`` for the
duration of the call, so that a non-CPU implementation can be fed the very same draws.
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

for _name in ("rlaopt._C", "wandb"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.path.insert(0, REF)

from rlaopt.linops import LinOp, SymmetricLinOp  # noqa: E402
from rlaopt.models import LinSys  # noqa: E402
from rlaopt.preconditioners import IdentityConfig, NystromConfig, SkPreConfig  # noqa: E402
from rlaopt.preconditioners.nystrom import Nystrom  # noqa: E402
from rlaopt.preconditioners.skpre import SkPre  # noqa: E402
from rlaopt.sketches import get_sketch  # noqa: E402
from rlaopt.solvers import PCGConfig, SAPAccelConfig, SAPConfig  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_kernel_utils", f"{REF}/tests/kernels/utils.py")
ref_kernel_utils = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_kernel_utils)

CPU = torch.device("cpu")
DT = {"f32": torch.float32, "f64": torch.float64}
KERNEL_FUNCS = {
    "rbf": ref_kernel_utils.rbf_kernel,
    "laplace": ref_kernel_utils.laplace_kernel,
    "matern12": ref_kernel_utils.matern12_kernel,
    "matern32": ref_kernel_utils.matern32_kernel,
    "matern52": ref_kernel_utils.matern52_kernel,
}


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


class Recorder:
    """Records torch.randn / torch.multinomial results while active."""

    def __init__(self):
        self.randn, self.multinomial = [], []

    def __enter__(self):
        self._randn, self._multi = torch.randn, torch.multinomial

        def randn(*a, **k):
            out = self._randn(*a, **k)
            self.randn.append(out.clone())
            return out

        def multinomial(*a, **k):
            out = self._multi(*a, **k)
            self.multinomial.append(out.clone())
            return out

        torch.randn, torch.multinomial = randn, multinomial
        return self

    def __exit__(self, *exc):
        torch.randn, torch.multinomial = self._randn, self._multi


class _Cfg:  # what the closed-form oracle reads (tests/kernels/utils.py:20-23)
    def __init__(self, lengthscale, const_scaling):
        self.lengthscale, self.const_scaling = lengthscale, const_scaling


def closed_form_K(kind, X1, X2, lengthscale, const_scaling):
    return ref_kernel_utils.compute_kernel_matrix(
        X1, X2, _Cfg(lengthscale, const_scaling), CPU, X1.dtype, KERNEL_FUNCS[kind]
    )


# ----------------------------------------------------------------------------------------
def gen_kernels():
    """Closed-form K at the reference's test shapes (tests/kernels/test_standard.py:44-56,
    114-130: A1 10x3, A2 5x3, const_scaling 2.0, scalar 1.0 / ARD [1,2,3]) and one ragged
    larger shape."""
    for tag, dt in DT.items():
        torch.manual_seed(100)
        shapes = {"small": (10, 5, 3), "ragged": (37, 23, 7)}
        data = {}
        for sname, (n1, n2, d) in shapes.items():
            X1 = torch.randn(n1, d, dtype=dt)
            X2 = torch.randn(n2, d, dtype=dt)
            ard = torch.arange(1, d + 1, dtype=dt)
            data[f"{sname}_X1"], data[f"{sname}_X2"], data[f"{sname}_ard"] = X1, X2, ard
            for kind in KERNEL_FUNCS:
                data[f"{sname}_{kind}_scalar"] = closed_form_K(kind, X1, X2, 1.0, 2.0)
                data[f"{sname}_{kind}_ard"] = closed_form_K(kind, X1, X2, ard, 2.0)
        save(f"kernels_{tag}", **data)


def gen_sketches():
    """Reference embeddings for fixed seeds (sketches/gauss.py, ortho.py, sparse.py)."""
    data = {}
    for tag, dt in DT.items():
        for name in ("gauss", "ortho", "sparse"):
            for mode in ("left", "right"):
                torch.manual_seed(7)
                Om = get_sketch(name, mode, 12, 40, dt, CPU).Omega_mat
                data[f"{name}_{mode}_{tag}"] = Om.contiguous()
    save("sketches", **data)


def gen_nystrom():
    """Nystrom build / apply on the reference's own test matrix shape
    (tests/preconditioners/test_nystrom.py:31-66: 50x50 A A^T, rank 20, rho 1) and on a
    decaying-spectrum SPD matrix accessed through a SymmetricLinOp."""
    for tag, dt in DT.items():
        data = {}
        torch.manual_seed(11)
        G = torch.randn(50, 50, dtype=dt)
        A50 = G @ G.T
        Q, _ = torch.linalg.qr(torch.randn(300, 300, dtype=torch.float64))
        A300 = (Q * (torch.arange(1, 301, dtype=torch.float64) ** -2.0)) @ Q.T
        A300 = ((A300 + A300.T) / 2).to(dt)
        x1 = torch.randn(50, dtype=dt)
        x2 = torch.randn(300, 3, dtype=dt)
        for case, A, rank, x, rho0 in (("a50", A50, 20, x1, 1.0), ("a300", A300, 40, x2, 1e-6)):
            data[f"{case}_A"], data[f"{case}_x"] = A, x
            for sketch in ("gauss", "ortho", "sparse"):
                for mode in ("adaptive", "non_adaptive"):
                    key = f"{case}_{sketch}_{mode}"
                    cfg = NystromConfig(rank=rank, rho=rho0, sketch=sketch, damping_mode=mode)
                    P = Nystrom(cfg)
                    op = A
                    if case == "a300":
                        op = SymmetricLinOp(CPU, A.shape, lambda v, A=A: A @ v, lambda v, A=A: A @ v, dtype=dt)
                    torch.manual_seed(5)
                    with Recorder() as rec:
                        P._update(op, CPU)
                    P._update_damping(0.5 if case == "a50" else 1e-6)
                    if sketch == "gauss":  # gauss.py:46-52
                        Om = (rec.randn[0] / rank**0.5).T.contiguous()
                    elif sketch == "ortho":
                        Om = torch.linalg.qr(rec.randn[0], mode="reduced")[0].contiguous()
                    else:
                        torch.manual_seed(5)
                        Om = get_sketch("sparse", "right", rank, A.shape[1], dt, CPU).Omega_mat.contiguous()
                    data[f"{key}_Omega"] = Om
                    data[f"{key}_U"], data[f"{key}_S"] = P.U, P.S
                    data[f"{key}_rho"] = torch.as_tensor(P.config.rho, dtype=torch.float64)
                    data[f"{key}_Px"] = P @ x
                    data[f"{key}_Pinvx"] = P._inv @ x
        save(f"nystrom_{tag}", **data)


def gen_skpre():
    """SkPre (preconditioners/skpre.py) on a tall matrix; gauss sketch recorded."""
    for tag, dt in DT.items():
        torch.manual_seed(21)
        A = torch.randn(400, 24, dtype=dt) * torch.logspace(0, -2, 24, dtype=dt)
        x = torch.randn(24, 2, dtype=dt)
        cfg = SkPreConfig(sketch_size=96, rho=1e-3, sketch="gauss")
        P = SkPre(cfg)
        torch.manual_seed(3)
        with Recorder() as rec:
            P._update(A, CPU)
        Om = (rec.randn[0] / 96**0.5).contiguous()
        save(f"skpre_{tag}", A=A, x=x, Omega=Om, L=P.L, Px=P @ x, Pinvx=P._inv @ x)


def _log_arrays(log):
    its = sorted(log)
    absr = torch.stack([log[i]["metrics"]["internal_metrics"]["abs_res"] for i in its])
    relr = torch.stack([log[i]["metrics"]["internal_metrics"]["rel_res"] for i in its])
    return np.array(its), absr, relr


def gen_pcg():
    """LinSys.solve(PCGConfig(NystromConfig)) on (i) a dense SPD matrix passed as a raw
    tensor (the C1 pattern) and (ii) an RBF KRR system whose K is the closed
    form wrapped in the reference's SymmetricLinOp (the C2/C3 pattern)."""
    for tag, dt in DT.items():
        data = {}
        # fp32 runs use a milder spectrum / larger reg so that the reference itself
        # converges in fp32 (with the fp64 settings it stagnates at 1e-3 or returns NaN:
        # chaotic trajectories pin nothing)
        f64 = dt == torch.float64
        tol = 1e-8 if f64 else 1e-4
        decay = -2.0 if f64 else -1.0
        # (i) dense SPD, eigenvalues i^-2
        torch.manual_seed(31)
        n = 384
        Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64))
        A = (Q * (torch.arange(1, n + 1, dtype=torch.float64) ** decay)) @ Q.T
        A = ((A + A.T) / 2).to(dt)
        # (ii) RBF KRR
        m, d = 320, 6
        X = (torch.randn(m, d, dtype=torch.float64) / d**0.5).to(dt)
        K = closed_form_K("rbf", X, X, 1.0, 1.0)
        data["dense_A"], data["krr_X"] = A, X  # K = closed form of X (kernels_*.npz pins it)
        for case, op_dense, nn, reg in (
            ("dense", A, n, 1e-6 if f64 else 1e-3),
            ("krr", K, m, (1e-5 if f64 else 1e-4) * m),
        ):
            for k in (1, 3):
                for sketch in ("gauss", "ortho"):
                    key = f"{case}_k{k}_{sketch}"
                    torch.manual_seed(41 + k)
                    B = torch.randn(nn, k, dtype=dt)
                    if case == "dense":
                        op = op_dense
                    else:
                        op = SymmetricLinOp(
                            CPU, op_dense.shape, lambda v, M=op_dense: M @ v,
                            lambda v, M=op_dense: M @ v, dtype=dt,
                        )
                    rank = 48
                    cfg = PCGConfig(
                        device=CPU, max_iters=300, atol=0.0, rtol=tol,
                        precond_config=NystromConfig(rank=rank, rho=reg, sketch=sketch),
                    )
                    sysm = LinSys(op, B if k > 1 else B[:, 0], reg=reg)
                    torch.manual_seed(51)
                    with Recorder() as rec:
                        W, log = sysm.solve(cfg, torch.zeros(nn, k, dtype=dt), callback_freq=5)
                    if sketch == "gauss":
                        Om = (rec.randn[0] / rank**0.5).T.contiguous()
                    else:
                        Om = torch.linalg.qr(rec.randn[0], mode="reduced")[0].contiguous()
                    its, absr, relr = _log_arrays(log)
                    data[f"{key}_B"], data[f"{key}_Omega"], data[f"{key}_W"] = B, Om, W
                    data[f"{key}_iters"], data[f"{key}_abs_res"], data[f"{key}_rel_res"] = its, absr, relr
                    data[f"{key}_reg"] = np.float64(reg)
                    data[f"{key}_rtol"] = np.float64(tol)
        save(f"pcg_{tag}", **data)


def gen_sap():
    """LinSys.solve(SAPConfig) (ASkotch) with dense closed-form RBF oracles wrapped in the
    reference's LinOp; every internal draw (block, sketch, power-iteration start) recorded."""
    for tag, dt in DT.items():
        torch.set_default_dtype(dt)  # spectral_norm.py:16 draws in the default dtype
        data = {}
        torch.manual_seed(61)
        n, d, k, blk_sz, rank = 480, 5, 2, 48, 16
        X = (torch.randn(n, d, dtype=torch.float64) / d**0.5).to(dt)
        B = torch.randn(n, k, dtype=dt)
        K = closed_form_K("rbf", X, X, 1.0, 1.0)
        reg = 1e-2

        def row_oracle(blk, K=K):
            M = K[blk]
            return LinOp(CPU, M.shape, lambda v: M @ v, lambda v: M @ v, dtype=dt)

        def blk_oracle(blk, K=K):
            M = K[blk][:, blk]
            return LinOp(CPU, M.shape, lambda v: M @ v, lambda v: M @ v, dtype=dt)

        op = SymmetricLinOp(CPU, K.shape, lambda v: K @ v, lambda v: K @ v, dtype=dt)
        data["X"], data["B"] = X, B  # K = closed form of X (kernels_*.npz pins it)
        for pname, pcfg in (
            ("nystrom", NystromConfig(rank=rank, rho=reg, sketch="gauss")),
            ("identity", IdentityConfig()),
        ):
            for accel in (True, False):
                key = f"{pname}_{'accel' if accel else 'plain'}"
                cfg = SAPConfig(
                    device=CPU, max_iters=40, atol=0.0, rtol=1e-12, precond_config=pcfg,
                    blk_sz=blk_sz, accel=accel,
                    accel_config=SAPAccelConfig(mu=reg, nu=10.0) if accel else None,
                )
                sysm = LinSys(op, B, reg=reg, A_row_oracle=row_oracle, A_blk_oracle=blk_oracle)
                torch.manual_seed(71)
                with Recorder() as rec:
                    W, log = sysm.solve(cfg, torch.zeros(n, k, dtype=dt), callback_freq=10)
                its, absr, relr = _log_arrays(log)
                data[f"{key}_W"] = W
                data[f"{key}_iters"], data[f"{key}_abs_res"], data[f"{key}_rel_res"] = its, absr, relr
                data[f"{key}_blks"] = torch.stack(rec.multinomial)
                if pname == "nystrom":  # draws alternate: sketch (rank x blk), start (blk,)
                    data[f"{key}_omegas"] = torch.stack(
                        [(g / rank**0.5).T.contiguous() for g in rec.randn[0::2]]
                    )
                    data[f"{key}_v0s"] = torch.stack(rec.randn[1::2])
                else:
                    data[f"{key}_v0s"] = torch.stack(rec.randn)
        data["reg"], data["nu"] = np.float64(reg), np.float64(10.0)
        save(f"sap_{tag}", **data)
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    gen_kernels()
    gen_sketches()
    gen_nystrom()
    gen_skpre()
    gen_pcg()
    gen_sap()
