"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo group on CPU.

Drives the package's SPMD code paths (row-sharded kernel operator, all-gather / all-reduce
combination, sharded Nystrom assembly and apply, sharded block PCG, LinSys metrics) with the
oracle injected as compute backend, and compares against a single-process dense solve.
"""

import os
import sys

import torch
import torch.distributed as td

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from _oracle_backend import OracleBackend  # noqa: E402
from oracle import kernels as ok  # noqa: E402
from oracle import linsys as ol  # noqa: E402
from oracle import nystrom as on  # noqa: E402
from oracle import pcg as opcg  # noqa: E402
from rlaopt_amd import dist as rdist  # noqa: E402
from rlaopt_amd import sketches  # noqa: E402
from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig  # noqa: E402
from rlaopt_amd.models import LinSys  # noqa: E402
from rlaopt_amd.ops import use_backend  # noqa: E402
from rlaopt_amd.preconditioners import NystromConfig  # noqa: E402
from rlaopt_amd.solvers import PCGConfig  # noqa: E402


def main():
    n = int(sys.argv[1])
    td.init_process_group("gloo")
    rank, size = td.get_rank(), td.get_world_size()
    cpu = torch.device("cpu")
    dt = torch.float64
    gen = torch.Generator().manual_seed(0)
    if n < 10:
        # fewer row blocks than ranks (torch.chunk(arange(4), 3) has 2 blocks): the last rank owns
        # no rows and must still take part in every collective (kernels/base.py:297-307)
        with use_backend(OracleBackend()):
            X = torch.randn(n, 3, dtype=dt, generator=gen)
            X2 = torch.randn(n + 1, 3, dtype=dt, generator=gen)
            M = torch.randn(n + 1, 2, dtype=dt, generator=gen)
            u = torch.randn(n, 2, dtype=dt, generator=gen)
            cfgk = KernelConfig(lengthscale=1.1)
            op = DistributedRBFLinOp(X, X2, cfgk, devices={cpu})
            K = ok.kernel_matrix("rbf", X, X2, 1.1)
            lo, hi = op.row_range
            assert (hi - lo == 0) == (rank >= len(rdist.row_chunks(n, size)))
            torch.testing.assert_close(op @ M, K @ M, rtol=1e-10, atol=1e-12)
            torch.testing.assert_close(op.T @ u, K.T @ u, rtol=1e-10, atol=1e-12)
            blk = torch.tensor([n - 1, 0])
            torch.testing.assert_close(op.row_oracle(blk) @ M, K[blk] @ M, rtol=1e-10, atol=1e-12)
            sq = DistributedRBFLinOp(X, X, cfgk, devices={cpu})
            Ksq = ok.kernel_matrix("rbf", X, X, 1.1)
            torch.testing.assert_close(sq @ u, Ksq @ u, rtol=1e-10, atol=1e-12)
            w = torch.randn(2, 2, dtype=dt, generator=gen)
            torch.testing.assert_close(sq.blk_oracle(blk) @ w, Ksq[blk][:, blk] @ w, rtol=1e-10, atol=1e-12)
            B = torch.randn(n, 1, dtype=dt, generator=gen)
            W, log = LinSys(sq, B, reg=0.5).solve(PCGConfig(device=cpu, max_iters=20, rtol=1e-12), torch.zeros(n, 1, dtype=dt), callback_freq=1)
            want = torch.linalg.solve(Ksq + 0.5 * torch.eye(n, dtype=dt), B)
            torch.testing.assert_close(W, want, rtol=1e-8, atol=1e-10)
            # the column-sampling Nystrom preconditioner on the same layout (a rank without rows contributes zeros)
            Wc, _ = LinSys(sq, B, reg=0.5).solve(
                PCGConfig(device=cpu, max_iters=20, rtol=1e-12,
                          precond_config=NystromConfig(rank=2, rho=0.5, sketch="colsample")),
                torch.zeros(n, 1, dtype=dt), callback_freq=1)
            torch.testing.assert_close(Wc, want, rtol=1e-8, atol=1e-10)
        td.barrier()
        if rank == 0:
            print(f"DIST-OK world={size} n={n} (fewer blocks than ranks)")
        td.destroy_process_group()
        return
    d, k, r = 5, 2, 24
    X = torch.randn(n, d, dtype=dt, generator=gen) / d**0.5
    X2 = torch.randn(n - 3, d, dtype=dt, generator=gen) / d**0.5
    B = torch.randn(n, k, dtype=dt, generator=gen)
    Om = torch.randn(r, n, dtype=dt, generator=gen).T.contiguous() / r**0.5
    M = torch.randn(n - 3, 3, dtype=dt, generator=gen)
    u = torch.randn(n, 2, dtype=dt, generator=gen)
    cfgk = KernelConfig(const_scaling=1.5, lengthscale=0.9)
    tol = dict(rtol=1e-9, atol=1e-9)
    with use_backend(OracleBackend()):
        # ---- collectives on ragged partitions -----------------------------------------------
        lo, hi = rdist.my_rows(n)
        full = torch.arange(n * 2, dtype=dt).reshape(n, 2)
        torch.testing.assert_close(rdist.all_gather_rows(full[lo:hi], n), full)
        sizes = [3 + 2 * i for i in range(size)]
        off = sum(sizes[:rank])
        stacked = torch.arange(sum(sizes), dtype=dt)
        torch.testing.assert_close(rdist.all_gather_rows(stacked[off : off + sizes[rank]], sizes), stacked)

        # ---- rectangular sharded operator: products, transpose, oracles -----------------------
        op = DistributedRBFLinOp(X, X2, cfgk, devices={cpu})
        assert op.is_row_sharded and op.row_range == (lo, hi)
        K = ok.kernel_matrix("rbf", X, X2, 0.9, 1.5)
        torch.testing.assert_close(op @ M, K @ M, **tol)
        torch.testing.assert_close(op @ M[:, 0].contiguous(), K @ M[:, 0], **tol)
        torch.testing.assert_close(op.T @ u, K.T @ u, **tol)
        torch.testing.assert_close(op.local_matmat(M), (K @ M)[lo:hi], **tol)
        blk = torch.tensor([5, 0, n - 1, 17, 9, 2, 30])
        torch.testing.assert_close(op.row_oracle(blk) @ M, K[blk] @ M, **tol)
        sq = DistributedRBFLinOp(X, X, cfgk, devices={cpu}, use_full_kernel=False)
        Ksq = ok.kernel_matrix("rbf", X, X, 0.9, 1.5)
        w = torch.randn(blk.numel(), 2, dtype=dt, generator=gen)
        torch.testing.assert_close(sq.blk_oracle(blk) @ w, Ksq[blk][:, blk] @ w, **tol)
        torch.testing.assert_close(sq.row_oracle(blk) @ u, Ksq[blk] @ u, **tol)

        # ---- sharded Nystrom + block PCG vs the single-process oracle ---------------------------
        A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={cpu})
        reg = 1e-4 * n
        system = LinSys(A, B, reg=reg)
        cfg = PCGConfig(device=cpu, max_iters=200, atol=0.0, rtol=1e-9,
                        precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
        with sketches.inject(Om):
            W, log = system.solve(cfg, torch.zeros_like(B), callback_freq=5)
        Kd = ok.kernel_matrix("rbf", X, X, 1.0)
        st = on.build(Kd, Om, reg)
        on.update_damping(st, reg, adaptive=True)
        ref = opcg.BlockPCG(lambda V: Kd @ V, lambda R: on.inverse_matmul(st, R), B, reg, torch.zeros_like(B))
        Wref, logref = ol.run(ref, lambda V: Kd @ V, B, reg, 200, 0.0, 1e-9, callback_freq=5)
        assert sorted(log) == sorted(logref), (sorted(log), sorted(logref))
        assert W.shape == B.shape
        res = torch.linalg.norm(B - (Kd @ W + reg * W), dim=0) / torch.linalg.norm(B, dim=0)
        assert bool((res <= 1e-9).all()), res
        early = sorted(log)[:3]
        got = torch.stack([log[i]["metrics"]["internal_metrics"]["rel_res"] for i in early])
        want = torch.stack([logref[i]["rel_res"] for i in early])
        torch.testing.assert_close(got, want, rtol=1e-6, atol=0)
        # collectives per block-PCG iteration with strip-packed SPMD storage: one all-gather (search
        # directions), one reduce-scatter (the partial products, in their own dtype, with P^T AP riding in every
        # slot), one all-reduce (U^T R + R^T R); the Gram blocks do not get collectives of their own.  Bytes on
        # the wire (ring accounting of rlaopt_amd/dist.py): the partial product used to travel as an n k float64
        # all-reduce (2 (p - 1) / p n k 8 B per rank and iteration)
        if A.storage_mode.startswith("sympack"):
            from rlaopt_amd.solvers.pcg import PCG

            with sketches.inject(Om):  # (a fresh system: the solved one has every column masked out)
                sv = PCG(LinSys(A, B, reg=reg), torch.zeros_like(B), cfg.precond_config, cpu)
            rdist.reset_wire_stats()
            for _ in range(3):
                sv._step()
            ws = rdist.wire_stats
            assert ws["by_kind"] == {"all_gather": 3, "reduce_scatter": 3, "all_reduce": 3}, ws
            k, es = B.shape[1], B.element_size()
            step = -(-n // size)
            f = (size - 1) / size
            per_iter = f * (size * step * k * es                      # all-gather of the search directions
                            + size * (step * k + k * k) * es          # reduce-scatter: partial rows + Gram tail
                            + 2 * (r * k + k * k) * 8)                # all-reduce of [U^T R, R^T R] (float64)
            assert abs(ws["bytes_sent_per_rank"] - 3 * per_iter) < 1e-6 * per_iter, (ws, per_iter)
            old_product_bytes = 2 * f * (n * k + k * k) * 8
            assert f * size * (step * k + k * k) * es <= 0.51 * old_product_bytes + 64
            # the all-reduce form stays available (RK_SPMD_REDUCE_SCATTER=0) and gives the same iterates
            import os

            with sketches.inject(Om):
                s2 = PCG(LinSys(A, B, reg=reg), torch.zeros_like(B), cfg.precond_config, cpu)
            os.environ["RK_SPMD_REDUCE_SCATTER"] = "0"
            try:
                rdist.reset_wire_stats()
                for _ in range(3):
                    s2._step()
                assert rdist.wire_stats["by_kind"] == {"all_gather": 3, "all_reduce": 6}, rdist.wire_stats
            finally:
                del os.environ["RK_SPMD_REDUCE_SCATTER"]
            torch.testing.assert_close(s2.W, sv.W, rtol=1e-9, atol=1e-12)
        # every rank returns the same full solution
        Wsum = W.clone()
        td.all_reduce(Wsum)
        torch.testing.assert_close(Wsum / size, W, rtol=1e-12, atol=1e-12)
        # ---- every rank draws the SAME random embedding (seed broadcast from rank 0), although the
        # ranks' own generators differ: gauss / ortho / sparse (sketches/*.py) ---------------------
        torch.manual_seed(1000 + rank)  # ranks deliberately out of step
        for name in ("sparse", "gauss", "ortho"):
            sk = sketches.get_sketch(name, "right", 8, n, dt, cpu)
            Om_mine = sk.Omega_mat.clone()
            Om_sum = Om_mine.clone()
            td.all_reduce(Om_sum)
            torch.testing.assert_close(Om_sum / size, Om_mine, rtol=1e-12, atol=1e-14)
        with torch.random.fork_rng():
            torch.manual_seed(1000 + rank)
            Wsp, logsp = LinSys(A, B, reg=reg).solve(
                PCGConfig(device=cpu, max_iters=200, atol=0.0, rtol=1e-9,
                          precond_config=NystromConfig(rank=r, rho=reg, sketch="sparse")),
                torch.zeros_like(B), callback_freq=5)
        ressp = torch.linalg.norm(B - (Kd @ Wsp + reg * Wsp), dim=0) / torch.linalg.norm(B, dim=0)
        assert bool((ressp <= 1e-9).all()), ressp
        # ---- column-sampling sketch on the sharded operator: every rank draws the same J, writes ITS rows of
        # K[:, J] (no product, no collective), the core is the all-reduced row gather; same preconditioner as
        # the oracle fed the dense Omega --------------------------------------------------------------------
        from rlaopt_amd.preconditioners.nystrom import Nystrom

        torch.manual_seed(2000 + rank)
        skc = sketches.get_sketch("colsample", "right", r, n, dt, cpu)
        Jsum = skc.indices.clone()
        td.all_reduce(Jsum)
        assert torch.equal(Jsum, skc.indices * size)
        torch.testing.assert_close(skc.columns_of(A, local=True), Kd[lo:hi][:, skc.indices] * skc.col_scale, **tol)
        torch.testing.assert_close(skc.columns_of(A), Kd[:, skc.indices] * skc.col_scale, **tol)
        # the preconditioner's own draw, captured (the ranks share one seed stream: a second get_sketch call
        # draws a new subset)
        import rlaopt_amd.preconditioners.nystrom as nymod

        seen = []
        real_get = nymod.get_sketch
        nymod.get_sketch = lambda *a, **k: seen.append(real_get(*a, **k)) or seen[-1]
        try:
            Pc = Nystrom(NystromConfig(rank=r, rho=reg, sketch="colsample"))
            rdist.reset_wire_stats()
            Pc._update(A, cpu)
            Pc._update_damping(reg)
        finally:
            nymod.get_sketch = real_get
        assert "all_gather" not in rdist.wire_stats["by_kind"], rdist.wire_stats  # nothing n-long moved
        Omc = seen[0].Omega_mat
        stc = on.build(Kd, Omc, reg)
        on.update_damping(stc, reg, adaptive=True)
        with sketches.inject(Omc):  # the dense route (products + collectives) on the same matrix
            Pd = Nystrom(NystromConfig(rank=r, rho=reg, sketch="gauss"))
            Pd._update(A, cpu)
            Pd._update_damping(reg)
        assert abs(Pc.config.rho - stc.rho) <= 1e-8 * abs(stc.rho) + 1e-12, (Pc.config.rho, Pd.config.rho, stc.rho)
        assert abs(Pd.config.rho - stc.rho) <= 1e-8 * abs(stc.rho) + 1e-12
        rloc = B[lo:hi]
        want_inv = on.inverse_matmul(stc, B)[lo:hi]
        torch.testing.assert_close(Pc._inv @ rloc if not Pc._sharded else Pc._inverse_matmul_with_gram(rloc)[0],
                                   want_inv, rtol=1e-7, atol=1e-9)
        Wcs, logcs = LinSys(A, B, reg=reg).solve(
            PCGConfig(device=cpu, max_iters=200, atol=0.0, rtol=1e-9,
                      precond_config=NystromConfig(rank=r, rho=reg, sketch="colsample")),
            torch.zeros_like(B), callback_freq=5)
        rescs = torch.linalg.norm(B - (Kd @ Wcs + reg * Wcs), dim=0) / torch.linalg.norm(B, dim=0)
        assert bool((rescs <= 1e-9).all()), rescs
        # ---- SPMD sketch-and-project (distributed row / block oracles, replicated iterates) --------
        from oracle import sap as osap
        from rlaopt_amd.solvers import SAPAccelConfig, SAPConfig

        blk_sz, rk, steps = 12, 6, 8
        gsap = torch.Generator().manual_seed(123)
        blocks = [torch.randperm(n, generator=gsap)[:blk_sz] for _ in range(steps)]
        omegas = [(torch.randn(rk, blk_sz, dtype=dt, generator=gsap) / rk**0.5).T.contiguous() for _ in range(steps)]
        starts = [torch.randn(blk_sz, dtype=dt, generator=gsap) for _ in range(steps)]
        regs = 1e-2
        Aop = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={cpu})
        sysm = LinSys(Aop, B, reg=regs, A_row_oracle=Aop.row_oracle, A_blk_oracle=Aop.blk_oracle)
        scfg = SAPConfig(device=cpu, max_iters=steps, atol=0.0, rtol=1e-14,
                         precond_config=NystromConfig(rank=rk, rho=regs, sketch="gauss"), blk_sz=blk_sz,
                         accel_config=SAPAccelConfig(mu=regs, nu=10.0))
        import rlaopt_amd.models.linsys as ml

        orig = ml._get_solver

        def patched(model, W_init, solver_config):
            sv = orig(model=model, W_init=W_init, solver_config=solver_config)
            sv.inject_draws(blocks=blocks, starts=starts)
            return sv

        ml._get_solver = patched
        try:
            with sketches.inject(*omegas):
                Wsap, _ = sysm.solve(scfg, torch.zeros_like(B), callback_freq=4)
        finally:
            ml._get_solver = orig
        import torch as _t

        _old = _t.get_default_dtype()
        _t.set_default_dtype(dt)
        try:
            ref = osap.SAP(n, lambda b: Kd[b], lambda b: Kd[b][:, b], B, regs, torch.zeros_like(B),
                           ("nystrom", rk, regs, "gauss", True), blk_sz=blk_sz, accel=True, mu=regs,
                           nu=10.0, blk_sequence=blocks)
            it_om, it_v0 = iter(omegas), iter(starts)
            import oracle.sketches as osk2
            import oracle.sap as osap_mod

            real_sk = osk2.get_sketch_matrix
            osk2.get_sketch_matrix = lambda *a, **k: next(it_om)
            real_pow = osap_mod.randomized_powering
            osap_mod.randomized_powering = lambda f, d_, dev_, max_iters=10, rtol=1e-3, v0=None: real_pow(
                f, d_, dev_, max_iters=max_iters, rtol=rtol, v0=next(it_v0))
            try:
                for _ in range(steps):
                    ref.step(torch.ones(k, dtype=torch.bool))
            finally:
                osk2.get_sketch_matrix, osap_mod.randomized_powering = real_sk, real_pow
        finally:
            _t.set_default_dtype(_old)
        torch.testing.assert_close(Wsap, ref.W, rtol=1e-8, atol=1e-10)
        # ---- the same solve with its own draws (blocks, Ortho sketches, start vectors): the ranks'
        # generators differ, yet every rank ends with the same iterate, and the whole solve costs ONE
        # object broadcast (the shared seed stream), not three per step -------------------------------
        count = {"n": 0}
        real_bcast = td.broadcast_object_list

        def counting(*a, **k):
            count["n"] += 1
            return real_bcast(*a, **k)

        td.broadcast_object_list = counting
        try:
            torch.manual_seed(500 + rank)
            scfg2 = SAPConfig(device=cpu, max_iters=steps, atol=0.0, rtol=1e-14,
                              precond_config=NystromConfig(rank=rk, rho=regs), blk_sz=blk_sz,
                              accel_config=SAPAccelConfig(mu=regs, nu=10.0))
            Wown, _ = sysm.solve(scfg2, torch.zeros_like(B), callback_freq=4)
        finally:
            td.broadcast_object_list = real_bcast
        assert count["n"] == 1, count
        Wsum = Wown.clone()
        td.all_reduce(Wsum)
        torch.testing.assert_close(Wsum / size, Wown, rtol=1e-12, atol=1e-12)
        assert bool(torch.isfinite(Wown).all()) and float(Wown.abs().max()) > 0
    td.barrier()
    if rank == 0:
        print(f"DIST-OK world={size} n={n} iters={max(log)}")
    td.destroy_process_group()


if __name__ == "__main__":
    main()
