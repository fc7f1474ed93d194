"""One fresh worker process per device for a caller that lives in ONE process.

The reference's multi-GPU user API is single-process: ``DistributedRBFLinOp(A1, A2, cfg, devices={cuda:0, ...})``
spawns one worker PROCESS per device and ships every operand of every product to it through queues
(rlaopt/linops/base.py:157-229).  This engine's multi-GPU layout is one rank per GPU over RCCL with the solver state
row-partitioned; a single process that names several devices is mapped onto that layout at the
granularity of the whole SOLVE instead of the single product:

    LinSys(A, B, reg).solve(cfg, W0)            # A built with devices={cuda:0, ..., cuda:p-1} in one process
      -> spmd.solve_kernel_system(...)          # p children, rank i on device i, torch.distributed ("nccl" = RCCL)
           each child: DistributedXLinOp(A1, A2, cfg, devices={its device}) - the SPMD operator - + LinSys.solve
      <- rank 0's (W, log)

Children are started with ``subprocess`` (a new interpreter each: nothing of this process's GPU state is inherited and
this process is never replaced - no exec from a process that has touched the GPU); operands travel once through a
file in shared memory (X is 51 MB at n = 200 000).  ``run`` is the generic entry: ``fn(ctx, *args)`` in every rank.

Status: the dispatch is covered by CPU tests (two gloo ranks); it has NOT run on more than one physical GPU (no
multi-GPU box was reachable by any round).
"""

from __future__ import annotations

import importlib
import os
import pickle
import socket
import subprocess
import sys
import tempfile
from dataclasses import dataclass, replace
from typing import Any, Callable, Sequence

import torch

__all__ = ["run", "solve_kernel_system", "WorkerContext"]


@dataclass
class WorkerContext:
    rank: int
    world: int
    device: torch.device
    backend: str


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _qualname(fn: Callable | str) -> str:
    if isinstance(fn, str):
        return fn
    mod, name = getattr(fn, "__module__", None), getattr(fn, "__qualname__", None)
    if not mod or not name or "<locals>" in name or mod == "__main__":
        raise ValueError("spmd.run needs a module-level function of an importable module (it is re-imported by name in "
                         "every worker process)")
    return f"{mod}:{name}"


def _resolve(spec: str) -> Callable:
    mod, name = spec.split(":", 1)
    obj: Any = importlib.import_module(mod)
    for part in name.split("."):
        obj = getattr(obj, part)
    return obj


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def run(fn: Callable | str, devices: Sequence[torch.device], args: tuple = (), kwargs: dict | None = None,
        backend: str | None = None, init: str | None = None, timeout: float | None = None, env: dict | None = None):
    """Call ``fn(ctx, *args, **kwargs)`` in ``len(devices)`` fresh processes - rank i on ``devices[i]``, joined in a
    ``torch.distributed`` group (``backend``: "nccl" when every device is a GPU, else "gloo") - and return rank 0's
    return value (tensors come back on the CPU).  ``init``: "module:function" called in every worker before the group
    is created (tests install their stand-in compute backend with it).  Raises RuntimeError with the workers' output
    when a rank fails."""
    devices = [torch.device(d) for d in devices]
    if not devices:
        raise ValueError("spmd.run: no devices")
    if backend is None:
        backend = "nccl" if all(d.type == "cuda" for d in devices) else "gloo"
    spec = _qualname(fn)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(prefix="rk_spmd_", dir=shm) as tmp:
        torch.save({"fn": spec, "args": _to_cpu(tuple(args)), "kwargs": _to_cpu(dict(kwargs or {})), "init": init,
                    "devices": [str(d) for d in devices], "backend": backend}, os.path.join(tmp, "payload.pt"))
        port = _free_port()
        procs = []
        for rank in range(len(devices)):
            e = dict(os.environ)
            e.update(env or {})
            e.update(RANK=str(rank), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(port), RK_SPMD_DIR=tmp)
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC)
            e["PYTHONPATH"] = os.pathsep.join(p for p in sys.path if p) + os.pathsep + e.get("PYTHONPATH", "")
            procs.append(subprocess.Popen([sys.executable, "-m", "rlaopt_amd.spmd"], env=e, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        outs = []
        try:
            for p in procs:
                out, _ = p.communicate(timeout=timeout)
                outs.append(out)
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            raise RuntimeError(f"spmd.run: workers did not finish within {timeout} s")
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
        if bad:
            tails = "\n".join(f"--- rank {r} (exit {rc}) ---\n{outs[r][-3000:]}" for r, rc in bad)
            raise RuntimeError(f"spmd.run: {len(bad)} of {len(procs)} workers failed\n{tails}")
        with open(os.path.join(tmp, "result.pkl"), "rb") as f:
            return pickle.load(f)


def _worker_main() -> None:
    tmp = os.environ["RK_SPMD_DIR"]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    payload = torch.load(os.path.join(tmp, "payload.pt"), weights_only=False)
    if payload["init"]:
        _resolve(payload["init"])()
    device = torch.device(payload["devices"][rank])
    import torch.distributed as td

    if device.type == "cuda":
        torch.cuda.set_device(device)
    if payload["backend"] == "nccl":
        td.init_process_group("nccl", device_id=device)
    else:
        td.init_process_group(payload["backend"])
    try:
        ctx = WorkerContext(rank=rank, world=world, device=device, backend=payload["backend"])
        result = _resolve(payload["fn"])(ctx, *payload["args"], **payload["kwargs"])
        if rank == 0:
            with open(os.path.join(tmp, "result.pkl"), "wb") as f:
                pickle.dump(_to_cpu(result), f)
        td.barrier()
    finally:
        td.destroy_process_group()


# ---- the whole solve of a kernel system in SPMD workers --------------------------------------------------------------
def _solve_worker(ctx: WorkerContext, kind: str, A1, A2, kernel_config, use_full_kernel: bool, B, reg: float,
                  solver_config, W_init, callback_freq: int, with_oracles: bool):
    from .kernels import base as _kb
    from .models import LinSys

    dev = ctx.device
    A1d = A1.to(dev)
    A2d = A1d if A2 is None else A2.to(dev)  # (None: the caller's A1 and A2 were the same points - keep them the same)
    A = _kb._DistributedKernelLinOp(A1d, A2d, kernel_config.to(dev), {dev}, use_full_kernel, _kernel_kind=kind)
    oracles = dict(A_row_oracle=A.row_oracle, A_blk_oracle=A.blk_oracle) if with_oracles else {}
    system = LinSys(A, B.to(dev), reg=reg, **oracles)
    cfg = replace(solver_config, device=dev)
    W, log = system.solve(cfg, W_init.to(dev), callback_freq=callback_freq)
    A.shutdown()
    return W, log


def solve_kernel_system(A, B, reg, solver_config, W_init, callback_freq=10, with_oracles=False, backend=None,
                        init=None, timeout=None):
    """``LinSys(A, B, reg).solve(solver_config, W_init)`` for a kernel operator built in ONE process over several
    devices, executed as one rank per device (see the module docstring).  Returns rank 0's (W, log) on the CPU."""
    from .kernels.base import _same_points

    devices = list(A._shard_devices)
    A2 = None if _same_points(A._A1, A._A2) else A._A2
    return run(_solve_worker, devices,
               args=(A._kind, A._A1, A2, A._kernel_config.to(torch.device("cpu")), bool(A._use_full_kernel), B, float(reg),
                     replace(solver_config, device=torch.device("cpu")), W_init, int(callback_freq), bool(with_oracles)),
               backend=backend, init=init, timeout=timeout)


if __name__ == "__main__":
    _worker_main()
