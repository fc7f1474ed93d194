"""SPMD control flow with the real HIP kernels on the one-GPU test box (see
tests/_dist_gpu_worker.py): 2 ranks sharing cuda:0 over gloo, and a single-rank RCCL group."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("backend,world,n", [("gloo", 2, 3001), ("gloo", 3, 2000), ("nccl", 1, 3001)])
def test_spmd_on_gpu(backend, world, n):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if backend == "nccl":
            env["RK_FORCE_SPMD"] = "1"
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), backend, str(n)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
        ))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
    assert outs[0].count("DIST-GPU-OK") == 2
