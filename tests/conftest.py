"""Shared pytest configuration.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu`` on the GPU box).
  Everything else runs on CPU (``-m "not gpu"``).
* The oracle (``oracle/``) is test infrastructure: tests may import it, the product
  package ``rlaopt_amd`` never does.
"""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# reference tolerances: tests/kernels/test_standard.py:102-105,
# tests/preconditioners/test_nystrom.py:69-72
TOL = {
    torch.float32: dict(rtol=1e-4, atol=1e-4),
    torch.float64: dict(rtol=1e-8, atol=1e-8),
}
DTYPES = {"f32": torch.float32, "f64": torch.float64}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels execute)")
    # The host side of the tests (the oracle) gets as many threads as this process may actually use: a GPU box
    # is a 16-CPU cgroup share of a 256-thread host, and torch / OpenMP pools sized for the whole host are only
    # throttled (the oracle solves of the parity tests ran several times slower with 128 threads).
    try:
        from oracle import cbind

        cpus = cbind.host_cpus()
        if cpus < torch.get_num_threads():
            torch.set_num_threads(cpus)
        if os.path.exists(cbind.LIB_PATH) and cpus < cbind.num_threads():
            cbind.set_num_threads(cpus)
    except Exception:  # noqa: BLE001 - a tuning step only
        pass


def load_golden(name: str) -> dict:
    """Load tests/golden/<name>.npz as a dict of torch tensors (CPU)."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: torch.from_numpy(np.array(z[k])) for k in z.files}


@pytest.fixture(params=["f32", "f64"])
def prec(request):
    return request.param
