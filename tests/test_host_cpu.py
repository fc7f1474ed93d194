"""Host-side logic of the package on CPU, with the oracle injected as compute backend
(``tests/_oracle_backend.py``).  The cases live in ``stack_cases.py``."""

import pytest
import torch

from _oracle_backend import OracleBackend
from rlaopt_amd.ops import use_backend
from stack_cases import *  # noqa: F401,F403


@pytest.fixture
def dev():
    return torch.device("cpu")


@pytest.fixture(autouse=True)
def oracle_backend():
    with use_backend(OracleBackend()):
        yield


def test_quiet_host_limits_and_restores_threads():
    """utils.quiet_host: small CPU thread pool inside solver loops, re-entrant, restored on exit."""
    from rlaopt_amd.utils import quiet_host, small_host_lapack

    prev = torch.get_num_threads()
    try:
        torch.set_num_threads(max(prev, 4))
        outer = torch.get_num_threads()
        with quiet_host(max_threads=2):
            assert torch.get_num_threads() == min(outer, 2)
            with quiet_host(max_threads=1):  # nested: the outermost context decides
                assert torch.get_num_threads() == min(outer, 2)
                with small_host_lapack(64):
                    assert torch.get_num_threads() == 1
                assert torch.get_num_threads() == min(outer, 2)
            assert torch.get_num_threads() == min(outer, 2)
        assert torch.get_num_threads() == outer
        with pytest.raises(RuntimeError):
            with quiet_host(max_threads=2):
                raise RuntimeError("boom")
        assert torch.get_num_threads() == outer  # restored after an exception too
    finally:
        torch.set_num_threads(prev)
