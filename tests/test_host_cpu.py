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
