"""The parity tests proper: the full stack through the C ABI on cuda:0 against the
reference's golden vectors.  The cases live in ``stack_cases.py``."""

import pytest
import torch

from stack_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")
