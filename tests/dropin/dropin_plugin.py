"""pytest plugin for tests/test_reference_suite_dropin.py: makes ``rlaopt`` resolve to
``rlaopt_amd`` and (CPU container only) installs the oracle as compute backend."""
import rlaopt  # noqa: F401  the alias package: import before pytest prepends the reference root

import pytest
from _oracle_backend import OracleBackend
from rlaopt_amd.ops import use_backend


@pytest.fixture(autouse=True, scope="session")
def _oracle_backend_for_cpu():
    with use_backend(OracleBackend()):
        yield
