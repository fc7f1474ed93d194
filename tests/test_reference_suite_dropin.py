"""Drop-in proof: the REFERENCE's own pytest files (tests/kernels, tests/preconditioners - the
suites that cover this hot path) run unmodified against ``rlaopt_amd`` with
``rlaopt`` aliased to it.  Build container only: the reference tree does not travel to the GPU
box (the test is skipped there) and none of its files are copied into this repository.  On the
CPU the package's host logic is driven with the oracle as compute backend; the same kernels are
checked on the GPU by tests/test_hip_stack.py."""

import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests", "kernels")), reason="reference tree not present")
def test_reference_kernel_and_preconditioner_suites_pass_against_rlaopt_amd(tmp_path):
    dropin = os.path.join(HERE, "dropin")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",  # never write into /root/reference
               PYTHONPATH=os.pathsep.join([dropin, ROOT, HERE, REF]))
    cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-p", "dropin_plugin",
           f"--rootdir={tmp_path}", "-q", "-W", "ignore",
           os.path.join(REF, "tests", "kernels"), os.path.join(REF, "tests", "preconditioners")]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout[-2500:]
    assert out.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    passed = int(tail.rsplit(" passed", 1)[0].split()[-1])
    assert passed >= 216, tail  # 216 tests in the two suites at the reference's 2025-05-16 tree
