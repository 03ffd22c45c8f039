"""Uninitialised-memory check: C3R_POISON=<byte> makes libc3r fill every fresh device allocation with that byte (c3r_lib.hip,
poison_byte).  A kernel that reads memory nobody wrote then sees 0x01010101 / NaN patterns instead of whatever the allocator
handed out, and the parity tests of the child run fail.  (GPU AddressSanitizer is not available on this pool.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("byte", [1, 255])
def test_parity_suites_pass_with_poisoned_allocations(byte):
    if os.environ.get("C3R_POISON"):
        pytest.skip("already inside a poisoned run")
    env = dict(os.environ, C3R_POISON=str(byte))
    files = ["tests/test_gpu_configs.py", "tests/test_gpu_sample.py", "tests/test_gpu_parity.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
