"""Runs the C++ host-mirror parity binary (tests/cpp/test_parity.cpp) on the GPU box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_host_mirror_parity():
    exe = os.path.join(HERE, "cpp", "test_parity")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(HERE, "cpp"), "-s"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout
