"""Runs the C++ host-mirror parity binary (tests/cpp/test_parity.cpp) on the GPU box."""
import os

import pytest

import spawner

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_cpp_host_mirror_parity():
    exe = os.path.join(HERE, "cpp", "test_parity")
    if not os.path.exists(exe):
        spawner.run(["make", "-C", os.path.join(HERE, "cpp"), "-s"], check=True)
    r = spawner.run([exe], timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout


def test_plain_c_caller_renders_cfg1_through_the_abi():
    """tests/c/abi_smoke.c: gcc + dlopen, no C++ / HIP headers -- resolves every declared symbol and renders BASELINE cfg1
    through swr_render_mesh_arrays (the array form of the reference's RenderMesh signature)."""
    from softwarerenderer_amd import _native
    exe = os.path.join(ROOT, "tests", "c", "abi_smoke")
    spawner.run(["make", "-C", os.path.join(ROOT, "tests", "c"), "-s"], check=True)
    out = spawner.run([exe, _native.LIB_PATH], timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "covered=8321 wrong=0 depth_touched=0" in out.stdout
