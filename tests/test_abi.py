"""The C-ABI library loads and exports every symbol include/swr.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from softwarerenderer_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "swr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_declare_the_same_entry_points():
    assert header_functions() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, missing
    lib.swr_abi_version.restype = ctypes.c_int
    assert lib.swr_abi_version() == 1


def test_struct_layouts_match_the_reference_types():
    assert ctypes.sizeof(_native.Vertex) == 48                 # Shaders.VertexInput, Shaders.cs:10-24
    assert _native.Vertex.uv.offset == 12 and _native.Vertex.normal.offset == 20 and _native.Vertex.color.offset == 32
    assert ctypes.sizeof(_native.Uniforms) == 4 * (4 + 4 + 4 + 4 + 4) + 4 * 32
    assert ctypes.sizeof(_native.Stats) == 64


def test_no_device_fails_loudly_instead_of_falling_back():
    """On a box without a usable GPU the product path must raise, never compute on the CPU."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    from softwarerenderer_amd import Device
    with pytest.raises(_native.SwrError) as e:
        Device(0)
    assert e.value.code in (_native.SWR_ERR_NO_DEVICE, _native.SWR_ERR_HIP)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "softwarerenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboswr" not in txt, f
