"""ctypes binding of libswr_hip.so (the C ABI declared in include/swr.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable the
import / context creation raises.  PyTorch is optional plumbing (device memory, streams,
torch.distributed); when it is already imported the library binds to the HIP runtime torch
loaded (same SONAME), so both live on one runtime.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWR_LIB selects another in-tree build of the SAME source (the numerics sensitivity builds libswr_hip_fma.so /
# libswr_hip_dotpw.so of csrc/Makefile, or a test build); there is no non-HIP alternative to select
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("SWR_LIB", "") or "libswr_hip.so"))

SWR_OK = 0
SWR_ABI_EXPECTED = 3
SWR_ERR_INVALID_ARG = -1
SWR_ERR_HIP = -2
SWR_ERR_OOM = -3
SWR_ERR_NO_DEVICE = -4
SWR_ERR_UNSUPPORTED = -5
SWR_STALE = 1          # swr_present_wait: not an error -- the copied frame predates a replayed batch, present again


class SwrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"swr error {code}: {msg}")
        self.code = code


class Vertex(C.Structure):  # Shaders.VertexInput, Shaders.cs:10-24 (48 bytes)
    _fields_ = [("position", C.c_float * 3), ("uv", C.c_float * 2), ("normal", C.c_float * 3), ("color", C.c_float * 4)]


class PointLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("range", C.c_float), ("color", C.c_float * 3), ("intensity", C.c_float)]


class Uniforms(C.Structure):  # swr_uniforms == fields of Renderer.cs:39-44 (+ build-defined Phong block)
    _fields_ = [
        ("light_direction", C.c_float * 3), ("_pad0", C.c_float),
        ("light_color", C.c_float * 4),
        ("fog_color", C.c_float * 4),
        ("fog_start", C.c_float), ("fog_end", C.c_float),
        ("shininess", C.c_float), ("_pad1", C.c_float),
        ("camera_position", C.c_float * 3), ("_pad2", C.c_float),
        ("lights", PointLight * 4),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "triangles_in", "triangles_setup", "triangles_clipped", "fragments_tested",
        "fragments_shaded", "fragments_written", "tile_pairs", "flushes")]


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("vertex_ms", "setup_ms", "bin_ms", "sort_ms", "cover_ms", "raster_ms", "clear_ms", "total_ms")] + \
               [("raster_launches", C.c_uint64), ("flushes", C.c_uint64)]


# every symbol include/swr.h declares; tests/test_abi.py checks the library exports all of them
EXPORTS = [
    "swr_abi_version", "swr_build_info", "swr_numerics_mode", "swr_set_transform_fma", "swr_get_transform_fma", "swr_set_pipelining", "swr_get_pipelining", "swr_last_error", "swr_create", "swr_destroy", "swr_resize", "swr_set_band", "swr_set_band_interleaved",
    "swr_bind_framebuffer", "swr_set_stream", "swr_clear_color", "swr_clear_depth", "swr_get_pixel",
    "swr_set_pixel", "swr_get_depth", "swr_set_depth", "swr_readback", "swr_readback_rgb", "swr_present_rgb_async", "swr_present_wait", "swr_flatten_rgb_device", "swr_flatten_rgb_device_async", "swr_replay_count", "swr_sync_count", "swr_host_register", "swr_host_unregister", "swr_upload", "swr_color_device_ptr",
    "swr_depth_device_ptr", "swr_texture_create", "swr_texture_destroy", "swr_texture_set_filter", "swr_texture_sample",
    "swr_mesh_create", "swr_mesh_destroy", "swr_set_state", "swr_initialize_tile_locks", "swr_render_mesh",
    "swr_render_mesh_arrays", "swr_mesh_bounds", "swr_is_sphere_in_frustum", "swr_render_mesh_culled", "swr_flush", "swr_sync", "swr_interpolate", "swr_get_stats", "swr_reset_stats",
    "swr_profile_enable", "swr_profile_get", "swr_profile_reset", "swr_profile_raster_samples", "swr_device_name", "swr_debug_counters", "swr_selftest_division",
]

_libs = {}


def load(name: str = None) -> C.CDLL:
    """Load libswr_hip.so (built by __graft_entry__.build()), or another in-tree build of the same source by file name
    (the numerics sensitivity builds, the test build).  Raises if it is absent.  The libraries are linked -Bsymbolic, so
    several builds can live in one process without binding to each other's kernels."""
    path = LIB_PATH if not name else os.path.join(_HERE, os.path.basename(name))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). softwarerenderer_amd has no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    P, I, F = C.c_void_p, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    sig = {
        "swr_abi_version": (I, []),
        "swr_build_info": (C.c_char_p, []),
        "swr_numerics_mode": (I, [C.POINTER(I), C.POINTER(I)]),
        "swr_set_transform_fma": (I, [P, I, I]),
        "swr_get_transform_fma": (I, [P, C.POINTER(I), C.POINTER(I)]),
        "swr_set_pipelining": (I, [P, I]),
        "swr_get_pipelining": (I, [P, C.POINTER(I)]),
        "swr_last_error": (C.c_char_p, [P]),
        "swr_create": (I, [I, C.POINTER(P)]),
        "swr_destroy": (None, [P]),
        "swr_resize": (I, [P, I, I]),
        "swr_set_band": (I, [P, I, I]),
        "swr_set_band_interleaved": (I, [P, I, I, I]),
        "swr_bind_framebuffer": (I, [P, P, P]),
        "swr_set_stream": (I, [P, P]),
        "swr_clear_color": (I, [P, fp]),
        "swr_clear_depth": (I, [P]),
        "swr_get_pixel": (I, [P, I, I, fp]),
        "swr_set_pixel": (I, [P, I, I, fp]),
        "swr_get_depth": (I, [P, I, I, fp]),
        "swr_set_depth": (I, [P, I, I, F]),
        "swr_readback": (I, [P, P, P]),
        "swr_readback_rgb": (I, [P, P]),
        "swr_present_rgb_async": (I, [P, P, C.POINTER(C.c_uint64)]),
        "swr_present_wait": (I, [P, C.c_uint64]),
        "swr_flatten_rgb_device": (I, [P, P]),
        "swr_flatten_rgb_device_async": (I, [P, P]),
        "swr_replay_count": (I, [P, C.POINTER(C.c_uint64)]),
        "swr_sync_count": (I, [P, C.POINTER(C.c_uint64)]),
        "swr_host_register": (I, [P, P, C.c_size_t]),
        "swr_host_unregister": (I, [P, P]),
        "swr_upload": (I, [P, P, P]),
        "swr_color_device_ptr": (I, [P, C.POINTER(P)]),
        "swr_depth_device_ptr": (I, [P, C.POINTER(P)]),
        "swr_texture_create": (I, [P, P, I, I, C.POINTER(P)]),
        "swr_texture_destroy": (I, [P, P]),
        "swr_texture_set_filter": (I, [P, P, I]),
        "swr_texture_sample": (I, [P, P, P, I, P]),
        "swr_mesh_create": (I, [P, P, I, P, I, C.POINTER(P)]),
        "swr_mesh_destroy": (I, [P, P]),
        "swr_set_state": (I, [P, F, F, I]),
        "swr_initialize_tile_locks": (I, [P, I, I]),
        "swr_render_mesh": (I, [P, P, fp, fp, fp, I, C.POINTER(Uniforms), P, I, I, I]),
        "swr_render_mesh_arrays": (I, [P, P, I, P, I, fp, fp, fp, I, C.POINTER(Uniforms), P, I, I, I]),
        "swr_mesh_bounds": (I, [P, P, fp]),
        "swr_is_sphere_in_frustum": (I, [P, fp, fp, fp, fp, C.POINTER(I)]),
        "swr_render_mesh_culled": (I, [P, P, fp, fp, fp, I, C.POINTER(Uniforms), P, I, I, I]),
        "swr_flush": (I, [P]),
        "swr_sync": (I, [P]),
        "swr_interpolate": (I, [P, P, P, I, I, P]),
        "swr_get_stats": (I, [P, C.POINTER(Stats)]),
        "swr_reset_stats": (I, [P]),
        "swr_profile_enable": (I, [P, I]),
        "swr_profile_get": (I, [P, C.POINTER(Profile)]),
        "swr_profile_reset": (I, [P]),
        "swr_profile_raster_samples": (I, [P, fp, I, C.POINTER(I)]),
        "swr_device_name": (I, [P, C.c_char_p, I]),
        "swr_debug_counters": (I, [P, C.POINTER(C.c_uint64)]),
        "swr_selftest_division": (I, [P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("SWR_ABLATE_LIB"):       # tools/ablate.py timing an OLDER build of the library (build_ab/): newer entry points may be absent
                continue
            raise ImportError(f"{path} does not export {name}: rebuild it (ABI {SWR_ABI_EXPECTED})")
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def check(ctx, rc: int, lib: C.CDLL = None) -> None:
    if rc != SWR_OK:
        msg = (lib or load()).swr_last_error(ctx)
        raise SwrError(rc, msg.decode() if msg else "")
