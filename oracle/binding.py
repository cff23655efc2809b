"""ctypes binding of oracle/liboswr.so -- the CPU restatement of the reference's raster path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by softwarerenderer_amd.  PARITY UNPINNED: see oracle/swr_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("triangles_in", "triangles_setup", "triangles_clipped",
                                          "fragments_tested", "fragments_shaded", "fragments_written")]


class OVertexOutput(C.Structure):     # oswr_vertex_output
    _fields_ = [("clip", C.c_float * 4), ("color", C.c_float * 4), ("texcoord", C.c_float * 2), ("normal", C.c_float * 3),
                ("screen", C.c_float * 2), ("world_normal", C.c_float * 3), ("world_pos", C.c_float * 4),
                ("has_data", C.c_int), ("interpolate", C.c_int), ("barycentric", C.c_float * 3)]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    so = os.path.join(_HERE, "liboswr_dpps.so")          # the last target of the Makefile's `all`
    src = os.path.join(_HERE, "swr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)


_libs = {}


VARIANTS = {"": "liboswr.so", "fma": "liboswr_fma.so", "dotpw": "liboswr_dotpw.so", "fma_dotpw": "liboswr_fma_dotpw.so",
            "dpps": "liboswr_dpps.so", "fma_dpps": "liboswr_fma_dpps.so",      # System.Numerics sensitivity builds (oracle/Makefile)
            "deadcount": "liboswr_deadcount.so"}      # tools/dead_fragments.py builds it (-DOSWR_DEAD_COUNT); not a checker


def load(fma: bool = False, variant: str = None) -> C.CDLL:
    name = VARIANTS[variant] if variant is not None else ("liboswr_fma.so" if fma else "liboswr.so")
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    P, I, F = C.c_void_p, C.c_int, C.c_float
    lib.oswr_create.restype = P; lib.oswr_create.argtypes = [I, I]
    lib.oswr_destroy.restype = None; lib.oswr_destroy.argtypes = [P]
    lib.oswr_resize.restype = I; lib.oswr_resize.argtypes = [P, I, I]
    lib.oswr_set_state.restype = None; lib.oswr_set_state.argtypes = [P, F, F, I]
    lib.oswr_set_threads.restype = None; lib.oswr_set_threads.argtypes = [P, I]
    lib.oswr_clear_color.restype = None; lib.oswr_clear_color.argtypes = [P, P]
    lib.oswr_clear_depth.restype = None; lib.oswr_clear_depth.argtypes = [P]
    lib.oswr_color_buffer.restype = C.POINTER(C.c_float); lib.oswr_color_buffer.argtypes = [P]
    lib.oswr_depth_buffer.restype = C.POINTER(C.c_float); lib.oswr_depth_buffer.argtypes = [P]
    lib.oswr_render_mesh.restype = I
    lib.oswr_render_mesh.argtypes = [P, P, I, P, I, P, P, P, I, P, P, I, I, I, I, I]
    lib.oswr_get_stats.restype = None; lib.oswr_get_stats.argtypes = [P, C.POINTER(OStats)]
    lib.oswr_reset_stats.restype = None; lib.oswr_reset_stats.argtypes = [P]
    lib.oswr_texture_sample.restype = None; lib.oswr_texture_sample.argtypes = [P, I, I, P, P]
    lib.oswr_texture_sample_bilinear.restype = None; lib.oswr_texture_sample_bilinear.argtypes = [P, I, I, P, P]
    lib.oswr_set_texture_filter.restype = None; lib.oswr_set_texture_filter.argtypes = [P, I]
    lib.oswr_interpolate.restype = None
    lib.oswr_interpolate.argtypes = [C.POINTER(OVertexOutput)] * 3 + [F, F, F, I, C.POINTER(OVertexOutput)]
    lib.oswr_lerp.restype = None
    lib.oswr_lerp.argtypes = [C.POINTER(OVertexOutput)] * 2 + [F, I, C.POINTER(OVertexOutput)]
    lib.oswr_vertex_shader.restype = None
    lib.oswr_vertex_shader.argtypes = [P, P, P, P, I, C.POINTER(OVertexOutput)]
    lib.oswr_blend.restype = None; lib.oswr_blend.argtypes = [P, P, I, P]
    lib.oswr_depth_func.restype = I; lib.oswr_depth_func.argtypes = [I, F, F]
    lib.oswr_edge_function.restype = F; lib.oswr_edge_function.argtypes = [P, P, P]
    lib.oswr_numerics_fma.restype = I; lib.oswr_numerics_fma.argtypes = []
    lib.oswr_dot_pairwise.restype = I; lib.oswr_dot_pairwise.argtypes = []
    lib.oswr_set_transform_fma.restype = None; lib.oswr_set_transform_fma.argtypes = [I, I]
    lib.oswr_get_transform_fma.restype = None; lib.oswr_get_transform_fma.argtypes = [C.POINTER(I), C.POINTER(I)]
    lib.oswr_bounding_sphere.restype = None; lib.oswr_bounding_sphere.argtypes = [P, I, P]
    lib.oswr_is_sphere_in_frustum.restype = I; lib.oswr_is_sphere_in_frustum.argtypes = [P, P, P, P]
    _libs[name] = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


class OracleRenderer:
    """Renders a softwarerenderer_amd.scenes.Scene (or its draws one by one) on the CPU oracle."""

    def __init__(self, width: int, height: int, threads: int = 1, fma: bool = False, variant: str = None, transform_fma=None):
        """transform_fma = (Transform fused?, TransformNormal fused?): the run-time half of the System.Numerics model (one setting per
        library instance, re-applied before every draw of this renderer); None = the library's compile-time default for both"""
        self.lib = load(fma, variant)
        d = bool(self.lib.oswr_numerics_fma())
        self.transform_fma = (d, d) if transform_fma is None else (bool(transform_fma[0]), bool(transform_fma[1]))
        self.lib.oswr_set_transform_fma(*[int(x) for x in self.transform_fma])
        self.ctx = self.lib.oswr_create(int(width), int(height))
        if not self.ctx:
            raise MemoryError("oswr_create failed")
        self.width, self.height = int(width), int(height)
        self.lib.oswr_set_threads(self.ctx, int(threads))

    def close(self):
        if self.ctx:
            self.lib.oswr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def set_state(self, near_clip=0.1, far_clip=1000.0, debug_mode=0):
        self.lib.oswr_set_state(self.ctx, float(near_clip), float(far_clip), int(debug_mode))

    def clear_color(self, rgba):
        c = _f32(rgba)
        self.lib.oswr_clear_color(self.ctx, c.ctypes.data)

    def clear_depth(self):
        self.lib.oswr_clear_depth(self.ctx)

    def upload(self, color=None, depth=None):
        if color is not None:
            self.color[...] = np.asarray(color, dtype=np.float32).reshape(self.height, self.width, 4)
        if depth is not None:
            self.depth[...] = np.asarray(depth, dtype=np.float32).reshape(self.height, self.width)

    @property
    def color(self) -> np.ndarray:      # live view of the oracle's ColorBuffer
        n = self.width * self.height * 4
        if n == 0:
            return np.zeros((self.height, self.width, 4), dtype=np.float32)
        return np.ctypeslib.as_array(self.lib.oswr_color_buffer(self.ctx), shape=(n,)).reshape(self.height, self.width, 4)

    @property
    def depth(self) -> np.ndarray:
        n = self.width * self.height
        if n == 0:
            return np.zeros((self.height, self.width), dtype=np.float32)
        return np.ctypeslib.as_array(self.lib.oswr_depth_buffer(self.ctx), shape=(n,)).reshape(self.height, self.width)

    def render_mesh(self, vertices, indices, model, view, projection, program, uniforms=None, texture=None,
                    cull=1, depth_test=2, blend=1) -> int:
        self.lib.oswr_set_transform_fma(*[int(x) for x in self.transform_fma])      # (per library instance: another renderer may have set it)
        v = np.ascontiguousarray(vertices)
        i = np.ascontiguousarray(indices, dtype=np.uint16).reshape(-1)
        m, vw, p = _f32(model), _f32(view), _f32(projection)
        tex_ptr, tw, th = None, 0, 0
        if texture is not None:
            t = np.ascontiguousarray(texture, dtype=np.uint8)
            tex_ptr, tw, th = t.ctypes.data, int(t.shape[1]), int(t.shape[0])
        uptr = C.cast(C.byref(uniforms), C.c_void_p) if uniforms is not None else None
        return self.lib.oswr_render_mesh(self.ctx, v.ctypes.data, int(v.shape[0]), i.ctypes.data, int(i.shape[0]),
                                         m.ctypes.data, vw.ctypes.data, p.ctypes.data, int(program), uptr,
                                         tex_ptr, tw, th, int(cull), int(depth_test), int(blend))

    def set_texture_filter(self, bilinear: bool):
        self.lib.oswr_set_texture_filter(self.ctx, 1 if bilinear else 0)

    def render_scene(self, scene, debug_mode=0):
        """RenderScene of Renderer.cs:404-419 for a softwarerenderer_amd.scenes.Scene; returns copies (color, depth)."""
        self.set_state(scene.near_clip, scene.far_clip, debug_mode)
        self.set_texture_filter(getattr(scene, "bilinear", False))
        if scene.clear_depth:
            self.clear_depth()
        if scene.clear_color is not None:
            self.clear_color(scene.clear_color)
        for d in scene.draws:
            tex = scene.textures[d.texture] if d.texture is not None else None
            rc = self.render_mesh(d.vertices, d.indices, d.model, d.view, d.projection, int(d.program), d.uniforms, tex,
                                  int(d.cull), int(d.depth_test), int(d.blend))
            if rc != 0:
                raise IndexError("oracle: index out of range")
        return self.color.copy(), self.depth.copy()

    def stats(self) -> dict:
        s = OStats()
        self.lib.oswr_get_stats(self.ctx, C.byref(s))
        return {n: int(getattr(s, n)) for n, _ in OStats._fields_}

    def reset_stats(self):
        self.lib.oswr_reset_stats(self.ctx)
