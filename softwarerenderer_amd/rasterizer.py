"""Host-side mirror of the reference's raster API over the HIP backend.

Same names, argument meaning and error behaviour as the C# reference so that callers and
tests read like the reference's own code (file:line under OCSYT/SoftwareRenderer):

  Rasterizer.RenderMesh / InitializeTileLocks / Interpolate, enums, statics   Rasterizer.cs:14-50,69,163,566
  Shaders.VertexInput / VertexOutput, shader delegates                        Shaders.cs:10-98
  Texture(pixels) / Width / Height / Sample / Dispose                          Texture.cs:31-68
  MainWindow.RenderWidth/RenderHeight/ColorBuffer/DepthBuffer/Get*/Set*/Clear* MainWindow.cs:25-31,378-436

C# delegates cannot cross the C ABI: a `ShaderProgram` (program id + uniform block + texture)
stands in for the (VertexShader, FragmentShader) delegate pair.  Everything below runs on the
GPU through libswr_hip.so; nothing here computes pixels on the host.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Optional

import numpy as np

from . import _native as N

# numpy view of Shaders.VertexInput (Shaders.cs:10-24): 12 consecutive float32 = 48 bytes
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("uv", "<f4", 2), ("normal", "<f4", 3), ("color", "<f4", 4)])
assert VERTEX_DTYPE.itemsize == 48 == C.sizeof(N.Vertex)


class DebugMode(enum.IntEnum):      # Rasterizer.cs:14-18
    None_ = 0
    Wireframe = 1


class BlendMode(enum.IntEnum):      # Rasterizer.cs:25-31
    None_ = 0
    Alpha = 1
    Additive = 2
    Multiply = 3


class DepthTest(enum.IntEnum):      # Rasterizer.cs:33-43
    Disabled = 0
    Less = 1
    LessEqual = 2
    Greater = 3
    GreaterEqual = 4
    Equal = 5
    NotEqual = 6
    Always = 7


class CullMode(enum.IntEnum):       # Rasterizer.cs:45-50
    None_ = 0
    Back = 1
    Front = 2


class Program(enum.IntEnum):        # built-in programs (include/swr.h)
    FlatColor = 0
    Gouraud = 1
    Dust2LambertFog = 2
    Phong4Point = 3
    DebugVaryings = 4               # build-defined: returns Normal / ScreenCoords / Barycentric (the varyings no other built-in reads)


def _f32(a, n=None):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    if n is not None and arr.size != n:
        raise ValueError(f"expected {n} floats, got {arr.size}")
    return arr


def _fptr(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_float))


def default_uniforms() -> N.Uniforms:
    """Uniform defaults of Renderer.cs:39-44 (FogStart 1, FogEnd 25, FogColor, LightDirection, LightColor)."""
    u = N.Uniforms()
    u.light_direction[:] = euler_to_direction(-45.0, -45.0, 0.0)
    u.light_color[:] = (1.0, 1.0, 1.0, 1.0)
    u.fog_color[:] = (1.0, 0.62, 0.5, 1.0)
    u.fog_start, u.fog_end = 1.0, 25.0
    u.shininess = 16.0
    return u


def euler_to_direction(pitch_deg: float, yaw_deg: float, roll_deg: float):
    """Renderer.EulerToDirection (Renderer.cs:967-972): -UnitZ rotated by CreateFromYawPitchRoll(yaw, pitch, roll),
    normalised.  Host-side input generator (float32); roll does not move the forward axis."""
    p = np.float32(pitch_deg) * np.float32(np.pi) / np.float32(180.0)
    y = np.float32(yaw_deg) * np.float32(np.pi) / np.float32(180.0)
    v = np.array([-np.sin(y) * np.cos(p), np.sin(p), -np.cos(y) * np.cos(p)], dtype=np.float32)
    v = v / np.float32(np.sqrt(np.float32(v @ v)))
    return tuple(float(t) for t in v)


class Device:
    """One swr_context = one GPU (one process per GPU)."""

    def __init__(self, device_id: int = 0, lib: str = None):
        """`lib`: file name of another in-tree build of the backend (e.g. "libswr_hip_fma.so"); default = the product library."""
        self._lib = N.load(lib)
        self._ctx = C.c_void_p()
        rc = self._lib.swr_create(int(device_id), C.byref(self._ctx))
        if rc != N.SWR_OK:
            msg = self._lib.swr_last_error(None)
            raise N.SwrError(rc, msg.decode() if msg else "swr_create failed")

    def close(self):
        if self._ctx:
            self._lib.swr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        N.check(self._ctx, rc, self._lib)

    @property
    def name(self) -> str:
        buf = C.create_string_buffer(256)
        self._ck(self._lib.swr_device_name(self._ctx, buf, 256))
        return buf.value.decode()

    def pin(self, array: np.ndarray):
        """Page-lock a long-lived host array (swr_host_register): read-backs into it then DMA at PCIe rate."""
        self._ck(self._lib.swr_host_register(self._ctx, C.c_void_p(array.ctypes.data), C.c_size_t(array.nbytes)))

    def unpin(self, array: np.ndarray):
        self._ck(self._lib.swr_host_unregister(self._ctx, C.c_void_p(array.ctypes.data)))

    def stats(self) -> dict:
        s = N.Stats()
        self._ck(self._lib.swr_get_stats(self._ctx, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in N.Stats._fields_}

    def reset_stats(self):
        self._ck(self._lib.swr_reset_stats(self._ctx))

    def profile_enable(self, on):
        """False/0 off, True/1 events around every stage, 2 around the raster kernel only, 3 around the raster kernel of every
        4th flush (an event pair costs about 10 us of stream time)."""
        self._ck(self._lib.swr_profile_enable(self._ctx, int(on)))

    def profile(self) -> dict:
        p = N.Profile()
        self._ck(self._lib.swr_profile_get(self._ctx, C.byref(p)))
        return {n: (float(getattr(p, n)) if t is C.c_double else int(getattr(p, n))) for n, t in N.Profile._fields_}

    def raster_samples(self) -> np.ndarray:
        """Duration (ms) of every raster-kernel launch that carried an event pair since profile_reset (swr_profile_raster_samples)."""
        n = C.c_int(0)
        self._ck(self._lib.swr_profile_raster_samples(self._ctx, None, 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.float32)
        self._ck(self._lib.swr_profile_raster_samples(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), int(out.size), C.byref(n)))
        return out[:int(n.value)]

    def profile_reset(self):
        self._ck(self._lib.swr_profile_reset(self._ctx))

    def selftest_division(self, samples: int = 1 << 30, seed: int = 1) -> dict:
        """swr_selftest_division: the kernels' exact-division / sqrt cores against the compiler's IEEE `/` and sqrtf."""
        out = (C.c_uint64 * 8)()
        self._ck(self._lib.swr_selftest_division(self._ctx, C.c_uint64(samples), C.c_uint64(seed), out))
        keys = ("divisions", "division_mismatches", "sqrts", "sqrt_mismatches", "bad_n_bits", "bad_d_bits", "bad_got_bits", "bad_want_bits")
        return dict(zip(keys, (int(v) for v in out)))

    def replay_count(self) -> int:
        out = C.c_uint64(0)
        self._ck(self._lib.swr_replay_count(self._ctx, C.byref(out)))
        return int(out.value)

    def sync_count(self) -> int:
        """Times an entry point has made the host wait for the stream (swr_sync_count): a steady-state frame loop adds none."""
        out = C.c_uint64(0)
        self._ck(self._lib.swr_sync_count(self._ctx, C.byref(out)))
        return int(out.value)

    def build_info(self) -> str:
        """`hipcc=...; csrc_sha256=...; fma=..; dot=..` of the loaded library (swr_build_info)."""
        return self._lib.swr_build_info().decode()

    def numerics_mode(self):
        """(fma, dot_order) the loaded library was compiled with (SWR_NUMERICS_FMA, SWR_DOT_PAIRWISE)."""
        f, d = C.c_int(0), C.c_int(0)
        self._ck(self._lib.swr_numerics_mode(C.byref(f), C.byref(d)))
        return int(f.value), int(d.value)

    def set_transform_fma(self, transform_fused: bool, transform_normal_fused: bool):
        """The run-time half of the System.Numerics model (swr_set_transform_fma): do Vector4.Transform (Renderer.cs:832-834) /
        Vector3.TransformNormal (:835) fuse their multiply-adds?  Default = the library's compile-time fma for both."""
        self._ck(self._lib.swr_set_transform_fma(self._ctx, int(bool(transform_fused)), int(bool(transform_normal_fused))))

    def transform_fma(self):
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self._lib.swr_get_transform_fma(self._ctx, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def set_pipelining(self, mode: int):
        """swr_set_pipelining: 0 = one stream (kernel timings are quoted on this), 1 = the front end of flush N+1 beside the raster
        kernel of flush N, every asynchronous batch (default), 2 = the same for small frames / batches only (<= 2^15 tiles or <= 2^17 triangles)."""
        self._ck(self._lib.swr_set_pipelining(self._ctx, int(mode)))

    def pipelining(self) -> int:
        m = C.c_int(0)
        self._ck(self._lib.swr_get_pipelining(self._ctx, C.byref(m)))
        return int(m.value)

    def set_stream(self, hip_stream: int):
        self._ck(self._lib.swr_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def flush(self):
        self._ck(self._lib.swr_flush(self._ctx))

    def sync(self):
        self._ck(self._lib.swr_sync(self._ctx))


class Texture:
    """Texture(Image<Rgba32>), Texture.cs:31-68: RGBA8 pixels, nearest sampling with wrap."""

    def __init__(self, device: Device, pixels_rgba8: np.ndarray):
        px = np.ascontiguousarray(pixels_rgba8, dtype=np.uint8)
        if px.ndim != 3 or px.shape[2] != 4:
            raise ValueError("pixels must be (height, width, 4) uint8")
        self._dev = device
        self.Height, self.Width = int(px.shape[0]), int(px.shape[1])
        self._h = C.c_void_p()
        device._ck(device._lib.swr_texture_create(device._ctx, px.ctypes.data, self.Width, self.Height, C.byref(self._h)))

    def Sample(self, uv) -> np.ndarray:
        """Texture.Sample, Texture.cs:43-63 (batched: uv of shape (..., 2) -> (..., 4)); runs on the GPU."""
        a = np.ascontiguousarray(np.asarray(uv, dtype=np.float32))
        flat = a.reshape(-1, 2)
        out = np.empty((flat.shape[0], 4), dtype=np.float32)
        self._dev._ck(self._dev._lib.swr_texture_sample(self._dev._ctx, self._h, flat.ctypes.data, flat.shape[0], out.ctypes.data))
        return out.reshape(a.shape[:-1] + (4,))

    def SetBilinear(self, on: bool):
        """Build-defined extension (the reference samples nearest): bilinear filter with wrap for later draws."""
        self._dev._ck(self._dev._lib.swr_texture_set_filter(self._dev._ctx, self._h, 1 if on else 0))

    def Dispose(self):
        if self._h:
            self._dev._ck(self._dev._lib.swr_texture_destroy(self._dev._ctx, self._h))
            self._h = C.c_void_p()


class VertexShader:
    def __init__(self, program: "ShaderProgram"):
        self.program = program


class FragmentShader:
    def __init__(self, program: "ShaderProgram"):
        self.program = program


class ShaderProgram:
    """Stands in for the reference's (VertexShader, FragmentShader) delegate pair (Shaders.cs:97-98)."""

    def __init__(self, program: Program, uniforms: Optional[N.Uniforms] = None, texture: Optional[Texture] = None):
        self.program = Program(program)
        self.uniforms = uniforms if uniforms is not None else default_uniforms()
        self.texture = texture
        self.VertexShader = VertexShader(self)
        self.FragmentShader = FragmentShader(self)


class Shaders:
    VertexInput = VERTEX_DTYPE      # Shaders.cs:10-24
    Program = Program

    @staticmethod
    def FlatColor():
        return ShaderProgram(Program.FlatColor)

    @staticmethod
    def Gouraud():
        return ShaderProgram(Program.Gouraud)

    @staticmethod
    def Dust2LambertFog(uniforms=None, texture=None):     # Renderer.cs:830-860
        return ShaderProgram(Program.Dust2LambertFog, uniforms, texture)

    @staticmethod
    def DebugVaryings():
        return ShaderProgram(Program.DebugVaryings)

    @staticmethod
    def Phong4Point(uniforms, texture=None):
        return ShaderProgram(Program.Phong4Point, uniforms, texture)


class MainWindow:
    """Framebuffer part of MainWindow (MainWindow.cs:25-31,378-436); buffers live in HBM."""

    def __init__(self, device: Device, render_width: int = 800, render_height: int = 600):
        self._dev = device
        self.RenderWidth = 0
        self.RenderHeight = 0
        self._band = None
        self._bound = None
        self.Resize(render_width, render_height)

    def _activate(self):
        """A context owns ONE framebuffer (one MainWindow in the reference).  Several MainWindow objects on one Device
        take turns: the one being used re-applies its geometry first; its previous pixels are then undefined.  The window that
        is already active issues only the call for what changed (BindFramebuffer -> swr_bind_framebuffer alone: no flush-and-wait,
        fragment counters keep accumulating); the C side also ignores a resize / band that changes nothing."""
        if getattr(self._dev, "_active_window", None) is self:
            return
        self._apply_size()
        self._apply_band()
        self._apply_binding()
        self._dev._active_window = self

    def _apply_size(self):
        self._dev._ck(self._dev._lib.swr_resize(self._dev._ctx, self.RenderWidth, self.RenderHeight))

    def _apply_band(self):
        lib, ctx = self._dev._lib, self._dev._ctx
        if self._band is not None and len(self._band) == 3:
            self._dev._ck(lib.swr_set_band_interleaved(ctx, *self._band))
        else:
            b = self._band if self._band is not None else (-1, -1)
            self._dev._ck(lib.swr_set_band(ctx, b[0], b[1]))

    def _apply_binding(self):
        cb = self._bound if self._bound is not None else (None, None)
        self._dev._ck(self._dev._lib.swr_bind_framebuffer(self._dev._ctx, C.c_void_p(cb[0]) if cb[0] else None,
                                                          C.c_void_p(cb[1]) if cb[1] else None))

    def _is_active(self):
        return getattr(self._dev, "_active_window", None) is self

    def Resize(self, width: int, height: int):          # HandleResize, MainWindow.cs:320-321
        self.RenderWidth, self.RenderHeight = int(width), int(height)
        if self._is_active():
            self._apply_size()
        else:
            self._activate()

    def SetBand(self, first_tile_row: int, n_tile_rows: int):
        self._band = (int(first_tile_row), int(n_tile_rows)) if first_tile_row >= 0 and n_tile_rows >= 0 else None
        if self._is_active():
            self._apply_band()
        else:
            self._activate()

    def SetBandInterleaved(self, rank: int, world: int, stripe_tile_rows: int):
        """Interleaved stripes instead of one contiguous band: stripe s (stripe_tile_rows tile rows) belongs to rank s % world;
        the window then holds this rank's stripes one after the other (multigpu.stripe_rows gives their frame rows)."""
        self._band = (int(rank), int(world), int(stripe_tile_rows))
        if self._is_active():
            self._apply_band()
        else:
            self._activate()

    def band_pixel_rows(self):
        tiles_y = (self.RenderHeight + 15) // 16
        if self._band is None:
            return 0, max(self.RenderHeight, 0)
        if len(self._band) == 3:
            from . import multigpu
            return 0, len(multigpu.stripe_rows(self.RenderHeight, self._band[1], self._band[2])[self._band[0]])
        t0 = min(max(self._band[0], 0), tiles_y)
        t1 = min(t0 + max(self._band[1], 0), tiles_y)
        y0, y1 = t0 * 16, min(self.RenderHeight, t1 * 16)
        return y0, max(0, y1 - y0)

    def BindFramebuffer(self, color_ptr: int, depth_ptr: int):
        """swr_bind_framebuffer: does not wait for the GPU.  Buffers bound earlier stay referenced until the next validating call
        (Device.sync / read-back / stats): see the lifetime rule in include/swr.h."""
        self._bound = (int(color_ptr), int(depth_ptr)) if color_ptr and depth_ptr else None
        if self._is_active():
            self._apply_binding()
        else:
            self._activate()

    def ClearColorBuffer(self, clear_color):             # MainWindow.cs:400-407
        self._activate()
        c = _f32(clear_color, 4)
        self._dev._ck(self._dev._lib.swr_clear_color(self._dev._ctx, _fptr(c)))

    def ClearDepthBuffer(self):                          # MainWindow.cs:429-436
        self._activate()
        self._dev._ck(self._dev._lib.swr_clear_depth(self._dev._ctx))

    def _read(self, want_color=True, want_depth=True):
        self._activate()
        _, rows = self.band_pixel_rows()
        w = max(self.RenderWidth, 0)
        col = np.empty((rows, w, 4), dtype=np.float32) if want_color else None
        dep = np.empty((rows, w), dtype=np.float32) if want_depth else None
        self._dev._ck(self._dev._lib.swr_readback(self._dev._ctx, col.ctypes.data if want_color else None,
                                                  dep.ctypes.data if want_depth else None))
        return col, dep

    @property
    def ColorBuffer(self) -> np.ndarray:                 # Vector4[W*H], idx = y*W + x
        return self._read(True, False)[0]

    @property
    def DepthBuffer(self) -> np.ndarray:
        return self._read(False, True)[1]

    def FlatColorBuffer(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """The Vector3[] `flatColorBuffer` of MainWindow.OnRender (MainWindow.cs:234-240), flattened on the GPU.
        `out`: a caller-owned (rows, W, 3) float32 array to fill -- page-lock it once with Device.pin() and the copy runs
        at PCIe rate."""
        _, rows = self.band_pixel_rows()
        shape = (rows, max(self.RenderWidth, 0), 3)
        if out is None:
            out = np.empty(shape, dtype=np.float32)
        elif out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {shape}")
        self._activate()
        if out.size:
            self._dev._ck(self._dev._lib.swr_readback_rgb(self._dev._ctx, out.ctypes.data))
        return out

    def PresentAsync(self, out: np.ndarray) -> int:
        """Asynchronous FlatColorBuffer (swr_present_rgb_async): flatten on the GPU behind the recorded draws, copy into `out`
        (page-lock it with Device.pin()) on the context's copy stream, return a ticket at once.  The next frame renders while this
        one crosses PCIe; `out` must stay untouched until PresentWait(ticket)."""
        _, rows = self.band_pixel_rows()
        shape = (rows, max(self.RenderWidth, 0), 3)
        if out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {shape}")
        self._activate()
        t = C.c_uint64(0)
        self._dev._ck(self._dev._lib.swr_present_rgb_async(self._dev._ctx, C.c_void_p(out.ctypes.data), C.byref(t)))
        return int(t.value)

    def PresentWait(self, ticket: int) -> bool:
        """Blocks until the present `ticket` has landed in its array.  True = the array holds the frame; False = a batch was replayed
        meanwhile (SWR_STALE: only while the pair buffers are still growing), present that frame again."""
        rc = self._dev._lib.swr_present_wait(self._dev._ctx, C.c_uint64(ticket))
        if rc == N.SWR_STALE:
            return False
        self._dev._ck(rc)
        return True

    def FlattenTo(self, device_ptr: int):
        """FlatColorBuffer into caller-owned DEVICE memory (band rows x W x 3 floats); completes with Device.sync()."""
        self._activate()
        self._dev._ck(self._dev._lib.swr_flatten_rgb_device(self._dev._ctx, C.c_void_p(device_ptr)))

    def FlattenToAsync(self, device_ptr: int):
        """FlattenTo without the validation sync (swr_flatten_rgb_device_async): for frame loops that chain render -> flatten
        -> gather in stream order and validate later with Device.sync() + Device.replay_count()."""
        self._activate()
        self._dev._ck(self._dev._lib.swr_flatten_rgb_device_async(self._dev._ctx, C.c_void_p(device_ptr)))

    def Upload(self, color=None, depth=None):
        self._activate()
        c = np.ascontiguousarray(color, dtype=np.float32) if color is not None else None
        d = np.ascontiguousarray(depth, dtype=np.float32) if depth is not None else None
        self._dev._ck(self._dev._lib.swr_upload(self._dev._ctx, c.ctypes.data if c is not None else None,
                                                d.ctypes.data if d is not None else None))

    def GetPixel(self, x: int, y: int) -> np.ndarray:    # MainWindow.cs:391-398
        self._activate()
        out = np.zeros(4, dtype=np.float32)
        self._dev._ck(self._dev._lib.swr_get_pixel(self._dev._ctx, int(x), int(y), _fptr(out)))
        return out

    def SetPixel(self, x: int, y: int, color):           # MainWindow.cs:382-388
        self._activate()
        c = _f32(color, 4)
        self._dev._ck(self._dev._lib.swr_set_pixel(self._dev._ctx, int(x), int(y), _fptr(c)))

    def GetDepth(self, x: int, y: int) -> float:         # MainWindow.cs:420-426
        self._activate()
        out = C.c_float(0.0)
        self._dev._ck(self._dev._lib.swr_get_depth(self._dev._ctx, int(x), int(y), C.byref(out)))
        return float(np.float32(out.value))

    def SetDepth(self, x: int, y: int, depth: float):    # MainWindow.cs:411-417
        self._activate()
        self._dev._ck(self._dev._lib.swr_set_depth(self._dev._ctx, int(x), int(y), float(depth)))


class Mesh:
    """Retained mesh: Mesh.Vertices / Mesh.Indices (ModelLoader.cs:45-47) uploaded once."""

    def __init__(self, device: Device, vertices: np.ndarray, indices: np.ndarray):
        v = as_vertex_array(vertices)
        i = np.ascontiguousarray(indices, dtype=np.uint16).reshape(-1)
        self._dev = device
        self.n_vertices, self.n_indices = int(v.shape[0]), int(i.shape[0])
        self._h = C.c_void_p()
        device._ck(device._lib.swr_mesh_create(device._ctx, v.ctypes.data, self.n_vertices, i.ctypes.data, self.n_indices, C.byref(self._h)))

    @property
    def SphereBounds(self) -> np.ndarray:
        """Mesh.SphereBounds = FrustumCuller.CalculateBoundingSphere(vertices) (ModelLoader.cs:291): {cx, cy, cz, r}, on the GPU."""
        out = np.zeros(4, dtype=np.float32)
        self._dev._ck(self._dev._lib.swr_mesh_bounds(self._dev._ctx, self._h, _fptr(out)))
        return out

    def Dispose(self):
        if self._h:
            self._dev._ck(self._dev._lib.swr_mesh_destroy(self._dev._ctx, self._h))
            self._h = C.c_void_p()


class FrustumCuller:
    """public static class FrustumCuller (FrustumCuller.cs:57): the two members the frame loop uses."""

    @staticmethod
    def CalculateBoundingSphere(mesh: "Mesh") -> np.ndarray:          # FrustumCuller.cs:59-151
        return mesh.SphereBounds

    @staticmethod
    def IsSphereInFrustum(window: "MainWindow", bounds, modelMatrix, viewMatrix, projectionMatrix) -> bool:   # :201-218
        b, m, v, p = _f32(bounds, 4), _f32(modelMatrix, 16), _f32(viewMatrix, 16), _f32(projectionMatrix, 16)
        out = C.c_int(0)
        dev = window._dev
        dev._ck(dev._lib.swr_is_sphere_in_frustum(dev._ctx, _fptr(b), _fptr(m), _fptr(v), _fptr(p), C.byref(out)))
        return bool(out.value)


def as_vertex_array(vertices) -> np.ndarray:
    v = np.asarray(vertices)
    if v.dtype == VERTEX_DTYPE:
        return np.ascontiguousarray(v).reshape(-1)
    v = np.ascontiguousarray(v, dtype=np.float32)
    if v.ndim != 2 or v.shape[1] != 12:
        raise ValueError("vertices must have dtype VERTEX_DTYPE or shape (n, 12) float32")
    return v.view(VERTEX_DTYPE).reshape(-1)


class Rasterizer:
    """public static class Rasterizer (Rasterizer.cs:12): statics + enums + RenderMesh."""

    DebugMode = DebugMode
    BlendMode = BlendMode
    DepthTest = DepthTest
    CullMode = CullMode

    NearClip = 0.1                       # Rasterizer.cs:20
    FarClip = 1000.0                     # Rasterizer.cs:21
    RenderDebugMode = DebugMode.None_    # Rasterizer.cs:22

    @staticmethod
    def InitializeTileLocks(window: MainWindow, width: int, height: int):
        """Rasterizer.cs:69-93; non-positive sizes raise (C#: ArgumentException)."""
        try:
            window._dev._ck(window._dev._lib.swr_initialize_tile_locks(window._dev._ctx, int(width), int(height)))
        except N.SwrError as e:
            if e.code == N.SWR_ERR_INVALID_ARG:
                raise ValueError("Width and height must be positive non-zero values.") from e
            raise

    @classmethod
    def RenderMesh(cls, window: MainWindow, vertices, indices, model, view, projection,
                   vertexShader: VertexShader, fragmentShader: FragmentShader,
                   cullMode: CullMode = CullMode.Back, depthTest: DepthTest = DepthTest.LessEqual,
                   blendMode: BlendMode = BlendMode.Alpha, frustumCull: bool = False):
        """Rasterizer.RenderMesh, Rasterizer.cs:163-174.  frustumCull=True (retained meshes only) prepends the
        reference's `if (!FrustumCuller.IsSphereInFrustum(mesh.SphereBounds, ...)) return;` (Renderer.cs:446), evaluated on the GPU.  `vertices` may be a retained `Mesh`
        (then `indices` is ignored) or the VertexInput[] / ushort[] arrays of the C# signature."""
        prog = fragmentShader.program
        if vertexShader.program is not prog:
            raise ValueError("vertexShader and fragmentShader must come from the same ShaderProgram")
        window._activate()
        dev = window._dev
        dev._ck(dev._lib.swr_set_state(dev._ctx, float(cls.NearClip), float(cls.FarClip), int(cls.RenderDebugMode)))
        m, v, p = _f32(model, 16), _f32(view, 16), _f32(projection, 16)
        tex = prog.texture._h if prog.texture is not None else None
        if isinstance(vertices, Mesh):
            fn = dev._lib.swr_render_mesh_culled if frustumCull else dev._lib.swr_render_mesh
            rc = fn(dev._ctx, vertices._h, _fptr(m), _fptr(v), _fptr(p), int(prog.program),
                                          C.byref(prog.uniforms), tex, int(cullMode), int(depthTest), int(blendMode))
        else:
            va = as_vertex_array(vertices)
            ia = np.ascontiguousarray(indices, dtype=np.uint16).reshape(-1)
            rc = dev._lib.swr_render_mesh_arrays(dev._ctx, va.ctypes.data, int(va.shape[0]), ia.ctypes.data, int(ia.shape[0]),
                                                 _fptr(m), _fptr(v), _fptr(p), int(prog.program), C.byref(prog.uniforms), tex,
                                                 int(cullMode), int(depthTest), int(blendMode))
        if rc == N.SWR_ERR_INVALID_ARG:
            msg = dev._lib.swr_last_error(dev._ctx).decode()
            if "index out of range" in msg:
                raise IndexError(msg)      # C#: IndexOutOfRangeException
        dev._ck(rc)

    @staticmethod
    def Interpolate(window: MainWindow, a, b, c, w, interpolate: bool = True) -> np.ndarray:
        """Rasterizer.Interpolate (public, Rasterizer.cs:566-640), batched over n weight triples.
        a, b, c: 20-float vertex records {clip4,color4,uv2,normal3,screen2,worldNormal3,pad2};
        returns (n, 24): {clip4,color4,uv2,normal3,screen2,worldNormal3,bary3,pad3}."""
        verts = np.concatenate([_f32(a, 20), _f32(b, 20), _f32(c, 20)])
        wts = np.ascontiguousarray(np.asarray(w, dtype=np.float32).reshape(-1, 3))
        out = np.empty((wts.shape[0], 24), dtype=np.float32)
        dev = window._dev
        dev._ck(dev._lib.swr_interpolate(dev._ctx, verts.ctypes.data, wts.ctypes.data, wts.shape[0], 1 if interpolate else 0, out.ctypes.data))
        return out
