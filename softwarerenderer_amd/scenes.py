"""Synthetic scenes of BASELINE.json's configurations plus edge-case scenes.

A Scene is backend-neutral data (numpy arrays + enums): the HIP backend renders it through
`render_scene` below, the tests render the same object through the CPU oracle.  All generators
are seeded (numpy default_rng) and produce float32 inputs, so CPU and GPU see identical bytes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import hostmath as hm
from . import _native as N
from .rasterizer import (VERTEX_DTYPE, BlendMode, CullMode, DepthTest, Device, MainWindow, Mesh, Program,
                         Rasterizer, ShaderProgram, Texture, default_uniforms)

CLEAR_COLOR = (0.9137, 0.7098, 0.6588, 1.0)      # Renderer.cs:45,413


@dataclass
class Draw:
    vertices: np.ndarray                 # VERTEX_DTYPE
    indices: np.ndarray                  # uint16, 3 per triangle
    model: np.ndarray
    view: np.ndarray
    projection: np.ndarray
    program: Program = Program.Gouraud
    uniforms: Optional[N.Uniforms] = None
    texture: Optional[int] = None        # index into Scene.textures
    cull: CullMode = CullMode.Back       # defaults of Rasterizer.cs:172-174
    depth_test: DepthTest = DepthTest.LessEqual
    blend: BlendMode = BlendMode.Alpha


@dataclass
class Scene:
    name: str
    width: int
    height: int
    draws: List[Draw] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)    # (h, w, 4) uint8
    clear_color: Optional[tuple] = CLEAR_COLOR                  # None = keep buffer contents
    clear_depth: bool = True
    near_clip: float = 0.1
    far_clip: float = 1000.0
    bilinear: bool = False                                       # build-defined texture filter extension (reference = nearest)

    @property
    def n_triangles(self) -> int:
        return int(sum(d.indices.size // 3 for d in self.draws))


def make_vertices(pos, uv=None, normal=None, color=None) -> np.ndarray:
    pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    n = pos.shape[0]
    v = np.zeros(n, dtype=VERTEX_DTYPE)
    v["position"] = pos
    v["uv"] = 0.0 if uv is None else np.asarray(uv, dtype=np.float32).reshape(n, 2)
    v["normal"] = (0.0, 0.0, 1.0) if normal is None else np.asarray(normal, dtype=np.float32).reshape(n, 3)
    v["color"] = (1.0, 1.0, 1.0, 1.0) if color is None else np.asarray(color, dtype=np.float32).reshape(n, 4)
    return v


def random_texture(size: int, seed: int, alpha: Optional[int] = 255) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, size=(size, size, 4), dtype=np.uint8)
    if alpha is not None:
        t[..., 3] = alpha
    return t


# ------------------------------------------------------------------------------------- cfg1
def cfg1() -> Scene:
    """256x256, one flat-shaded triangle in clip space, identity matrices, no texture, no depth."""
    pos = [(-0.5, -0.5, 0.0), (0.5, -0.5, 0.0), (0.0, 0.5, 0.0)]
    v = make_vertices(pos, color=[(1, 0, 0, 1), (0, 1, 0, 1), (0, 0, 1, 1)])
    I = hm.identity()
    d = Draw(v, np.array([0, 1, 2], dtype=np.uint16), I, I, I, program=Program.FlatColor,
             cull=CullMode.None_, depth_test=DepthTest.Disabled, blend=BlendMode.Alpha)
    return Scene("cfg1", 256, 256, [d], clear_color=(0.0, 0.0, 0.0, 1.0))


# ------------------------------------------------------------------------------------- cfg2
def _perspective(width, height, fov_deg=90.0, near=0.1, far=1000.0):
    return hm.create_perspective_fov(np.float32(fov_deg) * np.float32(np.pi) / np.float32(180.0),
                                     np.float32(width) / np.float32(height), near, far)


def cfg2(width=1920, height=1080, n_tris=10000, seed=1, min_area=10.0, max_area=10000.0) -> Scene:
    """Random depth-tested Gouraud triangles, one u16 mesh of 3*n unshared vertices (n <= 21845)."""
    assert 3 * n_tris <= 65535
    rng = np.random.default_rng(seed)
    proj = _perspective(width, height)
    xs, ys = float(proj[0, 0]), float(proj[1, 1])
    cx = rng.uniform(-1, 1, n_tris); cy = rng.uniform(-1, 1, n_tris)
    z = rng.uniform(1.0, 50.0, n_tris)
    area = np.exp(rng.uniform(np.log(min_area), np.log(max_area), n_tris))      # px^2, log-uniform
    r_px = np.sqrt(area / 1.299)                                                # circumradius of an equilateral triangle of that area
    ang0 = rng.uniform(0, 2 * np.pi, n_tris)
    pos = np.empty((n_tris, 3, 3))
    for k in range(3):
        ang = ang0 + k * 2 * np.pi / 3 + rng.uniform(-0.4, 0.4, n_tris)
        rr = r_px * rng.uniform(0.7, 1.3, n_tris)
        zk = z * rng.uniform(0.9, 1.1, n_tris)
        ndc_x = cx + rr * np.cos(ang) * 2.0 / width
        ndc_y = cy + rr * np.sin(ang) * 2.0 / height
        pos[:, k, 0] = ndc_x * zk / xs
        pos[:, k, 1] = ndc_y * zk / ys
        pos[:, k, 2] = -zk
    col = np.concatenate([rng.uniform(0, 1, (n_tris * 3, 3)), np.ones((n_tris * 3, 1))], axis=1)
    v = make_vertices(pos.reshape(-1, 3), color=col)
    idx = np.arange(3 * n_tris, dtype=np.uint16)
    I = hm.identity()
    d = Draw(v, idx, I, I, proj, program=Program.Gouraud, cull=CullMode.None_,
             depth_test=DepthTest.LessEqual, blend=BlendMode.Alpha)
    return Scene(f"cfg2_{width}x{height}_{n_tris}", width, height, [d])


# ------------------------------------------------------------------------------------- cfg3 / cfg4 / cfg5
def _patch_mesh(rng, nx, ny, centre_ndc, half_ndc, z_near, z_far, inv_vp, uv_lo=-2.0, uv_hi=3.0, wobble=0.15):
    """An indexed (nx x ny quads) grid patch: positions chosen in NDC/view depth, unprojected to model space."""
    gx, gy = np.meshgrid(np.linspace(-1, 1, nx + 1), np.linspace(-1, 1, ny + 1))
    jx = rng.uniform(-0.3, 0.3, gx.shape) / nx
    jy = rng.uniform(-0.3, 0.3, gy.shape) / ny
    ndc_x = centre_ndc[0] + (gx + jx) * half_ndc[0]
    ndc_y = centre_ndc[1] + (gy + jy) * half_ndc[1]
    # view-space distance: tilted plane + smooth wobble
    t = 0.5 * (gx + 1.0)
    dist = z_near + (z_far - z_near) * (0.5 * t + 0.25 * (gy + 1.0))
    dist = dist * (1.0 + wobble * np.sin(3.1 * gx + 1.3) * np.cos(2.7 * gy - 0.4))
    # clip = (ndc * w, w) with w = dist ; z_clip from the projection's z mapping is recomputed by the shader,
    # so unproject through view space directly: p_view = (ndc_x * dist / xs, ndc_y * dist / ys, -dist)
    xs, ys, inv_view_model = inv_vp
    pv = np.stack([ndc_x * dist / xs, ndc_y * dist / ys, -dist, np.ones_like(dist)], axis=-1).reshape(-1, 4)
    pm = pv @ inv_view_model                         # row-vector convention
    pos = (pm[:, :3] / pm[:, 3:4]).astype(np.float32)
    # smooth-ish normals from the height field, in model space (only their direction matters)
    nrm = np.stack([np.cos(3.1 * gx + 1.3) * 0.5, np.sin(2.7 * gy - 0.4) * 0.5, np.ones_like(gx)], axis=-1).reshape(-1, 3)
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    uv = np.stack([uv_lo + (uv_hi - uv_lo) * 0.5 * (gx + 1.0), uv_lo + (uv_hi - uv_lo) * 0.5 * (gy + 1.0)], axis=-1).reshape(-1, 2)
    col = np.concatenate([rng.uniform(0.4, 1.0, (pos.shape[0], 3)), np.ones((pos.shape[0], 1))], axis=1)
    v = make_vertices(pos, uv=uv, normal=nrm, color=col)
    # two triangles per quad; winding chosen so the reference's "front" (area < 0) faces the camera
    i0 = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    i1, i2, i3 = i0 + 1, i0 + (nx + 1), i0 + (nx + 2)
    tris = np.stack([i0, i1, i2, i1, i3, i2], axis=1).reshape(-1)
    assert v.shape[0] <= 65535
    return v, tris.astype(np.uint16)


def cfg3(width=4096, height=4096, grid=(4, 4), quads=(250, 125), seed=2, tex_size=2048,
         program=Program.Dust2LambertFog, name=None, bilinear=False) -> Scene:
    """grid[0]*grid[1] indexed patches of quads[0]*quads[1]*2 triangles each (default 16 x 62,500 = 1M),
    overlapping with depth complexity about 3, one random RGBA8 texture, UVs in [-2,3] (wrap).
    bilinear=True: the same frame through the BUILD-DEFINED bilinear filter (BASELINE.json words cfg3 as "bilinear-textured"; the
    reference's Texture.Sample is nearest, Texture.cs:43-63, which stays the default and the benchmark's configuration)."""
    rng = np.random.default_rng(seed)
    proj = _perspective(width, height)
    model = hm.create_scale(0.5)                                     # Renderer.cs:32 ModelMatrix
    view = hm.create_look_at((0.3, 0.2, 1.0), (0.0, 0.0, -10.0), (0.0, 1.0, 0.0))
    inv_view_model = np.linalg.inv(model.astype(np.float64) @ view.astype(np.float64))
    inv_vp = (float(proj[0, 0]), float(proj[1, 1]), inv_view_model)
    tex = random_texture(tex_size, seed + 100)
    uni = default_uniforms()
    if program == Program.Phong4Point:
        uni.camera_position[:] = (0.3, 0.2, 1.0)
        uni.shininess = 16.0
        lp = [(-6.0, 4.0, -8.0), (6.0, 4.0, -12.0), (-4.0, -5.0, -20.0), (5.0, -3.0, -30.0)]
        lc = [(1.0, 0.9, 0.8), (0.6, 0.7, 1.0), (0.9, 0.5, 0.5), (0.5, 1.0, 0.6)]
        for i in range(4):
            uni.lights[i].position[:] = lp[i]
            uni.lights[i].range = 60.0
            uni.lights[i].color[:] = lc[i]
            uni.lights[i].intensity = 1.5
    gx, gy = grid
    draws = []
    order = rng.permutation(gx * gy)
    for k in order:
        ix, iy = int(k % gx), int(k // gx)
        centre = (-1 + (2 * ix + 1) / gx + rng.uniform(-0.1, 0.1) / gx, -1 + (2 * iy + 1) / gy + rng.uniform(-0.1, 0.1) / gy)
        half = (1.75 / gx, 1.75 / gy)                 # patches 1.75x their cell: overlap -> depth complexity ~3
        zn = rng.uniform(2.0, 20.0)
        v, idx = _patch_mesh(rng, quads[0], quads[1], centre, half, zn, zn * rng.uniform(1.3, 2.2), inv_vp)
        draws.append(Draw(v, idx, model, view, proj, program=program, uniforms=uni, texture=0,
                          cull=CullMode.Back, depth_test=DepthTest.LessEqual, blend=BlendMode.Alpha))
    n = gx * gy * quads[0] * quads[1] * 2
    return Scene(name or f"cfg3_{width}x{height}_{n}" + ("_bilinear" if bilinear else ""), width, height, draws, textures=[tex],
                 bilinear=bool(bilinear))


def cfg4(**kw) -> Scene:
    """cfg3 geometry with the build-defined 4-point-light Phong program (no reference semantics)."""
    kw.setdefault("program", Program.Phong4Point)
    s = cfg3(**kw)
    s.name = s.name.replace("cfg3", "cfg4")
    return s


def cfg5(**kw) -> Scene:
    kw.setdefault("width", 8192); kw.setdefault("height", 8192)
    s = cfg3(**kw)
    s.name = s.name.replace("cfg3", "cfg5")
    return s


def from_model(model, width=640, height=480, eye=None, seed=3, tex_size=64, program=Program.Dust2LambertFog,
               name="model") -> Scene:
    """One frame of Renderer.RenderDust2 (Renderer.cs:422-468) over a loaded `modelloader.Model`: one RenderMesh per
    mesh in order, ModelMatrix = CreateScale(0.5), CullMode.Back / LessEqual / Alpha.  Image decoding (ImageSharp in the
    reference) is out of scope, so each mesh gets a procedural RGBA8 texture instead of its material's file."""
    proj = _perspective(width, height)
    mm = hm.create_scale(0.5)
    allp = np.concatenate([m.Vertices["position"] for m in model.Meshes]).astype(np.float64) * 0.5
    lo, hi = allp.min(axis=0), allp.max(axis=0)
    centre, radius = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2 + 1e-3
    if eye is None:
        eye = centre + np.array([0.35, 0.3, 1.0]) * radius * 1.6
    view = hm.create_look_at(tuple(float(v) for v in eye), tuple(float(v) for v in centre), (0.0, 1.0, 0.0))
    uni = default_uniforms()
    uni.fog_start, uni.fog_end = radius * 2.0, radius * 6.0
    textures = [random_texture(tex_size, seed + i) for i in range(len(model.Meshes))]
    draws = [Draw(m.Vertices, m.Indices, mm, view, proj, program=program, uniforms=uni, texture=i)
             for i, m in enumerate(model.Meshes)]
    return Scene(name, width, height, draws, textures=textures)


# ------------------------------------------------------------------------------------- edge-case scenes
def near_clip_scene(width=320, height=200, n_tris=300, seed=7, program=Program.Dust2LambertFog) -> Scene:
    """Triangles straddling the camera plane (some W<=0): exercises ClipTriangleAgainstNearPlane."""
    rng = np.random.default_rng(seed)
    proj = _perspective(width, height)
    pos = np.empty((n_tris, 3, 3))
    c = np.stack([rng.uniform(-3, 3, n_tris), rng.uniform(-2, 2, n_tris), rng.uniform(-4.0, 1.5, n_tris)], axis=1)
    for k in range(3):
        pos[:, k, :] = c + rng.uniform(-2.5, 2.5, (n_tris, 3))
    col = np.concatenate([rng.uniform(0, 1, (n_tris * 3, 3)), rng.uniform(0.3, 1.0, (n_tris * 3, 1))], axis=1)
    nrm = rng.normal(size=(n_tris * 3, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v = make_vertices(pos.reshape(-1, 3), uv=rng.uniform(-2, 3, (n_tris * 3, 2)), normal=nrm, color=col)
    I = hm.identity()
    d = Draw(v, np.arange(3 * n_tris, dtype=np.uint16), hm.create_rotation_y(0.3), I, proj, program=program,
             uniforms=default_uniforms(), texture=0, cull=CullMode.None_, depth_test=DepthTest.LessEqual, blend=BlendMode.Alpha)
    return Scene(f"nearclip_{program.name}", width, height, [d], textures=[random_texture(64, seed, alpha=None)])


def state_scene(width=200, height=150, n_tris=400, seed=11, cull=CullMode.None_, depth_test=DepthTest.LessEqual,
                blend=BlendMode.Alpha, alpha_range=(0.0, 1.0), program=Program.Gouraud, zero_alpha_fraction=0.2) -> Scene:
    """Small random soup for sweeping every CullMode / DepthTest / BlendMode enumerant (translucent + zero alpha)."""
    s = cfg2(width, height, n_tris, seed, min_area=30.0, max_area=4000.0)
    d = s.draws[0]
    rng = np.random.default_rng(seed + 1)
    a = rng.uniform(alpha_range[0], alpha_range[1], d.vertices.shape[0]).astype(np.float32)
    zero = rng.uniform(0, 1, d.vertices.shape[0] // 3) < zero_alpha_fraction
    a = a.reshape(-1, 3); a[zero, :] = 0.0
    d.vertices["color"][:, 3] = a.reshape(-1)
    d.cull, d.depth_test, d.blend, d.program = cull, depth_test, blend, program
    d.uniforms = default_uniforms()
    s.name = f"state_{cull.name}_{depth_test.name}_{blend.name}_{program.name}"
    return s


def stacked_scene(width=64, height=48, layers=200, seed=5) -> Scene:
    """Many full-screen, equal-depth, translucent triangles: long per-tile lists (sort paths) + order dependence."""
    rng = np.random.default_rng(seed)
    base = np.array([(-1.5, -1.5, -3.0), (3.5, -1.5, -3.0), (-1.5, 3.5, -3.0)])
    pos = np.tile(base, (layers, 1))
    col = np.repeat(np.concatenate([rng.uniform(0, 1, (layers, 3)), rng.uniform(0.2, 0.9, (layers, 1))], axis=1), 3, axis=0)
    v = make_vertices(pos, color=col)
    proj = _perspective(width, height)
    I = hm.identity()
    d = Draw(v, np.arange(3 * layers, dtype=np.uint16), I, I, proj, program=Program.Gouraud, cull=CullMode.None_,
             depth_test=DepthTest.LessEqual, blend=BlendMode.Alpha)
    return Scene(f"stacked_{layers}", width, height, [d])


def degenerate_scene(width=96, height=80) -> Scene:
    """NaN / Inf / zero-area / W==0 / off-screen triangles mixed with one visible triangle (all silently skipped)."""
    nan, inf = float("nan"), float("inf")
    tris = [
        [(-0.5, -0.5, -2.0), (0.5, -0.5, -2.0), (0.0, 0.5, -2.0)],          # visible
        [(nan, 0.0, -2.0), (0.5, 0.5, -2.0), (0.0, 0.5, -2.0)],             # NaN
        [(inf, 0.0, -2.0), (0.5, 0.5, -2.0), (0.0, 0.5, -2.0)],             # Inf
        [(0.0, 0.0, -2.0), (0.0, 0.0, -2.0), (0.0, 0.0, -2.0)],             # zero area
        [(0.0, 0.0, -2.0), (1.0, 1.0, -2.0), (2.0, 2.0, -2.0)],             # collinear
        [(0.1, 0.1, 0.0), (0.5, -0.5, -2.0), (0.0, 0.5, -2.0)],             # one vertex with W == 0 exactly
        [(50.0, 50.0, -2.0), (51.0, 50.0, -2.0), (50.0, 51.0, -2.0)],       # off screen
        [(0.0, 0.0, 5.0), (1.0, 0.0, 5.0), (0.0, 1.0, 5.0)],                # all behind
        [(3e38, 0.0, -1e-30), (0.5, 0.5, -2.0), (0.0, 0.5, -2.0)],          # overflow to inf in screen space
    ]
    pos = np.array(tris, dtype=np.float32).reshape(-1, 3)
    rng = np.random.default_rng(3)
    col = np.concatenate([rng.uniform(0, 1, (pos.shape[0], 3)), np.ones((pos.shape[0], 1))], axis=1)
    v = make_vertices(pos, color=col)
    I = hm.identity()
    d = Draw(v, np.arange(pos.shape[0], dtype=np.uint16), I, I, _perspective(width, height), program=Program.Gouraud,
             cull=CullMode.None_, depth_test=DepthTest.LessEqual, blend=BlendMode.Alpha)
    return Scene("degenerate", width, height, [d])


# ------------------------------------------------------------------------------------- rendering through the HIP backend
class SceneRenderer:
    """Uploads a Scene's meshes/textures once (retained handles) and replays its frame on the GPU."""

    def __init__(self, device: Device, scene: Scene, window: Optional[MainWindow] = None, retained: bool = True):
        self.dev, self.scene = device, scene
        self.window = window or MainWindow(device, scene.width, scene.height)
        self.textures = [Texture(device, t) for t in scene.textures]
        if scene.bilinear:
            for t in self.textures:
                t.SetBilinear(True)
        self.programs, self.meshes = [], []
        for d in scene.draws:
            tex = self.textures[d.texture] if d.texture is not None else None
            self.programs.append(ShaderProgram(d.program, d.uniforms, tex))
            self.meshes.append(Mesh(device, d.vertices, d.indices) if retained else None)
        self._calls = None

    def _prepare(self):
        """ctypes arguments of every RenderMesh call, converted once (the per-frame host cost is then ~16 FFI calls)."""
        ctx = self.dev._ctx
        fp = C.POINTER(C.c_float)
        calls = []
        for d, prog, mesh in zip(self.scene.draws, self.programs, self.meshes):
            if mesh is None:
                return None
            mats = [np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(-1)) for m in (d.model, d.view, d.projection)]
            calls.append((mats, (ctx, mesh._h, mats[0].ctypes.data_as(fp), mats[1].ctypes.data_as(fp), mats[2].ctypes.data_as(fp),
                                 int(prog.program), C.byref(prog.uniforms), prog.texture._h if prog.texture is not None else None,
                                 int(d.cull), int(d.depth_test), int(d.blend))))
        return calls

    def submit_frame(self):
        """One frame = RenderScene of Renderer.cs:404-419: clears, then one RenderMesh per mesh (not flushed)."""
        s, w = self.scene, self.window
        w._activate()
        Rasterizer.NearClip, Rasterizer.FarClip = s.near_clip, s.far_clip
        if s.clear_depth:
            w.ClearDepthBuffer()
        if s.clear_color is not None:
            w.ClearColorBuffer(s.clear_color)
        if self._calls is None:
            self._calls = self._prepare() or False
        if self._calls:
            dev = self.dev
            dev._ck(dev._lib.swr_set_state(dev._ctx, float(Rasterizer.NearClip), float(Rasterizer.FarClip), int(Rasterizer.RenderDebugMode)))
            render = dev._lib.swr_render_mesh
            for _, args in self._calls:
                rc = render(*args)
                if rc:
                    dev._ck(rc)
            return
        for d, prog, mesh in zip(s.draws, self.programs, self.meshes):
            Rasterizer.RenderMesh(w, mesh if mesh is not None else d.vertices, d.indices, d.model, d.view, d.projection,
                                  prog.VertexShader, prog.FragmentShader, d.cull, d.depth_test, d.blend)

    def jitter_views(self, frame_no: int, amplitude: float):
        """Host-side camera motion for benchmarks (bench.py --camera-jitter): frame `frame_no` sees every draw's view matrix followed
        by a small yaw and a translation of `amplitude` view-space units, different each frame (amplitude 0 restores the scene's own
        matrices).  Only the float arrays the prepared RenderMesh calls point at change: nothing is re-uploaded."""
        if self._calls is None:
            self._calls = self._prepare() or False
        if not self._calls:
            raise RuntimeError("jitter_views needs retained meshes")
        if not hasattr(self, "_base_views"):
            self._base_views = [mats[1].copy() for mats, _ in self._calls]
        a = float(amplitude)
        k = float(frame_no)
        yaw = 0.02 * a * np.sin(0.9 * k + 0.3)
        t = np.eye(4, dtype=np.float64)
        t[0, 0] = np.cos(yaw); t[0, 2] = -np.sin(yaw); t[2, 0] = np.sin(yaw); t[2, 2] = np.cos(yaw)       # row-vector convention
        t[3, 0] = a * np.sin(0.7 * k); t[3, 1] = a * np.cos(1.3 * k + 0.5); t[3, 2] = 0.5 * a * np.sin(0.31 * k)
        for (mats, _), base in zip(self._calls, self._base_views):
            mats[1][:] = (base.reshape(4, 4).astype(np.float64) @ t).astype(np.float32).reshape(-1) if a != 0.0 else base

    def render(self):
        self.submit_frame()
        return self.window._read(True, True)

    def close(self):
        for m in self.meshes:
            if m is not None:
                m.Dispose()
        for t in self.textures:
            t.Dispose()
