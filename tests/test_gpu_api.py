"""GPU tests of the C-ABI surface beyond whole-frame parity: accessors, texture sampler, Interpolate,
tile-row bands, flush boundaries, error behaviour, golden depth hashes, concurrent callers."""
import ctypes as C
import hashlib
import importlib.util
import json
import os
import threading

import numpy as np
import pytest

from softwarerenderer_amd import _native as N, hostmath as hm, multigpu, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from softwarerenderer_amd.rasterizer import (BlendMode, CullMode, DebugMode, DepthTest, MainWindow, Mesh, Program,
                                              Rasterizer, Shaders, Texture)
from util import assert_frame_parity, render_oracle, ulp_distance

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MINVAL = np.float32(-3.4028235e38)


def test_golden_depth_hashes(device):
    """The committed fixtures (tests/golden) hit from the GPU: depth hash exact, colour hash when 0 ULP."""
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    golden = json.load(open(os.path.join(HERE, "golden", "oracle_golden.json")))
    for scene in mg.golden_scenes():
        r = scenes.SceneRenderer(device, scene)
        c, d = r.render()
        r.close()
        g = golden[scene.name]
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["depth_sha256"], scene.name
        for s in g["samples"]:
            want = np.array(s["color_bits"], dtype=np.uint32).view(np.float32)
            assert ulp_distance(c[s["y"], s["x"]], want).max() <= 1, scene.name
            assert int(d[s["y"], s["x"]].view(np.uint32)) == s["depth_bits"], scene.name


def test_texture_sample_matches_oracle(device, oracle_lib):
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)        # non power of two, non square
    uv = np.concatenate([rng.uniform(-3, 4, (4000, 2)), np.array([[0, 0], [1, 1], [-1, -1], [0.999999, 1e-9], [-1e-9, 2.0],
                                                                 [1e20, -1e20], [np.nan, 0.5], [np.inf, -np.inf], [53.0, 37.0]])]).astype(np.float32)
    t = Texture(device, tex)
    got = t.Sample(uv)
    want = np.zeros_like(got)
    for i in range(uv.shape[0]):
        oracle_lib.oswr_texture_sample(tex.ctypes.data, 53, 37, uv[i].ctypes.data, want[i].ctypes.data)
    t.Dispose()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (t.Width, t.Height) == (53, 37)


def test_public_interpolate_matches_oracle(device, oracle_lib):
    from oracle import binding as ob
    rng = np.random.default_rng(4)
    recs = rng.uniform(-2, 2, (3, 20)).astype(np.float32)
    recs[:, 3] = rng.uniform(0.5, 5.0, 3)                                # clip.w > 0
    w = rng.uniform(-1.0, 0.0, (500, 3)).astype(np.float32)              # the reference's weights are negative (sum -1)
    win = MainWindow(device, 16, 16)
    for interp in (True, False):
        got = Rasterizer.Interpolate(win, recs[0], recs[1], recs[2], w, interp)
        vs = []
        for r in recs:
            v = ob.OVertexOutput()
            v.clip[:] = r[0:4]; v.color[:] = r[4:8]; v.texcoord[:] = r[8:10]; v.normal[:] = r[10:13]; v.screen[:] = r[13:15]
            v.world_normal[:] = r[15:18]; v.has_data = 1; v.interpolate = 1
            vs.append(v)
        for i in range(w.shape[0]):
            o = ob.OVertexOutput()
            oracle_lib.oswr_interpolate(C.byref(vs[0]), C.byref(vs[1]), C.byref(vs[2]), float(w[i, 0]), float(w[i, 1]), float(w[i, 2]), int(interp), C.byref(o))
            want = np.array(list(o.clip) + list(o.color) + list(o.texcoord) + list(o.normal) + list(o.screen) + list(o.world_normal) + list(o.barycentric), dtype=np.float32)
            assert np.array_equal(got[i, :21].view(np.uint32), want.view(np.uint32)), (interp, i)


def test_framebuffer_accessors(device):
    w = MainWindow(device, 40, 24)
    w.ClearColorBuffer((0.25, 0.5, 0.75, 1.0)); w.ClearDepthBuffer()
    assert tuple(w.GetPixel(3, 4)) == (0.25, 0.5, 0.75, 1.0) and w.GetDepth(3, 4) == MINVAL
    w.SetPixel(39, 23, (1, 2, 3, 4)); w.SetDepth(0, 0, 0.125)
    w.SetPixel(40, 0, (9, 9, 9, 9)); w.SetPixel(-1, 0, (9, 9, 9, 9)); w.SetDepth(0, 24, 7.0)     # ignored (MainWindow.cs:384,413)
    assert tuple(w.GetPixel(39, 23)) == (1, 2, 3, 4) and w.GetDepth(0, 0) == 0.125
    assert tuple(w.GetPixel(40, 0)) == (0, 0, 0, 0) and tuple(w.GetPixel(0, -1)) == (0, 0, 0, 0)  # Vector4.Zero, :397
    assert w.GetDepth(0, 24) == MINVAL and w.GetDepth(-5, 2) == MINVAL                           # float.MinValue, :425
    col, dep = w.ColorBuffer, w.DepthBuffer
    assert col.shape == (24, 40, 4) and dep.shape == (24, 40)
    assert dep[0, 0] == 0.125 and tuple(col[23, 39]) == (1, 2, 3, 4) and (dep.ravel()[1:] == MINVAL).all()


def test_present_side_rgb_flatten(device):
    scene = scenes.cfg2(123, 77, 150, seed=19)
    r = scenes.SceneRenderer(device, scene)
    c, _ = r.render()
    rgb = r.window.FlatColorBuffer()
    r.close()
    assert rgb.shape == (77, 123, 3) and np.array_equal(rgb.view(np.uint32), c[..., :3].view(np.uint32))


def test_zero_size_target_skips_silently_and_tile_locks_validate(device):
    w = MainWindow(device, 0, 0)
    s = scenes.cfg1()
    p = Shaders.FlatColor()
    Rasterizer.RenderMesh(w, s.draws[0].vertices, s.draws[0].indices, hm.identity(), hm.identity(), hm.identity(),
                          p.VertexShader, p.FragmentShader)                    # Rasterizer.cs:176: returns
    assert w.ColorBuffer.size == 0
    with pytest.raises(ValueError):                                            # Rasterizer.cs:71-74 ArgumentException
        Rasterizer.InitializeTileLocks(w, 0, 10)
    Rasterizer.InitializeTileLocks(w, 10, 10)


def test_error_behaviour(device):
    w = MainWindow(device, 32, 32)
    s = scenes.cfg1()
    p = Shaders.FlatColor()
    I = hm.identity()
    with pytest.raises(IndexError):                                            # C#: IndexOutOfRangeException
        Rasterizer.RenderMesh(w, s.draws[0].vertices, np.array([0, 1, 9], dtype=np.uint16), I, I, I, p.VertexShader, p.FragmentShader)
    q = Shaders.Gouraud()
    with pytest.raises(ValueError):
        Rasterizer.RenderMesh(w, s.draws[0].vertices, s.draws[0].indices, I, I, I, p.VertexShader, q.FragmentShader)
    # an index count that is not a multiple of 3 ignores the tail (indices.Length / 3, Rasterizer.cs:180)
    Rasterizer.RenderMesh(w, s.draws[0].vertices, np.array([0, 1, 2, 0, 1], dtype=np.uint16), I, I, I, p.VertexShader, p.FragmentShader,
                          CullMode.None_, DepthTest.Disabled)
    assert device.stats()["flushes"] >= 0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_row_bands_union_is_the_single_gpu_frame(device, world):
    """Multi-GPU partition property on one GPU: rendering each band separately and stacking them gives the
    full frame bit for bit (a triangle straddling a band edge is rasterised on both sides identically)."""
    scene = scenes.cfg3(300, 270, (3, 3), (20, 14), tex_size=64, seed=8)
    rc, rd, _ = render_oracle(scene)
    cols, deps, frag, tris = [], [], 0, []
    for band in multigpu.band_partition(scene.height, world):
        win = MainWindow(device, scene.width, scene.height)
        win.SetBand(*band)
        device.reset_stats()
        r = scenes.SceneRenderer(device, scene, window=win)
        c, d = r.render()
        frag += device.stats()["fragments_written"]
        tris.append(device.stats()["triangles_in"])
        r.close()
        assert c.shape[0] == multigpu.band_pixel_rows(scene.height, band)[1]
        cols.append(c); deps.append(d)
    win = MainWindow(device, scene.width, scene.height)
    win.SetBand(-1, -1)                                                        # back to the whole frame
    c, d = np.concatenate(cols, axis=0), np.concatenate(deps, axis=0)
    assert_frame_parity(c, d, rc, rd, 1, f"bands{world}")
    assert frag == render_oracle(scene)[2]["fragments_written"]
    # meshes whose bounding box projects outside a band are not even recorded there (band_rejects, swr_api.hip)
    if world >= 3:
        assert min(tris) < scene.n_triangles and max(tris) <= scene.n_triangles


def test_flush_boundaries_do_not_change_the_frame(device):
    """One batch vs a flush (and a readback) after every mesh: same frame."""
    scene = scenes.cfg3(256, 256, (3, 3), (16, 12), tex_size=64, seed=12)
    r = scenes.SceneRenderer(device, scene)
    c1, d1 = r.render()
    s, w = scene, r.window
    w.ClearDepthBuffer(); w.ClearColorBuffer(s.clear_color)
    for dr, prog, mesh in zip(s.draws, r.programs, r.meshes):
        Rasterizer.RenderMesh(w, mesh, None, dr.model, dr.view, dr.projection, prog.VertexShader, prog.FragmentShader,
                              dr.cull, dr.depth_test, dr.blend)
        device.flush()
        _ = w.GetPixel(0, 0)
    c2, d2 = w._read()
    r.close()
    assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32)) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))


def test_resize_and_odd_sizes(device):
    for size in [(1, 1), (17, 5), (16, 16), (15, 33), (640, 3)]:
        scene = scenes.cfg2(size[0], size[1], 120, seed=6, min_area=4.0, max_area=900.0)
        rc, rd, rst = render_oracle(scene)
        r = scenes.SceneRenderer(device, scene)
        c, d = r.render()
        r.close()
        assert_frame_parity(c, d, rc, rd, 1, f"size{size}")


def test_concurrent_callers_are_recorded_safely(device):
    """RenderMesh is invoked from Parallel.ForEach workers in the reference (Renderer.cs:444): the ABI must be
    thread-safe.  Opaque draws with distinct depths are order-independent, so any interleaving gives the frame."""
    scene = scenes.cfg3(256, 256, (3, 3), (16, 12), tex_size=64, seed=14)
    rc, rd, _ = render_oracle(scene)
    r = scenes.SceneRenderer(device, scene)
    w = r.window
    w.ClearDepthBuffer(); w.ClearColorBuffer(scene.clear_color)

    def work(i):
        dr, prog, mesh = scene.draws[i], r.programs[i], r.meshes[i]
        dev, lib = device, device._lib
        m, v, p = (np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in (dr.model, dr.view, dr.projection))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        rcode = lib.swr_render_mesh(dev._ctx, mesh._h, fp(m), fp(v), fp(p), int(prog.program), C.byref(prog.uniforms),
                                    prog.texture._h, int(dr.cull), int(dr.depth_test), int(dr.blend))
        assert rcode == 0
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(scene.draws))]
    [t.start() for t in ts]; [t.join() for t in ts]
    c, d = w._read()
    r.close()
    same_depth = np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    assert same_depth or (d != rd).mean() < 1e-4


def test_retained_mesh_reuse_and_stats(device):
    scene = scenes.cfg2(160, 120, 200, seed=17)
    rc, rd, rst = render_oracle(scene)
    r = scenes.SceneRenderer(device, scene)
    device.reset_stats()
    for _ in range(3):
        c, d = r.render()
    st = device.stats()
    r.close()
    assert_frame_parity(c, d, rc, rd, 1, "reuse")
    assert st["triangles_in"] == 3 * rst["triangles_in"] and st["fragments_tested"] == 3 * rst["fragments_tested"]
    assert st["flushes"] == 3 and st["tile_pairs"] > 0


def test_optimistic_flush_overflow_is_replayed_exactly():
    """Flushes run without reading the pair total back; a batch that does not fit the pair buffers poisons itself and
    every later batch on the device, and the host replays them at the next synchronisation point.  Force that path:
    size the buffers with a tiny scene, then submit two much larger, order-dependent frames back to back."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    device = Device(0)                       # a fresh context (the session's shared one may already own large pair buffers)
    small = scenes.cfg2(256, 256, 40, seed=50, min_area=10.0, max_area=60.0)
    big1 = scenes.cfg2(256, 256, 3000, seed=51, min_area=200.0, max_area=9000.0)
    big2 = scenes.state_scene(256, 256, 2500, seed=52, blend=BlendMode.Additive)
    big2.clear_color = None; big2.clear_depth = False          # accumulates on top of big1's frame
    o = OracleRenderer(256, 256)
    o.render_scene(small); o.reset_stats()
    o.render_scene(big1)
    rc, rd = o.render_scene(big2)
    rst = o.stats()

    r0 = scenes.SceneRenderer(device, small)
    r0.render()                                                 # synchronous sizing of the pair buffers
    device.reset_stats()
    r1 = scenes.SceneRenderer(device, big1, window=r0.window)
    r2 = scenes.SceneRenderer(device, big2, window=r0.window)
    r1.submit_frame(); device.flush()                           # optimistic: overflows -> poison
    r2.submit_frame(); device.flush()                           # skipped on the device while poisoned
    replays = device.replay_count()
    c, d = r0.window._read()                                    # sync point: validate + replay, then read
    assert device.replay_count() == replays + 1                 # the path under test really ran
    st = device.stats()
    for r in (r0, r1, r2):
        r.close()
    device.close()
    assert_frame_parity(c, d, rc, rd, 1, "replay")
    for k in ("triangles_in", "triangles_setup", "fragments_tested", "fragments_shaded", "fragments_written"):
        assert st[k] == rst[k], (k, st[k], rst[k])


def test_frustum_culler_bounds_and_test_match_oracle(device, oracle_lib):
    """FrustumCuller.CalculateBoundingSphere / IsSphereInFrustum (FrustumCuller.cs) on the GPU: bit-identical to the
    serial-schedule oracle, incl. degenerate inputs."""
    from softwarerenderer_amd.rasterizer import FrustumCuller
    rng = np.random.default_rng(77)
    win = MainWindow(device, 64, 64)
    cases = [scenes.cfg3(128, 128, (2, 2), (10, 6), tex_size=8, seed=s_).draws[0].vertices for s_ in (1, 2, 3)]
    cases.append(scenes.make_vertices(rng.normal(size=(5000, 3)) * [5, 1, 0.2]))
    cases.append(scenes.make_vertices(np.tile([[1.0, 2.0, 3.0]], (7, 1))))            # all identical points
    cases.append(scenes.make_vertices([[1.0, 2.0, 3.0]]))                               # single vertex
    cases.append(scenes.make_vertices([[0.0, 0.0, 0.0], [4.0, 0.0, 0.0], [2.0, 9.0, 0.0], [2.0, -9.0, 0.0], [2.0, 0.0, 9.5]]))
    for v in cases:
        v = np.ascontiguousarray(v)
        mesh = Mesh(device, v, np.zeros(3, dtype=np.uint16))
        got = FrustumCuller.CalculateBoundingSphere(mesh)
        want = np.zeros(4, np.float32)
        oracle_lib.oswr_bounding_sphere(v.ctypes.data, v.shape[0], want.ctypes.data)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
        for _ in range(40):
            model = hm.multiply(hm.create_scale(float(rng.uniform(0.2, 3))), hm.create_translation(*rng.uniform(-30, 30, 3)))
            view = hm.create_look_at(tuple(rng.uniform(-20, 20, 3)), tuple(rng.uniform(-5, 5, 3)), (0.0, 1.0, 0.0))
            proj = hm.create_perspective_fov(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)), 0.1, 1000.0)
            m_, v_, p_ = (np.ascontiguousarray(a, dtype=np.float32) for a in (model, view, proj))
            w_ = oracle_lib.oswr_is_sphere_in_frustum(want.ctypes.data, m_.ctypes.data, v_.ctypes.data, p_.ctypes.data)
            assert FrustumCuller.IsSphereInFrustum(win, got, model, view, proj) == bool(w_)
        mesh.Dispose()


def test_device_side_frustum_culling_equals_host_side_decision(device, oracle_lib):
    """RenderMesh(frustumCull=True) == the reference's `if (!IsSphereInFrustum(...)) return; RenderMesh(...)`."""
    from oracle.binding import OracleRenderer
    scene = scenes.cfg3(320, 240, (4, 4), (10, 8), tex_size=32, seed=33)
    # narrow the view so that several of the 16 patches fall outside the frustum
    proj = scenes._perspective(320, 240, fov_deg=35.0)
    o = OracleRenderer(scene.width, scene.height)
    o.clear_depth(); o.clear_color(scene.clear_color)
    kept = 0
    for i, d in enumerate(scene.draws):
        d.projection = proj
        if i % 3 == 1:                                      # push every third mesh far out of view
            d.model = hm.multiply(d.model, hm.create_translation(400.0 + 50.0 * i, 0.0, 0.0))
        v = np.ascontiguousarray(d.vertices)
        sph = np.zeros(4, np.float32)
        oracle_lib.oswr_bounding_sphere(v.ctypes.data, v.shape[0], sph.ctypes.data)
        m_, v_, p_ = (np.ascontiguousarray(a, dtype=np.float32) for a in (d.model, d.view, d.projection))
        if oracle_lib.oswr_is_sphere_in_frustum(sph.ctypes.data, m_.ctypes.data, v_.ctypes.data, p_.ctypes.data):
            kept += 1
            o.render_mesh(d.vertices, d.indices, d.model, d.view, d.projection, int(d.program), d.uniforms, scene.textures[0],
                          int(d.cull), int(d.depth_test), int(d.blend))
    assert 0 < kept < len(scene.draws)
    rst = o.stats()
    r = scenes.SceneRenderer(device, scene)
    w = r.window
    device.reset_stats()
    w.ClearDepthBuffer(); w.ClearColorBuffer(scene.clear_color)
    for d, prog, mesh in zip(scene.draws, r.programs, r.meshes):
        Rasterizer.RenderMesh(w, mesh, None, d.model, d.view, d.projection, prog.VertexShader, prog.FragmentShader,
                              d.cull, d.depth_test, d.blend, frustumCull=True)
    c, dz = w._read()
    st = device.stats()
    r.close()
    assert_frame_parity(c, dz, o.color, o.depth, 1, "frustum-culled frame")
    for k in ("triangles_in", "triangles_setup", "fragments_tested", "fragments_written"):
        assert st[k] == rst[k], (k, st[k], rst[k])


def test_device_side_rgb_flatten_equals_host_readback(device):
    """swr_flatten_rgb_device (the payload a multi-GPU frame gathers) == ColorBuffer[..., :3], bit for bit.
    The device buffer comes straight from the HIP runtime the library already loaded (no torch in this process)."""
    hip = C.CDLL("libamdhip64.so")
    s = scenes.cfg2(200, 120, 300, seed=12)
    r = scenes.SceneRenderer(device, s)
    r.submit_frame()
    nbytes = s.height * s.width * 3 * 4
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(nbytes)) == 0
    try:
        r.window.FlattenTo(dptr.value)
        device.sync()
        got = np.empty((s.height, s.width, 3), dtype=np.float32)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), dptr, C.c_size_t(nbytes), 2) == 0      # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(dptr)
    c = r.window.ColorBuffer
    r.close()
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(c[..., :3]).view(np.uint32))


def test_pinned_host_buffer_readback(device):
    """swr_host_register: the same bytes arrive in a page-locked caller buffer; wrong shapes are refused."""
    s = scenes.cfg2(160, 96, 200, seed=13)
    r = scenes.SceneRenderer(device, s)
    r.submit_frame()
    ref = r.window.FlatColorBuffer()
    out = np.zeros_like(ref)
    device.pin(out)
    try:
        r.submit_frame()
        got = r.window.FlatColorBuffer(out=out)
        assert got is out and np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    finally:
        device.unpin(out)
    with pytest.raises(ValueError):
        r.window.FlatColorBuffer(out=np.zeros((3, 3, 3), dtype=np.float32))
    r.close()


def test_exact_division_core_matches_ieee_division(device):
    """The raster kernel divides by per-pair / per-draw denominators through the division's own mul + 4 fma core on a
    staged refined reciprocal (csrc/swr_device.h).  2^31 operand pairs -- random over the guarded range, quotients next to
    rounding midpoints, guard-boundary exponents / significands, renderer-shaped values -- must give the compiler's
    correctly rounded quotient bit for bit; likewise the unscaled sqrt core.  The reciprocal core (1 / d: Interpolate's 1 / sum,
    the normal's 1 / length) is checked on EVERY float d with |d| in [2^-40, 2^83], both signs, in each call."""
    recip_operands = 2 * (0x69000000 - 0x2B800000 + 1)
    total = {"divisions": 0, "sqrts": 0}
    for seed in (1, 2):
        r = device.selftest_division(1 << 30, seed)
        assert r["division_mismatches"] == 0, f"n={r['bad_n_bits']:#010x} d={r['bad_d_bits']:#010x} got={r['bad_got_bits']:#010x} want={r['bad_want_bits']:#010x}"
        assert r["sqrt_mismatches"] == 0
        assert r["divisions"] > recip_operands + 900_000_000
        total["divisions"] += r["divisions"] - recip_operands; total["sqrts"] += r["sqrts"]
    assert total["divisions"] > 2_000_000_000 and total["sqrts"] > 2_000_000_000
    r = device.selftest_division(0, 3)                      # no sampled pairs: the exhaustive reciprocal sweep alone
    assert r["divisions"] == recip_operands and r["division_mismatches"] == 0


def test_tile_list_overflow_poisons_the_batch_and_is_replayed_exactly(monkeypatch):
    """Counters::overflow is no longer write-only: a list position beyond the capacity (forced here with the debug hook
    SWR_DEBUG_FILL_CAPACITY, which makes k_bin<FILL> of an optimistic flush see a shorter list than COUNT was checked
    against) poisons the batch before its raster kernel runs, and the host replays it synchronously: same frame, same counters."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    monkeypatch.setenv("SWR_DEBUG_FILL_CAPACITY", "500")
    dev = Device(0, lib="libswr_hip_test.so")        # the hook exists in the test build only (-DSWR_TEST_HOOKS): the product reads no such variable
    monkeypatch.delenv("SWR_DEBUG_FILL_CAPACITY")
    try:
        scene = scenes.cfg3(320, 240, (3, 3), (20, 14), tex_size=64, seed=21)
        o = OracleRenderer(scene.width, scene.height)
        rc, rd = o.render_scene(scene)
        rst = o.stats()
        r = scenes.SceneRenderer(dev, scene)
        r.render()                                   # synchronous flush: sizes the pair buffers (the hook does not apply)
        dev.reset_stats()
        c, d = r.render()                            # optimistic flush: FILL overflows its 500 entries -> poison -> replay
        st = dev.stats()
        r.close()
        assert_frame_parity(c, d, rc, rd, 1, "fill overflow replay")
        for k in ("triangles_in", "triangles_setup", "fragments_tested", "fragments_shaded", "fragments_written"):
            assert st[k] == rst[k], (k, st[k], rst[k])
    finally:
        dev.close()


def test_more_than_2_pow_20_tiles_is_refused_not_overrun(device):
    """the tile-count scan holds 4096 block sums of 256 tiles; a larger unbanded target used to write past the block sums.
    Now the draw is refused (SWR_ERR_UNSUPPORTED) and the same target renders in tile-row bands."""
    from softwarerenderer_amd._native import SwrError, SWR_ERR_UNSUPPORTED
    scene = scenes.cfg1()
    win = MainWindow(device, 16400, 16400)              # 1025 x 1025 tiles
    d0 = scene.draws[0]
    prog = scenes.ShaderProgram(d0.program, d0.uniforms, None)
    Rasterizer.RenderMesh(win, d0.vertices, d0.indices, d0.model, d0.view, d0.projection, prog.VertexShader, prog.FragmentShader,
                          d0.cull, d0.depth_test, d0.blend)
    with pytest.raises(SwrError) as e:
        device.sync()
    assert e.value.code == SWR_ERR_UNSUPPORTED
    win.SetBand(256, 512)                                # the middle half: 1025 x 512 tiles
    win.ClearColorBuffer((0, 0, 0, 1)); win.ClearDepthBuffer()
    Rasterizer.RenderMesh(win, d0.vertices, d0.indices, d0.model, d0.view, d0.projection, prog.VertexShader, prog.FragmentShader,
                          d0.cull, d0.depth_test, d0.blend)
    device.sync()
    assert device.stats()["fragments_written"] > 0
    px = win.GetPixel(8200, 8200)                        # centre of the frame, inside the triangle and the band
    assert px[3] == 1.0 and px[:3].sum() > 0
    MainWindow(device, 64, 64)                           # release the 5 GB target


def test_band_union_with_large_cancelling_translations(device):
    """band_rejects projected the mesh box in double precision with a fixed 2-pixel margin, but the device (like
    the reference) chains three float32 products; with model and view translations of 1e5 that cancel, float32 screen positions
    move by many pixels.  The margin now carries the float32 error bound, so the union of the bands still equals the frame
    rendered in one piece."""
    import softwarerenderer_amd.hostmath as hm
    scene = scenes.cfg3(300, 270, (3, 3), (20, 14), tex_size=64, seed=8)
    t = 1.0e5
    for d in scene.draws:
        d.model = hm.multiply(d.model, hm.create_translation(t, -t, t))
        d.view = hm.multiply(hm.create_translation(-t, t, -t), d.view)
    whole = scenes.SceneRenderer(device, scene)
    c0, d0 = whole.render()
    whole.close()
    assert (d0 != np.float32(-3.40282347e+38)).any()
    for world in (2, 5):
        cols, deps = [], []
        for band in multigpu.band_partition(scene.height, world):
            win = MainWindow(device, scene.width, scene.height)
            win.SetBand(*band)
            r = scenes.SceneRenderer(device, scene, window=win)
            c, d = r.render()
            r.close()
            cols.append(c); deps.append(d)
        assert np.array_equal(np.concatenate(deps).view(np.uint32), d0.view(np.uint32))
        assert np.array_equal(np.concatenate(cols).view(np.uint32), c0.view(np.uint32))


@pytest.mark.parametrize("world,k", [(2, 1), (3, 2), (4, 3)])
def test_interleaved_stripes_union_is_the_single_gpu_frame(device, world, k):
    """swr_set_band_interleaved: stripes of k tile rows dealt round-robin to `world` ranks (load balance for clustered scenes,
    SURVEY.md section 8e).  Each rank's window renders its stripes; scattering them back gives the frame of the oracle."""
    scene = scenes.cfg3(300, 270, (3, 3), (20, 14), tex_size=64, seed=8)
    rc, rd, ost = render_oracle(scene)
    cols, deps, frag = [], [], 0
    for rank in range(world):
        win = MainWindow(device, scene.width, scene.height)
        win.SetBandInterleaved(rank, world, k)
        device.reset_stats()
        r = scenes.SceneRenderer(device, scene, window=win)
        c, d = r.render()
        frag += device.stats()["fragments_written"]
        r.close()
        cols.append(c); deps.append(d)
        # accessors address the interleaved layout too
        rows = multigpu.stripe_rows(scene.height, world, k)[rank]
        y = int(rows[len(rows) // 2])
        assert np.array_equal(win.GetPixel(5, y), c[len(rows) // 2, 5])
        other = next(yy for yy in range(scene.height) if yy not in set(rows.tolist()))
        assert win.GetDepth(5, other) == MINVAL and not win.GetPixel(5, other).any()          # out of band = out of bounds
    MainWindow(device, scene.width, scene.height).SetBand(-1, -1)
    c = multigpu.assemble_stripes(cols, scene.height, world, k)
    d = multigpu.assemble_stripes(deps, scene.height, world, k)
    assert_frame_parity(c, d, rc, rd, 1, f"stripes world={world} k={k}")
    assert frag == ost["fragments_written"]


def test_async_flatten_and_deferred_validation():
    """The multi-GPU frame loop's contract (bench.py, N > 1): flatten without validating, consume in stream order, validate later.
    swr_replay_count tells the caller that a batch was replayed, i.e. that a payload flattened before the validating swr_sync is
    stale; flattening again after it gives the frame of the oracle."""
    import torch
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    device = Device(0)                                                # a fresh context: its pair buffers start empty
    small = scenes.cfg2(256, 256, 40, seed=60, min_area=10.0, max_area=60.0)
    big = scenes.cfg2(256, 256, 3000, seed=61, min_area=200.0, max_area=9000.0)
    o = OracleRenderer(256, 256)
    rc, _ = o.render_scene(big)
    r0 = scenes.SceneRenderer(device, small)
    r0.render()                                                       # synchronous sizing of the pair buffers (small)
    rgb = torch.zeros((256, 256, 3), dtype=torch.float32, device="cuda")
    before = device.replay_count()
    r1 = scenes.SceneRenderer(device, big, window=r0.window)
    r1.submit_frame()                                                 # optimistic flush that does not fit -> poisoned on the device
    r0.window.FlattenToAsync(rgb.data_ptr())                          # no validation, no wait: flattens the UNCHANGED framebuffer
    device.sync()                                                     # validates: replays the batch
    assert device.replay_count() == before + 1                        # ... and says so
    stale = rgb.cpu().numpy().copy()
    r0.window.FlattenToAsync(rgb.data_ptr())                          # the caller's reaction: flatten (and send) again
    device.sync()
    assert device.replay_count() == before + 1
    fresh = rgb.cpu().numpy()
    assert ulp_distance(fresh, rc[..., :3]).max() <= 1
    assert not np.array_equal(stale, fresh)                           # the first payload really was stale
    # steady state: the next frame fits, nothing is replayed, one flatten suffices
    r1.submit_frame()
    r0.window.FlattenToAsync(rgb.data_ptr())
    device.sync()
    assert device.replay_count() == before + 1
    assert ulp_distance(rgb.cpu().numpy(), rc[..., :3]).max() <= 1
    r0.close(); r1.close()
    device.close()


def test_bind_framebuffer_neither_waits_nor_resets_the_counters():
    """the double-buffered frame loop of bench.py binds another band buffer every frame.  That must be ONE
    swr_bind_framebuffer -- no flush-and-wait (swr_sync_count unchanged), no restart of the fragment counters -- and a resize / band
    call that changes nothing must not wait either."""
    import torch
    from softwarerenderer_amd import Device, MainWindow
    device = Device(0)
    scene = scenes.cfg2(256, 256, 200, seed=77, min_area=20.0, max_area=400.0)
    win = MainWindow(device, 256, 256)
    bufs = [(torch.zeros((256, 256, 4), dtype=torch.float32, device="cuda"), torch.zeros((256, 256), dtype=torch.float32, device="cuda"))
            for _ in range(2)]
    win.BindFramebuffer(bufs[0][0].data_ptr(), bufs[0][1].data_ptr())
    r = scenes.SceneRenderer(device, scene, window=win)
    r.submit_frame(); device.sync()                                   # sizes the pair buffers
    device.reset_stats()
    one = None
    s0 = device.sync_count()
    for i in range(6):
        win.BindFramebuffer(bufs[i & 1][0].data_ptr(), bufs[i & 1][1].data_ptr())
        win.Resize(256, 256)                                          # unchanged geometry: no-ops on the C side
        win.SetBand(-1, -1)
        r.submit_frame(); device.flush()
        assert device.sync_count() == s0, "BindFramebuffer / unchanged Resize / SetBand / flush made the host wait"
    st = device.stats()                                               # (this one validates and waits)
    device.reset_stats()
    r.submit_frame(); device.flush()
    one = device.stats()
    assert st["fragments_tested"] == 6 * one["fragments_tested"] > 0
    assert st["fragments_written"] == 6 * one["fragments_written"]
    # counters also survive a real change of geometry (they move into the carry words)
    device.reset_stats()
    r.submit_frame(); device.flush()
    win.SetBand(0, 8)                                                 # upper half only from now on
    r.submit_frame(); device.flush()
    both = device.stats()
    assert one["fragments_tested"] < both["fragments_tested"] < 2 * one["fragments_tested"]
    r.close(); device.close()


def test_build_identity_matches_the_verified_pair():
    """the lane-to-lane LDS hand-offs of k_cover / k_raster_c carry no fence (every fence form costs 25 %); they are
    verified per (compiler, kernel source) pair by the parity suite and the sweeps.  The library says which pair it is
    (swr_build_info) and profiles/verified_build.json holds the pair the sweeps of this round ran on."""
    from softwarerenderer_amd import _native
    if os.environ.get("SWR_DEV_BUILD") == "1":
        pytest.skip("development A/B build (tools/ab/*.sh): the verified pair is recorded for the round's final build only")
    info = _native.load().swr_build_info().decode()
    fields = dict(kv.split("=", 1) for kv in info.split("; "))
    assert set(fields) == {"hipcc", "csrc_sha256", "fma", "dot", "extra"} and len(fields["csrc_sha256"]) == 64
    assert fields["extra"] == "", "an A/B variant (make EXTRA=...) sits in the product library's place: rebuild it"
    with open(os.path.join(ROOT, "profiles", "verified_build.json")) as f:
        ver = json.load(f)
    assert fields["hipcc"] == ver["hipcc"], "built with another compiler than the one the LDS hand-offs were verified with"
    assert fields["csrc_sha256"] == ver["csrc_sha256"], "kernel sources changed since the sweeps ran: re-run tools/parity_sweep*.py and tools/record_verified_build.py"
    if not os.environ.get("SWR_LIB"):
        assert (fields["fma"], fields["dot"]) == ("0", "0")


def test_asynchronous_present_overlaps_and_delivers_exact_frames():
    """swr_present_rgb_async / swr_present_wait -- the present payload (Vector4 -> Vector3 flatten, MainWindow.cs:234-240)
    of frame i crosses PCIe on the context's copy stream while frame i + 1 renders.  Two frames in flight, each array must hold
    exactly its own frame; nothing on the way makes the host wait except swr_present_wait itself."""
    from softwarerenderer_amd import Device
    dev = Device(0)
    a = scenes.cfg3(512, 384, (2, 2), (30, 20), tex_size=64, seed=91)
    b = scenes.cfg2(512, 384, 1500, seed=92)
    ra = scenes.SceneRenderer(dev, a)
    rb = scenes.SceneRenderer(dev, b, window=ra.window)
    want_a = ra.render()[0][..., :3].copy()                           # synchronous frames first: sizes the pair buffers too
    want_b = rb.render()[0][..., :3].copy()
    bufs = [np.zeros((384, 512, 3), dtype=np.float32) for _ in range(2)]
    for x in bufs:
        dev.pin(x)
    try:
        s0 = dev.sync_count()
        ra.submit_frame(); t0 = ra.window.PresentAsync(bufs[0])
        rb.submit_frame(); t1 = ra.window.PresentAsync(bufs[1])       # frame b renders and is flattened while frame a is copied
        assert dev.sync_count() == s0                                 # neither present made the host wait for the stream
        assert ra.window.PresentWait(t0) and ra.window.PresentWait(t1)
        assert np.array_equal(bufs[0].view(np.uint32), want_a.view(np.uint32))
        assert np.array_equal(bufs[1].view(np.uint32), want_b.view(np.uint32))
        # a steady loop: present i, wait i - 1; every array ends up with the frame that was current at its present
        tickets = [None, None]
        for i in range(6):
            (ra if i % 2 == 0 else rb).submit_frame()
            k = i & 1
            if tickets[k] is not None:
                assert ra.window.PresentWait(tickets[k])
                assert np.array_equal(bufs[k].view(np.uint32), (want_a if k == 0 else want_b).view(np.uint32))
            tickets[k] = ra.window.PresentAsync(bufs[k])
        for k in (0, 1):
            assert ra.window.PresentWait(tickets[k])
        assert ra.window.PresentWait(tickets[0])                      # waiting twice for a ticket is harmless
    finally:
        for x in bufs:
            dev.unpin(x)
    ra.close(); rb.close(); dev.close()


def test_asynchronous_present_reports_a_stale_frame_after_a_replay():
    """The optimistic flush of a batch that does not fit poisons itself; an asynchronous present enqueued behind it copies the
    UNCHANGED framebuffer.  swr_present_wait must say so (SWR_STALE -> False) after replaying, and presenting again gives the frame."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    dev = Device(0)                                                   # a fresh context: its pair buffers start empty
    small = scenes.cfg2(256, 256, 40, seed=60, min_area=10.0, max_area=60.0)
    big = scenes.cfg2(256, 256, 3000, seed=61, min_area=200.0, max_area=9000.0)
    o = OracleRenderer(256, 256)
    rc, _ = o.render_scene(big)
    o.close()
    r0 = scenes.SceneRenderer(dev, small)
    r0.render()                                                       # synchronous sizing of the pair buffers (small)
    r1 = scenes.SceneRenderer(dev, big, window=r0.window)
    out = np.zeros((256, 256, 3), dtype=np.float32)
    before = dev.replay_count()
    r1.submit_frame()                                                 # does not fit: poisoned on the device
    t = r0.window.PresentAsync(out)
    assert r0.window.PresentWait(t) is False                          # stale, and the batch has been replayed by now
    assert dev.replay_count() == before + 1
    t = r0.window.PresentAsync(out)                                   # the caller's reaction: present again
    assert r0.window.PresentWait(t) is True
    assert ulp_distance(out, rc[..., :3]).max() <= 1
    r0.close(); r1.close(); dev.close()


def test_overflow_replay_while_the_next_frames_front_end_is_already_running():
    """with frames in flight the front end of flush N+1 runs beside the raster kernel of flush N, so a batch can poison
    itself while an EARLIER batch is still rasterising and while LATER batches' front ends are queued or running.  Four order-dependent
    frames back to back, no synchronisation in between: fits, fits (big: its raster kernel is long), does NOT fit, fits -- the bad one's
    own kernels and every later batch must leave the framebuffer alone (batch_poisoned goes by first_bad, not by the sticky flag: the
    running raster kernel of the good batch must finish ALL its tiles), and the host replays from the bad one exactly."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    dev = Device(0)
    assert dev.pipelining() == 1                                   # the default
    W = H = 512
    sizing = scenes.cfg2(W, H, 2500, seed=70, min_area=100.0, max_area=4000.0)       # sizes the pair buffers (both sets)
    good = scenes.cfg2(W, H, 2400, seed=71, min_area=100.0, max_area=4000.0)         # fits, and rasterises for a while
    bad = scenes.state_scene(W, H, 9000, seed=72, blend=BlendMode.Additive)          # far more pairs than the buffers hold
    after = scenes.state_scene(W, H, 900, seed=73, blend=BlendMode.Multiply)
    for s_ in (bad, after):
        s_.clear_color = None; s_.clear_depth = False              # accumulate: every frame's pixels depend on all frames before
    o = OracleRenderer(W, H)
    o.render_scene(sizing); o.render_scene(sizing); o.reset_stats()
    o.render_scene(good); o.render_scene(bad)
    rc, rd = o.render_scene(after)
    rst = o.stats(); o.close()

    r0 = scenes.SceneRenderer(dev, sizing)
    r0.render(); r0.render()                                       # synchronous sizing, then one optimistic frame: both raster sets exist
    dev.reset_stats()
    rs = [scenes.SceneRenderer(dev, s_, window=r0.window) for s_ in (good, bad, after)]
    replays = dev.replay_count()
    for r in rs:
        r.submit_frame(); dev.flush()                              # three flushes, the host never waits
    c, d = r0.window._read()                                       # sync point: validate + replay from the bad batch
    assert dev.replay_count() == replays + 1
    st = dev.stats()
    assert_frame_parity(c, d, rc, rd, 1, "replay with frames in flight")
    for k in ("triangles_in", "triangles_setup", "fragments_tested", "fragments_shaded", "fragments_written"):
        assert st[k] == rst[k], (k, st[k], rst[k])
    # and the same sequence again on the grown buffers: no replay this time, same pixels
    dev.reset_stats()
    rs[0].submit_frame(); dev.flush(); rs[1].submit_frame(); dev.flush(); rs[2].submit_frame(); dev.flush()
    o = OracleRenderer(W, H)
    o.upload(rc, rd)
    o.render_scene(good); o.render_scene(bad)
    rc2, rd2 = o.render_scene(after); o.close()
    c2, d2 = r0.window._read()
    assert dev.replay_count() == replays + 1
    assert_frame_parity(c2, d2, rc2, rd2, 1, "frames in flight, grown buffers")
    for r in [r0] + rs:
        r.close()
    dev.close()


def test_pipelining_modes_render_the_same_frames():
    """swr_set_pipelining: frames in flight (1, default; 2 = front stream at default priority) and one stream (0) differ in scheduling
    only -- a sequence of order-dependent frames gives the same words in every mode, and switching in the middle is safe."""
    from softwarerenderer_amd import Device
    dev = Device(0)
    a = scenes.cfg3(640, 480, (3, 3), (30, 20), tex_size=64, seed=81)
    b = scenes.state_scene(640, 480, 1500, seed=82, blend=BlendMode.Alpha)
    b.clear_color = None; b.clear_depth = False
    ra = scenes.SceneRenderer(dev, a)
    rb = scenes.SceneRenderer(dev, b, window=ra.window)
    frames = {}
    for mode in (1, 0, 2, 1):
        dev.set_pipelining(mode)
        assert dev.pipelining() == mode
        for _ in range(3):
            ra.submit_frame(); dev.flush(); rb.submit_frame(); dev.flush()
        c, d = ra.window._read()
        frames.setdefault("c", c); frames.setdefault("d", d)
        assert np.array_equal(c.view(np.uint32), frames["c"].view(np.uint32)) and np.array_equal(d.view(np.uint32), frames["d"].view(np.uint32)), mode
    with pytest.raises(Exception):
        dev.set_pipelining(3)
    ra.close(); rb.close(); dev.close()


def test_big_and_small_batches_alternate_between_one_stream_and_frames_in_flight():
    """Mode 2 pipelines small frames (<= 2^15 tiles) and small batches (<= 2^17 triangles) and runs the rest on the context's stream
    alone; both kinds share the front-end-only buffers, so the hand-over between the two placements is ordered by events in both
    directions.  An order-dependent sequence small, BIG, small, BIG, small without any synchronisation on a 3072^2 target (36,864
    tiles) must give the oracle's frame; so must mode 1 (default: every batch pipelined) and mode 0 (none)."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    W, H = 3072, 3072
    big = scenes.cfg3(W, H, (2, 2), (190, 95), tex_size=64, seed=95)              # 144,400 triangles > 2^17
    assert big.n_triangles > (1 << 17) and (W // 16) * (H // 16) > (1 << 15)
    small = scenes.state_scene(W, H, 1200, seed=96, blend=BlendMode.Alpha)
    small2 = scenes.state_scene(W, H, 900, seed=97, blend=BlendMode.Additive)
    for s_ in (small, small2, big):
        s_.clear_color = None; s_.clear_depth = False
    first = scenes.cfg2(W, H, 800, seed=98)                                         # clears
    seq = [first, big, small, big, small2, big, small]
    o = OracleRenderer(W, H)
    for s_ in seq:
        rc, rd = o.render_scene(s_)
    o.close()
    for mode in (1, 2, 0):
        dev = Device(0)
        dev.set_pipelining(mode)
        rs = {}
        win = None
        for s_ in (first, big, small, small2):
            rs[id(s_)] = scenes.SceneRenderer(dev, s_, window=win)
            win = rs[id(s_)].window
        for s_ in seq:                                       # sizes the buffers (first round synchronous), then the real thing
            rs[id(s_)].submit_frame(); dev.flush()
        dev.sync()
        syncs = dev.sync_count()
        for s_ in seq:
            rs[id(s_)].submit_frame(); dev.flush()
        assert dev.sync_count() == syncs, mode             # neither placement makes the host wait
        c, d = win._read()
        assert_frame_parity(c, d, rc, rd, 1, f"mixed batch sizes, pipelining mode {mode}")
        for r in rs.values():
            r.close()
        dev.close()


def test_present_loop_over_array_draws_never_waits_for_the_next_frame():
    """swr_render_mesh_arrays -- the reference's RenderMesh(vertices, indices, ...) -- makes a transient mesh per
    call; swr_present_wait used to hipFree the retired ones, which synchronises the device, i.e. waits for frame i + 1.  Now they are
    recycled: a present loop over array draws allocates nothing in steady state, never drains (swr_sync_count stays), and every
    delivered frame is exact."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    dev = Device(0)
    scene = scenes.cfg3(384, 256, (2, 2), (24, 16), tex_size=64, seed=93)
    o = OracleRenderer(scene.width, scene.height)
    rc, _ = o.render_scene(scene); o.close()
    r = scenes.SceneRenderer(dev, scene, retained=False)            # every draw goes through swr_render_mesh_arrays
    outs = [np.zeros((scene.height, scene.width, 3), dtype=np.float32) for _ in range(2)]
    for a in outs:
        dev.pin(a)
    r.submit_frame(); t_prev = r.window.PresentAsync(outs[0])       # frame 0 (synchronous sizing happens here)
    r.submit_frame(); t_cur = r.window.PresentAsync(outs[1])
    assert r.window.PresentWait(t_prev) is True
    syncs = dev.sync_count()
    for i in range(2, 14):
        r.submit_frame()
        t_next = r.window.PresentAsync(outs[i & 1])                 # overwrites the buffer whose ticket was waited for last round
        assert r.window.PresentWait(t_cur) is True                  # frame i - 1 arrives while frame i renders
        assert ulp_distance(outs[(i - 1) & 1], rc[..., :3]).max() <= 1
        t_cur = t_next
    assert r.window.PresentWait(t_cur) is True
    assert dev.sync_count() == syncs                                # nothing in the loop drained the stream (or freed: frees drain)
    for a in outs:
        dev.unpin(a)
    r.close(); dev.close()


def test_a_frame_presented_before_a_later_overflow_is_not_reported_stale():
    """SWR_STALE goes by WHICH batch poisoned itself (Ctrl::first_bad against the batches the present covers), not by the
    global flag: frame A is presented, frame B (flushed after the present) overflows -- A's pixels are good and must be delivered
    as such; B is replayed at the next synchronisation point."""
    from oracle.binding import OracleRenderer
    from softwarerenderer_amd import Device
    dev = Device(0)
    small = scenes.cfg2(256, 256, 60, seed=64, min_area=10.0, max_area=80.0)
    big = scenes.cfg2(256, 256, 3000, seed=65, min_area=200.0, max_area=9000.0)
    o = OracleRenderer(256, 256)
    ra_, _ = o.render_scene(small)
    rb_, rbd = o.render_scene(big)
    o.close()
    r0 = scenes.SceneRenderer(dev, small)
    r0.render()                                                       # synchronous sizing (small buffers)
    r1 = scenes.SceneRenderer(dev, big, window=r0.window)
    out = np.zeros((256, 256, 3), dtype=np.float32)
    before = dev.replay_count()
    r0.submit_frame()
    t = r0.window.PresentAsync(out)                                   # frame A: fits
    r1.submit_frame(); dev.flush()                                    # frame B: flushed AFTER the present, does not fit
    assert r0.window.PresentWait(t) is True                           # A is good: not stale, no replay forced here
    assert ulp_distance(out, ra_[..., :3]).max() <= 1
    c, d = r0.window._read()                                          # sync point: B is replayed now
    assert dev.replay_count() == before + 1
    assert_frame_parity(c, d, rb_, rbd, 1, "late overflow")
    r0.close(); r1.close(); dev.close()


def test_a_single_draw_beyond_the_batch_limit_is_refused_when_it_is_recorded():
    """one batch holds fewer than 2^26 vertex-stage records (vertices + 4 x triangles).  Several draws are flushed in time; a
    SINGLE mesh beyond the limit used to reach execute_batch and fail there with advice that could not help ("flush more often").  It
    is refused by swr_render_mesh itself now, with what to do, and the context stays usable."""
    from softwarerenderer_amd import Device
    from softwarerenderer_amd._native import SwrError, SWR_ERR_UNSUPPORTED
    device = Device(0)                                       # its own context: the accepted draw below leaves multi-GB buffers behind
    n_tris = (1 << 24) + 8                                   # 4 x n_tris alone reaches 2^26
    v = scenes.make_vertices([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    idx = np.zeros(3 * n_tris, dtype=np.uint16)              # degenerate triangles: only their number matters
    win = MainWindow(device, 64, 64)
    prog = scenes.ShaderProgram(scenes.Program.FlatColor, None, None)
    I = np.eye(4, dtype=np.float32)
    with pytest.raises(SwrError) as e:
        Rasterizer.RenderMesh(win, v, idx, I, I, I, prog.VertexShader, prog.FragmentShader)
    assert e.value.code == SWR_ERR_UNSUPPORTED and "2^26" in str(e.value) and "split" in str(e.value)
    # one triangle fewer than the limit allows is recorded (and rendered: every triangle is degenerate, nothing is written)
    ok = np.zeros(3 * ((1 << 24) - 4), dtype=np.uint16)
    Rasterizer.RenderMesh(win, v, ok, I, I, I, prog.VertexShader, prog.FragmentShader)
    device.sync()
    assert device.stats()["triangles_in"] >= (1 << 24) - 4
    device.close()
