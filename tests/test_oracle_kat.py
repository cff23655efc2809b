"""Known-answer tests that pin the CPU oracle WITHOUT the reference (which ships no tests or vectors and
cannot be built here: SURVEY.md section 8c).  Every expected value below is derived by hand / exact
arithmetic from the reference's formulas (file:line cited), not produced by the oracle itself."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from softwarerenderer_amd import hostmath as hm, scenes
from softwarerenderer_amd.rasterizer import BlendMode, CullMode, DepthTest, Program

MINVAL = np.float32(-3.4028235e38)      # float.MinValue, MainWindow.cs:434


def cfg1_mask():
    """Analytic coverage of cfg1: screen vertices (64,192),(192,192),(128,64) (Rasterizer.cs:383-386), all
    arithmetic exact in float32; a pixel (integer sample point, no +0.5, no top-left rule: :493-494) is covered
    iff 64 <= y <= 192 and 2*|x-128| <= y-64."""
    y, x = np.mgrid[0:256, 0:256]
    return (y >= 64) & (y <= 192) & (2 * np.abs(x - 128) <= (y - 64))


def test_cfg1_coverage_colour_and_untouched_depth(oracle_lib):
    s = scenes.cfg1()
    o = ob.OracleRenderer(s.width, s.height)
    color, depth = o.render_scene(s)
    mask = cfg1_mask()
    assert mask.sum() == 8321            # sum_{k=0..128} (2*floor(k/2)+1)
    # flat (Interpolate=false): attributes come from outputs[0] = original v2 (Rasterizer.cs:367,622-627) = blue
    assert np.array_equal(color[mask], np.tile(np.float32([0, 0, 1, 1]), (8321, 1)))
    assert np.array_equal(color[~mask], np.tile(np.float32([0, 0, 0, 1]), ((~mask).sum(), 1)))
    # DepthTest.Disabled never writes Z (Rasterizer.cs:517-518): still the clear value
    assert np.all(depth == MINVAL)
    st = o.stats()
    assert st["fragments_tested"] == st["fragments_written"] == 8321 and st["triangles_setup"] == 1


def test_blend_none_early_out_breaks_the_row_within_a_tile(oracle_lib):
    """canEarlyOut (Rasterizer.cs:520-523): alpha 0 + BlendMode.None -> the first covered pixel of each row of
    each 16x16 tile is shaded, fails `W > 0`, and ends that row segment.  Nothing is written."""
    s = scenes.cfg1()
    s.draws[0].vertices["color"][:, 3] = 0.0
    s.draws[0].blend = BlendMode.None_
    o = ob.OracleRenderer(s.width, s.height)
    color, _ = o.render_scene(s)
    assert np.array_equal(color, np.tile(np.float32([0, 0, 0, 1]), (256, 256, 1)))
    mask = cfg1_mask()
    segments = sum(int(mask[y, tx * 16:(tx + 1) * 16].any()) for y in range(256) for tx in range(16))
    st = o.stats()
    assert st["fragments_shaded"] == segments and st["fragments_written"] == 0


def test_depth_function_names_are_inverted_as_written(oracle_lib):
    f = oracle_lib.oswr_depth_func
    # GetDepthTestFunction, Rasterizer.cs:543-559: LessEqual -> new >= old, Less -> new > old, ...
    assert f(DepthTest.LessEqual, 0.5, 0.25) == 1 and f(DepthTest.LessEqual, 0.25, 0.5) == 0 and f(DepthTest.LessEqual, 0.5, 0.5) == 1
    assert f(DepthTest.Less, 0.5, 0.5) == 0 and f(DepthTest.Less, 0.75, 0.5) == 1
    assert f(DepthTest.Greater, 0.25, 0.5) == 1 and f(DepthTest.Greater, 0.5, 0.5) == 0
    assert f(DepthTest.GreaterEqual, 0.5, 0.5) == 1 and f(DepthTest.GreaterEqual, 0.75, 0.5) == 0
    assert f(DepthTest.Equal, 0.5, 0.5 + 5e-7) == 1 and f(DepthTest.Equal, 0.5, 0.5 + 2e-6) == 0
    assert f(DepthTest.NotEqual, 0.5, 0.5 + 2e-6) == 1 and f(DepthTest.NotEqual, 0.5, 0.5) == 0
    assert f(DepthTest.Disabled, 0.0, 1.0) == 1 and f(DepthTest.Always, 0.0, 1.0) == 1
    assert f(DepthTest.LessEqual, 0.0, float(MINVAL)) == 1       # everything beats the cleared buffer


def test_nearer_fragment_wins_and_ties_go_to_the_later_triangle(oracle_lib):
    """Hand derivation (corrects SURVEY.md fact 1).  With s0=(0,0), s1=(1,0), s2=(0,1):
    area = EdgeFunction(s0,s1,s2) = -1 (Rasterizer.cs:411,562) while w0 at p=s0 is
    a12*(0-s1.x) + b12*(0-s1.y) = (-1)(-1) + 0 = +1 (Rasterizer.cs:445-447,481): each edge weight at its own
    vertex is -area, so w0f+w1f+w2f = -1 and the stored depth is MINUS the interpolated (ndc.z+1)/2.
    (ndc.z+1)/2 grows with distance, its negation shrinks, and LessEqual is `new >= old` (:545-546):
    the NEARER fragment wins; equal depths -> the later triangle overwrites (>=)."""
    proj = hm.create_perspective_fov(np.pi / 2, 1.0, 0.1, 1000.0)
    I = hm.identity()

    def tri(z, rgb):
        p = [(-4 * z, -4 * z, -z), (4 * z, -4 * z, -z), (0, 4 * z, -z)]
        return scenes.make_vertices(p, color=[rgb + (1.0,)] * 3)

    def render(order):
        v = np.concatenate([tri(z, rgb) for z, rgb in order])
        d = scenes.Draw(v, np.arange(len(v), dtype=np.uint16), I, I, proj, program=Program.Gouraud, cull=CullMode.None_)
        o = ob.OracleRenderer(32, 32)
        c, dz = o.render_scene(scenes.Scene("t", 32, 32, [d]))
        return tuple(c[16, 16, :3]), float(dz[16, 16])

    near, far = (2.0, (1.0, 0.0, 0.0)), (5.0, (0.0, 1.0, 0.0))
    for order in ([near, far], [far, near]):
        rgb, z = render(order)
        assert rgb == (1.0, 0.0, 0.0)                      # near (red) wins in either order
        # ndc.z at z_view=-2: (2*1.0001 - 0.10001)/2 = 0.95005 -> (ndc.z+1)/2 = 0.975025 ; stored NEGATED
        assert abs(z + 0.975025) < 1e-4
    same_a, same_b = (3.0, (1.0, 0.0, 0.0)), (3.0, (0.0, 0.0, 1.0))
    assert render([same_a, same_b])[0] == (0.0, 0.0, 1.0)  # tie: the later triangle overwrites
    assert render([same_b, same_a])[0] == (1.0, 0.0, 0.0)


def test_blend_formulas(oracle_lib):
    def blend(src, dst, mode):
        s, d, o = np.float32(src), np.float32(dst), np.zeros(4, np.float32)
        oracle_lib.oswr_blend(s.ctypes.data, d.ctypes.data, int(mode), o.ctypes.data)
        return tuple(o)
    src, dst = (0.5, 0.25, 1.0, 0.5), (1.0, 1.0, 0.0, 1.0)
    assert blend(src, dst, BlendMode.None_) == src
    assert blend(src, dst, BlendMode.Alpha) == (0.75, 0.625, 0.5, 0.75)      # src*a + dst*(1-a) on all four channels
    assert blend(src, dst, BlendMode.Additive) == (1.0, 1.0, 1.0, 1.0)       # min(src+dst, 1)
    assert blend((0.25, 0.25, 0.25, 0.25), (0.5, 0.5, 0.5, 0.5), BlendMode.Additive) == (0.75,) * 4
    assert blend(src, dst, BlendMode.Multiply) == (0.5, 0.25, 0.0, 0.5)


def test_texture_sample_nearest_wrap(oracle_lib):
    tex = np.arange(2 * 2 * 4, dtype=np.uint8).reshape(2, 2, 4) * 10     # texel (x,y) = tex[y, x]
    inv255 = np.float32(1.0) / np.float32(255.0)

    def sample(u, v):
        uv, out = np.float32([u, v]), np.zeros(4, np.float32)
        oracle_lib.oswr_texture_sample(tex.ctypes.data, 2, 2, uv.ctypes.data, out.ctypes.data)
        return out

    def texel(x, y):
        return tex[y, x].astype(np.float32) * inv255      # byte * (1f/255f), Texture.cs:57-62

    assert np.array_equal(sample(0.25, 0.25), texel(0, 0))
    assert np.array_equal(sample(0.75, 0.25), texel(1, 0))
    assert np.array_equal(sample(0.25, 0.75), texel(0, 1))
    assert np.array_equal(sample(-0.25, 0.25), texel(1, 0))   # u - (int)u = -0.25 -> +1 = 0.75
    assert np.array_equal(sample(2.75, -1.75), texel(1, 0))   # wrap: 0.75, 0.25
    assert np.array_equal(sample(1.0, 3.0), texel(0, 0))      # exact integers wrap to 0
    assert np.array_equal(sample(-1e-9, 0.25), texel(0, 0))   # -1e-9 + 1 rounds to 1.0f -> (int)(2.0) % 2 = 0
    # .NET 9 float->int conversions saturate (documented assumption): (int)1e20 = int.MaxValue, MaxValue % 2 = 1
    assert np.array_equal(sample(1e20, 0.25), texel(1, 0))
    assert np.array_equal(sample(float("nan"), 0.25), texel(0, 0))   # (int)NaN = 0


def test_edge_function(oracle_lib):
    def e(a, b, c):
        a, b, c = np.float32(a), np.float32(b), np.float32(c)
        return oracle_lib.oswr_edge_function(a.ctypes.data, b.ctypes.data, c.ctypes.data)
    assert e((0, 0), (1, 0), (0, 1)) == -1.0      # (c.x-a.x)*(b.y-a.y) - (c.y-a.y)*(b.x-a.x), Rasterizer.cs:562-563
    assert e((0, 0), (0, 1), (1, 0)) == 1.0
    assert e((1, 1), (2, 2), (3, 3)) == 0.0


def _vo(clip, color=(0, 0, 0, 0), uv=(0, 0), normal=(0, 0, 0), screen=(0, 0), wn=(0, 0, 1), interp=1):
    v = ob.OVertexOutput()
    v.clip[:] = clip; v.color[:] = color; v.texcoord[:] = uv; v.normal[:] = normal; v.screen[:] = screen
    v.world_normal[:] = wn; v.has_data = 1; v.interpolate = interp
    return v


def test_interpolate_perspective_correct(oracle_lib):
    """Rasterizer.Interpolate, Rasterizer.cs:566-640: r_i = w_i / W_i; attr = ((A*rA + B*rB) + C*rC) * (1 / sum r)."""
    a = _vo((1, 2, 3, 1), color=(1, 0, 0, 1), uv=(0, 0), wn=(1, 0, 0))
    b = _vo((4, 5, 6, 2), color=(0, 1, 0, 1), uv=(1, 0), wn=(0, 1, 0))
    c = _vo((7, 8, 9, 4), color=(0, 0, 1, 0.5), uv=(0, 1), wn=(0, 0, 1))
    out = ob.OVertexOutput()
    oracle_lib.oswr_interpolate(C.byref(a), C.byref(b), C.byref(c), 0.5, 0.25, 0.25, 1, C.byref(out))
    f = np.float32
    ra, rb, rc = f(0.5) / f(1), f(0.25) / f(2), f(0.25) / f(4)        # 0.5, 0.125, 0.0625 (exact)
    w = f(1.0) / ((ra + rb) + rc)
    exp_clip = [((f(x) * ra + f(y) * rb) + f(z) * rc) * w for x, y, z in zip(a.clip, b.clip, c.clip)]
    assert list(out.clip) == exp_clip
    assert list(out.texcoord) == [((f(0) * ra + f(1) * rb) + f(0) * rc) * w, ((f(0) * ra + f(0) * rb) + f(1) * rc) * w]
    assert out.color[3] == ((f(1) * ra + f(1) * rb) + f(0.5) * rc) * w
    wa, wb, wc = ra * w, rb * w, rc * w
    assert list(out.barycentric) == [wa, wb, wc]
    n = np.array([wa, wb, wc], dtype=f)                                # Data: normalised weights, then renormalise (:680-688)
    ln = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]
    s = f(1.0) / np.sqrt(ln)
    assert list(out.world_normal) == [n[0] * s, n[1] * s, n[2] * s]
    # flat (interpolate=false): Normal/Color/Data copied from a; position/uv still interpolated (:622-627)
    oracle_lib.oswr_interpolate(C.byref(a), C.byref(b), C.byref(c), 0.5, 0.25, 0.25, 0, C.byref(out))
    assert list(out.color) == [1, 0, 0, 1] and list(out.world_normal) == [1, 0, 0] and list(out.clip) == exp_clip


def test_shaders_lerp(oracle_lib):
    """Shaders.Lerp, Shaders.cs:50-95 with Vector.Lerp = a*(1-t) + b*t; WorldNormal is NOT renormalised."""
    a = _vo((0, 0, 0, 1), color=(1, 1, 1, 1), uv=(0, 4), wn=(1, 0, 0))
    b = _vo((8, 4, 2, 3), color=(0, 0.5, 1, 0), uv=(2, 0), wn=(0, 1, 0))
    out = ob.OVertexOutput()
    oracle_lib.oswr_lerp(C.byref(a), C.byref(b), 0.25, 1, C.byref(out))
    assert list(out.clip) == [2.0, 1.0, 0.5, 1.5]
    assert list(out.texcoord) == [0.5, 3.0]
    assert list(out.color) == [0.75, 0.875, 1.0, 0.75]
    assert list(out.world_normal) == [0.75, 0.25, 0.0]
    assert out.interpolate == 1 and list(out.screen) == [0.0, 0.0]


def test_vertex_shader_three_chained_transforms(oracle_lib):
    """Renderer.VertexShader, Renderer.cs:830-846: clip = ((pos,1)*model)*view*proj, WorldNormal = Normalize(TransformNormal)."""
    vin = scenes.make_vertices([(1.0, 2.0, 3.0)], uv=[(0.25, 0.75)], normal=[(0.0, 0.0, 2.0)], color=[(0.1, 0.2, 0.3, 0.4)])
    model = hm.create_scale(2.0)
    view = hm.create_translation(10.0, 20.0, 30.0)
    proj = hm.identity(); proj[2, 3] = -1.0                       # w' = w - z
    out = ob.OVertexOutput()
    oracle_lib.oswr_vertex_shader(vin.ctypes.data, model.ctypes.data, view.ctypes.data, proj.ctypes.data,
                                  int(Program.Dust2LambertFog), C.byref(out))
    assert list(out.clip) == [12.0, 24.0, 36.0, 1.0 - 36.0]
    assert list(out.world_normal) == [0.0, 0.0, 1.0]
    assert list(out.texcoord) == [0.25, 0.75] and out.interpolate == 1
    assert np.allclose(list(out.color), [0.1, 0.2, 0.3, 0.4], rtol=0, atol=0) or list(out.color) == list(vin["color"][0])


def test_out_of_range_index_is_an_error(oracle_lib):
    s = scenes.cfg1()
    s.draws[0].indices = np.array([0, 1, 7], dtype=np.uint16)
    o = ob.OracleRenderer(s.width, s.height)
    with pytest.raises(IndexError):
        o.render_scene(s)


def test_fma_variant_is_a_different_build(oracle_lib):
    assert oracle_lib.oswr_numerics_fma() == 0
    assert ob.load(fma=True).oswr_numerics_fma() == 1


def test_bounding_sphere_serial_schedule_and_last_outside_vertex_quirk(oracle_lib):
    """FrustumCuller.CalculateBoundingSphere (FrustumCuller.cs:59-151), hand-computed.
    p0=(0,0,0) -> p1 = farthest from p0 = (4,0,0) -> p2 = farthest from p1 = (0,0,0): centre (2,0,0), r = 2.
    The third loop's per-partition `local` keeps only the LAST vertex outside that sphere; with one partition the single
    update uses (2,3,0) (distance 3): r' = 2.5, centre += (0,3,0) * (0.5/3) = (2,0.5,0) -- although (2,5,0) came first
    and stays outside the result."""
    def sphere(points):
        v = scenes.make_vertices(points)
        out = np.zeros(4, np.float32)
        oracle_lib.oswr_bounding_sphere(v.ctypes.data, v.shape[0], out.ctypes.data)
        return tuple(float(x) for x in out)
    assert sphere([(0, 0, 0), (4, 0, 0)]) == (2.0, 0.0, 0.0, 2.0)
    assert sphere([(0, 0, 0), (4, 0, 0), (2, 3, 0)]) == (2.0, 0.5, 0.0, 2.5)
    # (2,3.4,0) is outside the first sphere too and comes first, but only the LAST outside vertex (2,3,0) is applied:
    # same result as above, and (2,3.4,0) stays outside it (distance 2.9 from the new centre > 2.5)
    assert sphere([(0, 0, 0), (4, 0, 0), (2, 3.4, 0), (2, 3, 0)]) == (2.0, 0.5, 0.0, 2.5)
    assert sphere([(1, 2, 3)]) == (1.0, 2.0, 3.0, 0.0)                               # single vertex (:67-68)
    assert sphere([(1, 2, 3), (1, 2, 3), (1, 2, 3)]) == (1.0, 2.0, 3.0, 0.0)         # nothing farther than 0: p1 = p2 = p0


def test_sphere_in_frustum(oracle_lib):
    """IsSphereInFrustum (FrustumCuller.cs:201-224): camera at the origin looking down -Z, fov 90, aspect 1."""
    proj = hm.create_perspective_fov(np.pi / 2, 1.0, 0.1, 1000.0)
    I = hm.identity()

    def inside(c, r, model=I):
        s = np.float32([c[0], c[1], c[2], r])
        m, v, p = (np.ascontiguousarray(a, dtype=np.float32) for a in (model, I, proj))
        return bool(oracle_lib.oswr_is_sphere_in_frustum(s.ctypes.data, m.ctypes.data, v.ctypes.data, p.ctypes.data))
    assert inside((0, 0, -5), 1)
    assert not inside((0, 0, 5), 1)                 # behind the camera: fails the near plane
    assert inside((0, 0, 0.5), 1)                   # straddles the near plane
    assert not inside((100, 0, -5), 1)              # right of the right plane (x = -z at fov 90)
    assert inside((5.5, 0, -5), 1)                  # distance to the right plane = (5 - 5.5)/sqrt(2) = -0.35 > -1
    assert not inside((7, 0, -5), 1)                # (5 - 7)/sqrt(2) = -1.41 < -1
    # centre and radius both go through the model matrix: scale 0.5 -> centre (3.5,0,-2.5), r 0.5; (2.5-3.5)/sqrt(2) = -0.707 < -0.5
    assert not inside((7, 0, -5), 1, model=hm.create_scale(0.5))
    assert inside((5.5, 0, -5), 1, model=hm.create_scale(0.5))     # (2.5-2.75)/sqrt(2) = -0.18 > -0.5


def _cfg1_dust2(fog_start, fog_end, z):
    """cfg1's triangle (area -2^14 in screen space: every weight is an exact dyadic rational, the weights sum to -1
    and the perspective factor is exactly -1) shaded by Renderer.FragmentShader: normals +Z, light along -Z."""
    s = scenes.cfg1()
    d = s.draws[0]
    d.program = Program.Dust2LambertFog
    d.vertices["normal"] = (0.0, 0.0, 1.0)
    d.vertices["position"][:, 2] = z
    from softwarerenderer_amd.rasterizer import default_uniforms
    u = default_uniforms()
    u.light_direction[:] = (0.0, 0.0, -1.0)
    u.light_color[:] = (1.0, 1.0, 1.0, 1.0)
    u.fog_color[:] = (1.0, 0.5, 0.25, 1.0)
    u.fog_start, u.fog_end = fog_start, fog_end
    d.uniforms = u
    d.texture = None
    return s


def test_fragment_shader_lambert_and_fog_constants(oracle_lib):
    """Renderer.cs:848-860 by hand.  diffuse = max(0.25, dot(n, -L)) = 1, so 0.1f + 0.9f*1 rounds to exactly 1.0f;
    no texture -> white (Renderer.cs:852); base = interpolated vertex colour = the barycentric coordinates."""
    # fog factor 1 (clip z 0, FogStart 0, FogEnd 1): smoothstep(1) = 1 -> lerp(fog, lit, 1) = lit exactly
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(_cfg1_dust2(0.0, 1.0, 0.0))
    assert tuple(color[192, 64]) == (1.0, 0.0, 0.0, 1.0)            # vertex 0 (red)
    assert tuple(color[64, 128]) == (0.0, 0.0, 1.0, 1.0)            # vertex 2 (blue)
    assert tuple(color[128, 128]) == (0.25, 0.25, 0.5, 1.0)         # lambda = (1/4, 1/4, 1/2)
    assert tuple(color[192, 128]) == (0.5, 0.5, 0.0, 1.0)           # middle of the bottom edge
    # fog factor 1/2 (clip z 1, FogStart 0, FogEnd 2): t*t*(3 - 2t) = 1/2 -> the exact mean of fog colour and lit colour
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(_cfg1_dust2(0.0, 2.0, 1.0))
    assert tuple(color[128, 128]) == (0.625, 0.375, 0.375, 1.0)     # ((1, .5, .25) + (.25, .25, .5)) / 2, alpha = base alpha
    # beyond FogEnd: clamp to 0 -> pure fog colour, alpha still the base alpha
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(_cfg1_dust2(0.0, 0.5, 1.0))
    assert tuple(color[128, 128]) == (1.0, 0.5, 0.25, 1.0)


def test_alpha_zero_fragments_neither_blend_nor_write_depth(oracle_lib):
    """Rasterizer.cs:511-518: a shaded fragment with W <= 0 is skipped before the blend AND before the depth write."""
    s = scenes.cfg1()
    d = s.draws[0]
    d.vertices["color"][:, 3] = 0.0
    d.depth_test = DepthTest.LessEqual
    o = ob.OracleRenderer(256, 256)
    color, depth = o.render_scene(s)
    assert np.array_equal(color, np.tile(np.float32([0, 0, 0, 1]), (256, 256, 1)))
    assert np.all(depth == MINVAL)
    st = o.stats()
    assert st["fragments_tested"] == st["fragments_shaded"] == 8321 and st["fragments_written"] == 0


def test_no_top_left_rule_shared_edge_is_covered_twice(oracle_lib):
    """Two triangles sharing the diagonal of the square [64,192]^2: `all >= 0 || all <= 0` (Rasterizer.cs:493-494)
    accepts zero edge values on both sides, so the 129 diagonal pixels belong to both triangles."""
    P = {"a": (-0.5, 0.5), "b": (0.5, 0.5), "c": (0.5, -0.5), "d": (-0.5, -0.5)}     # screen (64,64) (192,64) (192,192) (64,192)
    pos = [P["a"] + (0.0,), P["b"] + (0.0,), P["c"] + (0.0,), P["d"] + (0.0,)]
    v = scenes.make_vertices(pos, color=[(0.25, 0.25, 0.25, 0.25)] * 4)
    I = hm.identity()
    d = scenes.Draw(v, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16), I, I, I, program=Program.FlatColor,
                    cull=CullMode.None_, depth_test=DepthTest.Disabled, blend=BlendMode.Additive)
    s = scenes.Scene("shared_edge", 256, 256, [d], clear_color=(0.0, 0.0, 0.0, 0.0))
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(s)
    st = o.stats()
    assert st["fragments_tested"] == 129 * 129 + 129
    y, x = np.mgrid[0:256, 0:256]
    inside = (x >= 64) & (x <= 192) & (y >= 64) & (y <= 192)
    diag = inside & (x == y)
    assert np.all(color[diag] == np.float32(0.5)) and np.all(color[inside & ~diag] == np.float32(0.25))
    assert np.all(color[~inside] == 0)


def test_cull_mode_front_is_negative_screen_area(oracle_lib):
    """Rasterizer.cs:411-417: `front = area < 0` with area = EdgeFunction(s0, s1, s2) over outputs {v2, v1, v0} in
    y-down screen space.  cfg1 is counter-clockwise in NDC: outputs reversed + y flipped -> area = -2^14 < 0 -> front."""
    for cull, drawn in ((CullMode.Back, True), (CullMode.Front, False), (CullMode.None_, True)):
        s = scenes.cfg1()
        s.draws[0].cull = cull
        o = ob.OracleRenderer(256, 256)
        o.render_scene(s)
        assert o.stats()["fragments_written"] == (8321 if drawn else 0), cull
    s = scenes.cfg1()                                   # reversed winding: back-facing
    s.draws[0].indices = np.array([0, 2, 1], dtype=np.uint16)
    s.draws[0].cull = CullMode.Back
    o = ob.OracleRenderer(256, 256)
    o.render_scene(s)
    assert o.stats()["fragments_written"] == 0


def near_clip_kat_scene():
    """One vertex behind the camera plane (W = -1), NearClip = 0, projection x'=x, y'=y, z'=z, w'=z+0.5.
    ClipTriangleAgainstNearPlane (Rasterizer.cs:95-160) keeps v0, v1 (z' = 0.5 >= 0) and cuts the two edges at
    t = 0.25 and t = 0.75 (exact), giving the polygon v0, v1, I1 = lerp(v1, v2, 1/4), I2 = lerp(v2, v0, 3/4) with
    screen positions (64,192) (192,192) (192,64) (0,64); the fan is (v0,v1,I1), (v0,I1,I2).  (The first draft of this
    test expected the flat colour everywhere; the oracle was right and the expectation wrong -- see the test body.)"""
    P = np.eye(4, dtype=np.float32)
    P[2, 3] = 1.0          # M34: w' += z
    P[3, 3] = 0.5          # M44
    pos = [(-0.5, -0.5, 0.5), (0.5, -0.5, 0.5), (-0.5, 2.5, -1.5)]
    v = scenes.make_vertices(pos, color=[(0.5, 0.25, 1.0, 1.0)] * 3)
    I = hm.identity()
    d = scenes.Draw(v, np.array([0, 1, 2], dtype=np.uint16), I, I, P, program=Program.FlatColor,
                    cull=CullMode.None_, depth_test=DepthTest.Disabled, blend=BlendMode.Alpha)
    return scenes.Scene("near_clip_kat", 256, 256, [d], clear_color=(0.0, 0.0, 0.0, 1.0), near_clip=0.0)


def test_near_plane_clip_polygon_and_fan(oracle_lib):
    s = near_clip_kat_scene()
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(s)
    st = o.stats()
    assert st["triangles_in"] == 1 and st["triangles_clipped"] == 1 and st["triangles_setup"] == 2
    y, x = np.mgrid[0:256, 0:256]
    t1 = (y >= 64) & (y <= 192) & (x <= 192) & (x + y >= 256)                 # (64,192) (192,192) (192,64)
    t2 = (y >= 64) & (y <= 192) & (2 * x >= y - 64) & (x + y <= 256)          # (64,192) (192,64) (0,64)
    assert t1.sum() == 8385 and t2.sum() == 12481 and (t1 & t2).sum() == 129  # the fan's inner edge is covered twice
    assert st["fragments_tested"] == st["fragments_written"] == 8385 + 12481
    covered = t1 | t2
    assert np.array_equal(color[~covered], np.tile(np.float32([0, 0, 0, 1]), ((~covered).sum(), 1)))
    # The cut vertices come from Shaders.Lerp(current, next, t, true) (Rasterizer.cs:142): Interpolate = TRUE even under
    # a flat program, and outputs[0] of both fan triangles is such a vertex, so every fragment goes through the
    # perspective interpolation: colour = c * fl(invSum * fl(1 / invSum)), i.e. c or a float neighbour of it ...
    c = np.float32([0.5, 0.25, 1.0, 1.0])
    got = color[covered].astype(np.float64) / c
    # ... and alpha = 1 - 2^-24 at those fragments, so the Alpha blend scales the colour once more: never exactly flat
    assert np.all(np.abs(got - 1.0) <= 2.0 ** -22)
    assert (got != 1.0).any()


def wireframe_kat_scene():
    """Right triangle with screen vertices (64,64) (192,64) (64,192) for DebugMode.Wireframe."""
    pos = [(-0.5, 0.5, 0.0), (0.5, 0.5, 0.0), (-0.5, -0.5, 0.0)]
    v = scenes.make_vertices(pos, color=[(0.5, 0.25, 1.0, 1.0)] * 3)
    I = hm.identity()
    d = scenes.Draw(v, np.array([0, 1, 2], dtype=np.uint16), I, I, I, program=Program.FlatColor,
                    cull=CullMode.None_, depth_test=DepthTest.Disabled, blend=BlendMode.Alpha)
    return scenes.Scene("wireframe_kat", 256, 256, [d], clear_color=(0.0, 0.0, 0.0, 1.0))


def test_wireframe_drawline_by_hand(oracle_lib):
    """DrawLine (Rasterizer.cs:232-340): bbox from TRUNCATED endpoints, a pixel is drawn when its centre (x+.5, y+.5)
    is within 0.5 of the segment.  Horizontal edge y = 64: the bbox holds row 64 only (centre 64.5, distance 0.5: in),
    x = 64..191 (at x = 192 the clamped closest point is the end point, distance^2 = 0.5: out): 128 pixels; the
    vertical edge likewise; the diagonal x + y = 256: centres with x + y + 1 = 256, i.e. x + y = 255 (distance^2 = 0.125;
    the neighbours are at 0.5: out): 128 pixels.  The three corner pixels (64,64), (191,64), (64,191) lie on two edges."""
    s = wireframe_kat_scene()
    o = ob.OracleRenderer(256, 256)
    color, _ = o.render_scene(s, debug_mode=1)
    st = o.stats()
    assert st["fragments_tested"] == st["fragments_written"] == 3 * 128
    y, x = np.mgrid[0:256, 0:256]
    rng = (x >= 64) & (x <= 191) & (y >= 64) & (y <= 191)
    expect = rng & ((y == 64) | (x == 64) | (x + y == 255))
    assert expect.sum() == 3 * 128 - 3
    assert np.array_equal(color[expect], np.tile(np.float32([0.5, 0.25, 1.0, 1.0]), (expect.sum(), 1)))
    assert np.array_equal(color[~expect], np.tile(np.float32([0, 0, 0, 1]), ((~expect).sum(), 1)))


def test_debug_varyings_program_returns_the_interpolated_normal_screencoords_barycentric(oracle_lib):
    """SWR_PROG_DEBUG_VARYINGS by hand: Interpolate at the centroid of a triangle whose three clip.w are 1 (weights -1/3 each after the
    area division, i.e. (rA, rB, rC) = (-1/3, -1/3, -1/3), w = -1, Barycentric = (1/3, 1/3, 1/3)): Normal and ScreenCoords come out
    as the plain averages, and the program returns (Screen.x + Normal.x, Screen.y + Normal.y, Bary.x + Normal.z, Bary.y + 0.5)."""
    import ctypes as C
    from oracle.binding import OVertexOutput
    lib = oracle_lib
    verts = (OVertexOutput * 3)()
    normals = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    screens = [(0.0, 0.0), (0.75, 0.0), (0.0, 0.75)]
    for i in range(3):
        verts[i].clip[:] = (0.0, 0.0, 0.5, 1.0)
        verts[i].normal[:] = normals[i]
        verts[i].screen[:] = screens[i]
        verts[i].interpolate = 1
        verts[i].has_data = 0
    out = OVertexOutput()
    third = np.float32(-1.0) / np.float32(3.0)
    lib.oswr_interpolate.restype = None
    lib.oswr_interpolate(C.byref(verts[0]), C.byref(verts[1]), C.byref(verts[2]), C.c_float(third), C.c_float(third), C.c_float(third),
                         1, C.byref(out))
    bary = [float(x) for x in out.barycentric]
    assert all(abs(b - 1.0 / 3.0) < 1e-6 for b in bary)
    assert abs(out.screen[0] - 0.25) < 1e-6 and abs(out.screen[1] - 0.25) < 1e-6
    assert all(abs(float(out.normal[i]) - 1.0 / 3.0) < 1e-6 for i in range(3))
    rgba = (C.c_float * 4)()
    lib.oswr_fragment_shader.restype = C.c_int
    assert lib.oswr_fragment_shader(4, None, C.byref(out), None, 0, 0, rgba) == 1
    want = (np.float32(out.screen[0]) + np.float32(out.normal[0]), np.float32(out.screen[1]) + np.float32(out.normal[1]),
            np.float32(out.barycentric[0]) + np.float32(out.normal[2]), np.float32(out.barycentric[1]) + np.float32(0.5))
    assert tuple(np.float32(x) for x in rgba) == want
