"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): depth words bit-exact; colour within 1 ULP per float channel
(tolerance written here: COLOR_ULP = 1).  Stats counters must match exactly as well.
"""
import numpy as np
import pytest

from softwarerenderer_amd import scenes
from softwarerenderer_amd.rasterizer import BlendMode, CullMode, DepthTest, Program
from util import assert_frame_parity, render_oracle

pytestmark = pytest.mark.gpu
COLOR_ULP = 1


def run_both(device, scene, retained=True):
    rc, rd, rst = render_oracle(scene)
    device.reset_stats()
    r = scenes.SceneRenderer(device, scene, retained=retained)
    c, d = r.render()
    st = device.stats()
    r.close()
    n_inexact = assert_frame_parity(c, d, rc, rd, COLOR_ULP, scene.name)
    for k in ("triangles_in", "triangles_setup", "triangles_clipped", "fragments_tested", "fragments_shaded", "fragments_written"):
        assert st[k] == rst[k], f"{scene.name}: stats[{k}] gpu={st[k]} oracle={rst[k]}"
    return n_inexact, st


def test_cfg1_single_flat_triangle(device):
    n_inexact, st = run_both(device, scenes.cfg1())
    assert n_inexact == 0
    assert st["fragments_written"] == 8321


@pytest.mark.parametrize("size", [(480, 270, 2000), (333, 217, 900)])
def test_cfg2_gouraud_depth(device, size):
    run_both(device, scenes.cfg2(*size))


def test_cfg2_array_signature(device):
    run_both(device, scenes.cfg2(256, 256, 500, seed=4), retained=False)


def test_cfg3_textured_lambert_fog_small(device):
    run_both(device, scenes.cfg3(512, 512, (4, 4), (32, 16), tex_size=256))


def test_cfg3_odd_size(device):
    run_both(device, scenes.cfg3(501, 333, (3, 2), (40, 24), tex_size=128, seed=9))


def test_cfg4_phong_small(device):
    run_both(device, scenes.cfg4(width=384, height=384, grid=(3, 3), quads=(24, 16), tex_size=128))


@pytest.mark.parametrize("program", [Program.Dust2LambertFog, Program.Gouraud, Program.FlatColor, Program.Phong4Point])
def test_near_clip(device, program):
    s = scenes.near_clip_scene(program=program)
    if program == Program.Phong4Point:
        u = s.draws[0].uniforms
        u.camera_position[:] = (0.0, 0.0, 0.0)
        for i in range(4):
            u.lights[i].position[:] = (2.0 * i - 3.0, 1.0, -2.0)
            u.lights[i].range = 20.0
            u.lights[i].color[:] = (1.0, 0.8, 0.6)
            u.lights[i].intensity = 1.0
    run_both(device, s)


@pytest.mark.parametrize("depth_test", list(DepthTest))
def test_every_depth_test(device, depth_test):
    run_both(device, scenes.state_scene(depth_test=depth_test))


@pytest.mark.parametrize("blend", list(BlendMode))
def test_every_blend_mode(device, blend):
    run_both(device, scenes.state_scene(blend=blend, seed=21))


@pytest.mark.parametrize("cull", list(CullMode))
def test_every_cull_mode(device, cull):
    run_both(device, scenes.state_scene(cull=cull, seed=31))


def test_blend_none_early_out_textured(device):
    s = scenes.state_scene(blend=BlendMode.None_, program=Program.Dust2LambertFog, seed=41, zero_alpha_fraction=0.4)
    s.textures = [scenes.random_texture(32, 5, alpha=None)]
    s.draws[0].texture = 0
    run_both(device, s)


@pytest.mark.parametrize("layers", [3, 64, 65, 200, 2500])
def test_stacked_translucent_order(device, layers):
    """Equal-depth translucent layers: the result depends on submission order (>= ties, alpha blend)."""
    run_both(device, scenes.stacked_scene(layers=layers))


def test_degenerate_inputs_are_skipped(device):
    run_both(device, scenes.degenerate_scene())


def test_no_clear_accumulates(device):
    """Two frames without clearing in between: the second starts from the first's buffers."""
    s1 = scenes.cfg2(200, 120, 300, seed=2)
    s2 = scenes.state_scene(200, 120, 300, seed=3, blend=BlendMode.Additive)
    s2.clear_color = None
    s2.clear_depth = False
    from oracle.binding import OracleRenderer
    o = OracleRenderer(200, 120)
    o.render_scene(s1)
    rc, rd = o.render_scene(s2)
    r1 = scenes.SceneRenderer(device, s1)
    r1.render()
    r2 = scenes.SceneRenderer(device, s2, window=r1.window)
    c, d = r2.render()
    assert_frame_parity(c, d, rc, rd, COLOR_ULP, "no-clear")
    r1.close(); r2.close()


@pytest.mark.parametrize("make", [
    lambda: scenes.cfg2(300, 200, 400, seed=61, min_area=50.0, max_area=6000.0),
    lambda: scenes.near_clip_scene(program=Program.Dust2LambertFog),
    lambda: scenes.state_scene(blend=BlendMode.None_, seed=62),
    lambda: scenes.state_scene(blend=BlendMode.Additive, depth_test=DepthTest.Always, seed=63),
    lambda: scenes.cfg3(256, 256, (2, 2), (12, 8), tex_size=64, seed=64),
    lambda: scenes.degenerate_scene(),
    lambda: __import__("test_oracle_kat").wireframe_kat_scene(),
], ids=["gouraud", "nearclip", "blend_none", "additive_always", "textured_patches", "degenerate", "hand_derived_kat"])
def test_wireframe_debug_mode(device, make):
    """DebugMode.Wireframe = three DrawLine calls per triangle (Rasterizer.cs:232-340,419-425), incl. its quirks:
    every edge uses depths[0..1] / outputs[0..1] of the triangle, alpha test is `!= 0`, no row early-out."""
    from softwarerenderer_amd.rasterizer import DebugMode, Rasterizer
    from oracle.binding import OracleRenderer
    scene = make()
    o = OracleRenderer(scene.width, scene.height)
    rc, rd = o.render_scene(scene, debug_mode=1)
    rst = o.stats()
    Rasterizer.RenderDebugMode = DebugMode.Wireframe
    try:
        device.reset_stats()
        r = scenes.SceneRenderer(device, scene)
        c, d = r.render()
        st = device.stats()
        r.close()
    finally:
        Rasterizer.RenderDebugMode = DebugMode.None_
    assert_frame_parity(c, d, rc, rd, COLOR_ULP, "wireframe " + scene.name)
    for k in ("triangles_in", "triangles_setup", "triangles_clipped", "fragments_tested", "fragments_shaded", "fragments_written"):
        assert st[k] == rst[k], (k, st[k], rst[k])
    assert rst["fragments_written"] > 0 or scene.name == "degenerate"
    if scene.name == "wireframe_kat":
        assert st["fragments_written"] == 3 * 128           # tests/test_oracle_kat.py::test_wireframe_drawline_by_hand


def _random_scene(seed):
    """A random small frame: random size (incl. non-multiples of 16 and tiny targets), a mix of tiny / huge / sliver /
    behind-camera / off-screen triangles, random matrices and a random state vector."""
    import softwarerenderer_amd.hostmath as hm
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(1, 220)), int(rng.integers(1, 160))
    n = int(rng.integers(1, 500))
    proj = scenes._perspective(W, H, fov_deg=float(rng.uniform(40, 110)))
    kind = rng.integers(0, 5, n)
    c = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-12.0, 1.0, n)], axis=1)
    size = np.where(kind == 0, 0.05, np.where(kind == 1, 0.5, np.where(kind == 2, 3.0, np.where(kind == 3, 12.0, 1.0))))
    pos = c[:, None, :] + rng.normal(size=(n, 3, 3)) * size[:, None, None]
    sliver = kind == 4
    pos[sliver, 2, :] = pos[sliver, 0, :] * 0.5 + pos[sliver, 1, :] * 0.5 + rng.normal(size=(int(sliver.sum()), 3)) * 1e-3
    col = np.concatenate([rng.uniform(0, 1, (3 * n, 3)), rng.choice([0.0, 0.3, 1.0], size=(3 * n, 1), p=[0.1, 0.3, 0.6])], axis=1)
    nrm = rng.normal(size=(3 * n, 3)); nrm[rng.uniform(size=3 * n) < 0.02] = 0.0          # a few zero normals -> NaN WorldNormal
    v = scenes.make_vertices(pos.reshape(-1, 3), uv=rng.uniform(-3, 4, (3 * n, 2)), normal=nrm, color=col)
    model = hm.multiply(hm.create_scale(float(rng.uniform(0.5, 1.5))), hm.create_rotation_y(float(rng.uniform(-0.5, 0.5))))
    view = hm.create_look_at((float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 2.0), (0.0, 0.0, -5.0), (0.0, 1.0, 0.0))
    program = Program(int(rng.integers(0, 3)))
    d = scenes.Draw(v, np.arange(3 * n, dtype=np.uint16), model, view, proj, program=program, uniforms=scenes.default_uniforms(),
                    texture=0 if rng.uniform() < 0.7 else None, cull=CullMode(int(rng.integers(0, 3))),
                    depth_test=DepthTest(int(rng.integers(0, 8))), blend=BlendMode(int(rng.integers(0, 4))))
    return scenes.Scene(f"random{seed}", W, H, [d], textures=[scenes.random_texture(int(rng.integers(1, 40)), seed, alpha=None)],
                        near_clip=float(rng.choice([0.1, 0.5, 0.01])))


@pytest.mark.parametrize("mixed_states", [False, True])
def test_every_draw_has_its_own_uniforms_and_texture(device, mixed_states):
    """One flush, nine overlapping draws, each with its own uniform block (light direction / colour, fog colour / range -- one
    range degenerate, one outside the division core's guard), its own texture (different sizes, one draw untextured).  The raster
    kernel holds the per-draw constants in SGPRs and reloads them when the draw changes inside a tile's stream (DrawConsts):
    with one program the state-specialised kernel runs; `mixed_states` also varies program / blend / depth test (generic kernel)."""
    from softwarerenderer_amd.rasterizer import Program, BlendMode, DepthTest
    rng = np.random.default_rng(5)
    s = scenes.cfg3(300, 220, (3, 3), (14, 10), tex_size=32, seed=9)
    s.textures = [scenes.random_texture(n, 40 + n, alpha=None if n % 2 else 255) for n in (32, 7, 64, 1, 19)]
    fog_ranges = [(1.0, 25.0), (0.5, 6.0), (3.0, 3.0), (2.0, 2.0 + 1e-20), (10.0, 4.0), (0.0, 1e15), (1.0, 25.0), (5.0, 40.0), (1.0, 2.0)]
    for i, d in enumerate(s.draws):
        u = scenes.default_uniforms()
        ld = rng.normal(size=3); ld /= np.linalg.norm(ld)
        u.light_direction[:] = [float(x) for x in ld]
        u.light_color[:] = [float(x) for x in rng.uniform(0.2, 1.5, 4)]
        u.fog_color[:] = [float(x) for x in rng.uniform(0.0, 1.0, 4)]
        u.fog_start, u.fog_end = fog_ranges[i]
        d.uniforms = u
        d.texture = None if i == 4 else i % len(s.textures)
        if mixed_states:
            d.program = [Program.Dust2LambertFog, Program.Gouraud, Program.FlatColor][i % 3]
            d.blend = list(BlendMode)[i % 4]
            d.depth_test = [DepthTest.LessEqual, DepthTest.Less, DepthTest.Always][i % 3]
    s.name = "per_draw_constants" + ("_mixed" if mixed_states else "")
    run_both(device, s)


@pytest.mark.parametrize("seed", range(60))
def test_randomised_scenes(device, seed):
    run_both(device, _random_scene(1000 + seed))


@pytest.mark.parametrize("seed,depth_test", [(71, DepthTest.LessEqual), (72, DepthTest.Always), (73, DepthTest.Disabled), (74, DepthTest.Greater)])
def test_blend_none_early_out_with_alpha_gradients(device, seed, depth_test):
    """BlendMode.None row early-out (Rasterizer.cs:520-523) with alpha varying ACROSS each triangle (vertex alphas in
    [-0.6, 1]): rows die part-way through, segments straddle the 64-fragment chunks of the stream kernel, and small
    triangles in front shift their alignment."""
    rng = np.random.default_rng(seed)
    s = scenes.cfg2(200, 136, 260, seed=seed, min_area=4.0, max_area=30000.0)
    d = s.draws[0]
    d.vertices["color"][:, 3] = rng.uniform(-0.6, 1.0, d.vertices.shape[0]).astype(np.float32)
    d.blend = BlendMode.None_
    d.depth_test = depth_test
    s.name = f"none_gradient_{seed}"
    run_both(device, s)


def test_bilinear_extension_matches_its_oracle_definition(device):
    """Opt-in bilinear filter (row N4): build-defined, no reference semantics -- GPU and oracle implement one formula."""
    s = scenes.cfg3(384, 320, (3, 3), (20, 14), tex_size=37, seed=81)
    s.bilinear = True
    s.name += "_bilinear"
    run_both(device, s)
    near = scenes.cfg3(384, 320, (3, 3), (20, 14), tex_size=37, seed=81)
    c_near, _, _ = render_oracle(near)
    c_bil, _, _ = render_oracle(s)
    assert not np.array_equal(c_near, c_bil)          # the filter really changes the image


@pytest.mark.parametrize("tex_size", [64, 36, 30])
def test_bilinear_texture_layouts_give_the_same_frame(device, tex_size):
    """The bilinear filter reads a block-linear device copy (4 x 4-texel blocks of 64 B, made by swr_texture_set_filter) when the
    texture's sides are multiples of 4 (64, 36) and the row-major original otherwise (30): the layout only moves addresses, so both
    must give the oracle's frame (wrap at the texture's edges included: UVs run over [-2, 3])."""
    s = scenes.cfg3(320, 240, (2, 2), (20, 14), tex_size=tex_size, seed=83)
    s.bilinear = True
    s.name += f"_bilinear_{tex_size}"
    run_both(device, s)


def test_loaded_gltf_model_renders_like_the_oracle(device, tmp_path):
    """Row N4 end to end: glTF -> modelloader.Model -> one RenderDust2-style frame, GPU against the oracle."""
    import json
    from softwarerenderer_amd.modelloader import Model
    Model._model_cache.clear()
    nu, nv = 24, 16
    u, v = np.meshgrid(np.linspace(0, 2 * np.pi, nu + 1), np.linspace(0.05, np.pi - 0.05, nv + 1))
    nrm = np.stack([np.sin(v) * np.cos(u), np.cos(v), np.sin(v) * np.sin(u)], axis=-1).reshape(-1, 3).astype(np.float32)
    pos = (nrm * np.float32(2.0)).astype(np.float32)
    uv = np.stack([u / (2 * np.pi) * 3, v / np.pi * 2], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            idx += [a, a + nu + 1, a + 1, a + 1, a + nu + 1, a + nu + 2]
    blob = pos.tobytes() + nrm.tobytes() + uv.tobytes() + np.asarray(idx, dtype=np.uint16).tobytes()
    (tmp_path / "s.bin").write_bytes(blob)
    n = pos.shape[0]
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1]}],
           "nodes": [{"mesh": 0, "translation": [-1.5, 0, 0], "scale": [1, 1.3, 1]},
                     {"mesh": 0, "translation": [1.5, 0.5, -1], "rotation": [0.0, 0.3826834, 0.0, 0.9238795]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3}]}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": n, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": n, "type": "VEC3"},
                         {"bufferView": 2, "componentType": 5126, "count": n, "type": "VEC2"},
                         {"bufferView": 3, "componentType": 5123, "count": len(idx), "type": "SCALAR"}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": n * 12},
                           {"buffer": 0, "byteOffset": n * 12, "byteLength": n * 12},
                           {"buffer": 0, "byteOffset": n * 24, "byteLength": n * 8},
                           {"buffer": 0, "byteOffset": n * 32, "byteLength": len(idx) * 2}],
           "buffers": [{"byteLength": len(blob), "uri": "s.bin"}]}
    (tmp_path / "s.gltf").write_text(json.dumps(doc))
    model = Model().LoadModel(str(tmp_path / "s.gltf"))
    assert len(model.Meshes) == 2 and all(m.Indices.size == len(idx) for m in model.Meshes)
    s = scenes.from_model(model, 400, 300, name="gltf_spheres")
    run_both(device, s)


@pytest.mark.parametrize("depth_test", [DepthTest.Less, DepthTest.LessEqual])
@pytest.mark.parametrize("program", [Program.Dust2LambertFog, Program.Gouraud])
def test_hierarchical_z_exact_ties_and_occlusion(device, depth_test, program):
    """The raster kernel drops pairs that provably fail the depth test (hi-Z).  Worst cases for that proof: the same
    surfaces drawn twice (every fragment ties with the stored depth: passes under LessEqual, fails under Less), fully
    occluded layers, and a near layer drawn first; with translucent vertex colours so that a wrongly dropped or kept
    fragment changes the blended colour."""
    base = scenes.cfg3(320, 256, (2, 2), (24, 12), tex_size=64, seed=91, program=program)
    draws = []
    for d in base.draws:
        d.depth_test = depth_test
        d.vertices = d.vertices.copy()
        d.vertices["color"][:, 3] = 0.6
    # near-to-far would hide everything behind the first layer; far-to-near nothing: use both orders plus exact repeats
    draws = list(base.draws) + list(reversed(base.draws)) + list(base.draws[:2]) + list(base.draws[:2])
    s = scenes.Scene(f"hiz_ties_{depth_test.name}_{program.name}", base.width, base.height, draws, textures=base.textures)
    _, st = run_both(device, s)
    assert st["fragments_tested"] > st["fragments_shaded"] > 0


def test_hand_derived_kat_scenes_on_the_gpu(device):
    """The hand-derived known answers of tests/test_oracle_kat.py hold for the HIP path too (exact constants, no oracle)."""
    from test_oracle_kat import _cfg1_dust2, near_clip_kat_scene
    _, st = run_both(device, near_clip_kat_scene())
    assert st["fragments_written"] == 8385 + 12481 and st["triangles_clipped"] == 1
    for args, px, expect in (((0.0, 1.0, 0.0), (128, 128), (0.25, 0.25, 0.5, 1.0)),
                             ((0.0, 2.0, 1.0), (128, 128), (0.625, 0.375, 0.375, 1.0)),
                             ((0.0, 0.5, 1.0), (128, 128), (1.0, 0.5, 0.25, 1.0))):
        s = _cfg1_dust2(*args)
        r = scenes.SceneRenderer(device, s)
        c, _ = r.render()
        r.close()
        assert tuple(c[px[1], px[0]]) == expect
        run_both(device, s)


@pytest.mark.parametrize("make", [
    lambda: scenes.cfg3(320, 256, (2, 2), (16, 12), tex_size=32, seed=71, program=Program.DebugVaryings),
    lambda: scenes.near_clip_scene(program=Program.DebugVaryings),                 # the clipper's vertices: Normal lerped, ScreenCoords recomputed
    lambda: scenes.state_scene(program=Program.DebugVaryings, blend=BlendMode.None_, seed=72),     # row early-out kernel
], ids=["patches", "nearclip", "blend_none"])
def test_debug_varyings_program_carries_normal_screencoords_barycentric(device, make):
    """R2 complete: the varyings of Shaders.VertexOutput that no other built-in program reads -- Normal
    (Rasterizer.cs:610-613), ScreenCoords (:390,598-601), Barycentric (:638) -- travel through the raster path and come out of the
    build-defined program SWR_PROG_DEBUG_VARYINGS exactly as the oracle's Rasterizer.Interpolate delivers them."""
    scene = make()
    run_both(device, scene)
    rc, _, _ = render_oracle(scene)
    assert np.isfinite(rc).all() and float(np.abs(rc[..., :3]).max()) > 0.0


def test_debug_varyings_in_wireframe_is_refused_not_wrong(device):
    """DrawLine interpolates the TRIANGLE's outputs[0..1] on every edge (Rasterizer.cs:421-423); the backend's line records do not
    keep those two vertices' screen positions, so the combination is refused (SWR_ERR_UNSUPPORTED) rather than rendered differently."""
    from softwarerenderer_amd import _native
    from softwarerenderer_amd.rasterizer import DebugMode, Rasterizer
    scene = scenes.cfg3(128, 128, (1, 1), (6, 4), tex_size=16, seed=73, program=Program.DebugVaryings)
    Rasterizer.RenderDebugMode = DebugMode.Wireframe
    try:
        r = scenes.SceneRenderer(device, scene)
        with pytest.raises(_native.SwrError) as e:
            r.render()
        assert e.value.code == _native.SWR_ERR_UNSUPPORTED
        r.close()
    finally:
        Rasterizer.RenderDebugMode = DebugMode.None_
    device.sync()
    run_both(device, scenes.cfg2(200, 120, 300, seed=74))          # the context is fine afterwards


def test_tile_order_history_of_another_scene_changes_nothing(device):
    """The raster kernel's dispatch order takes its per-tile weights from the PREVIOUS flush on the device (any permutation of the tiles
    is correct, swr_binning.hip.h).  Frames of unrelated scenes of one size -- the history of one is the other's -- a resize in between
    (which drops the history) and the same scene twice in a row must each equal the oracle."""
    a = scenes.cfg3(384, 320, (3, 3), (24, 16), tex_size=64, seed=21)                 # many small textured triangles
    b = scenes.cfg2(384, 320, 300, seed=22, min_area=500.0, max_area=40000.0)         # few big ones: very different tile weights
    c = scenes.cfg2(200, 136, 400, seed=23)                                           # another size: another tiling
    for scene in (a, b, a, a, c, b, b):
        run_both(device, scene)
