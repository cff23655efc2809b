"""libswr_hip_test.so against the product library, bit for bit.

The test build (csrc/Makefile) differs from the product in exactly the two places that the product leaves to verification:
  * SWR_WAVE_LDS_FENCE: a real wavefront-scope release / wave_barrier / acquire at every lane-to-lane LDS hand-off of k_raster_c
    (staging -> fragment stream, the election bitmap's clear -> its atomics), where the product relies on in-order LDS execution
    plus may-alias program order (every fence form costs 25 %, swr_raster_c.hip.h);
  * SWR_NO_SIMPLE_SELECT: every pair takes the general k-th-set-bit search instead of the run select.
Both builds must therefore produce the SAME frame words and counters on everything -- and both must equal the oracle.  A compiler
that reordered an unfenced hand-off would show up here as a difference between the two libraries."""
import os

import numpy as np
import pytest

from softwarerenderer_amd import Device, _native, scenes
from softwarerenderer_amd.rasterizer import BlendMode, DebugMode, DepthTest, Program, Rasterizer
from util import assert_frame_parity, render_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def testlib_device():
    lib = "libswr_hip_test.so"
    if not os.path.exists(os.path.join(os.path.dirname(_native.LIB_PATH), lib)):
        pytest.fail(f"{lib} is missing: __graft_entry__.build() makes it (make -C softwarerenderer_amd/csrc variants)")
    dev = Device(0, lib=lib)
    yield dev
    dev.close()


SCENES = {
    "cfg3_small": lambda: scenes.cfg3(768, 512, (3, 3), (40, 24), tex_size=256),
    "cfg3_dense": lambda: scenes.cfg3(256, 256, (4, 4), (48, 32), tex_size=128, seed=21),        # many pairs per tile: several batches
    "cfg2": lambda: scenes.cfg2(640, 360, 3000),
    "phong": lambda: scenes.cfg4(width=512, height=384, grid=(2, 2), quads=(40, 24), tex_size=128),
    "nearclip": lambda: scenes.near_clip_scene(),
    "blend_none_earlyout": lambda: scenes.state_scene(blend=BlendMode.None_, seed=31),
    "stacked_translucent": lambda: scenes.stacked_scene(layers=600),
    "additive_always": lambda: scenes.state_scene(blend=BlendMode.Additive, depth_test=DepthTest.Always, seed=33),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_fenced_general_build_equals_the_product_build(device, testlib_device, name):
    scene = SCENES[name]()
    frames = []
    for dev in (device, testlib_device):
        dev.reset_stats()
        r = scenes.SceneRenderer(dev, scene)
        c, d = r.render()
        frames.append((c, d, dev.stats()))
        r.close()
    (c0, d0, s0), (c1, d1, s1) = frames
    assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), "depth words differ between the unfenced product and the fenced test build"
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32)), "colour words differ between the unfenced product and the fenced test build"
    for k in ("fragments_tested", "fragments_shaded", "fragments_written", "tile_pairs"):
        assert s0[k] == s1[k], k
    rc, rd, _ = render_oracle(scene)
    assert_frame_parity(c1, d1, rc, rd, 1, "test build / " + name)


def test_fenced_build_equals_the_product_build_in_wireframe(device, testlib_device):
    scene = scenes.cfg3(256, 256, (2, 2), (12, 8), tex_size=64, seed=64)
    Rasterizer.RenderDebugMode = DebugMode.Wireframe
    try:
        out = []
        for dev in (device, testlib_device):
            r = scenes.SceneRenderer(dev, scene)
            out.append(r.render())
            r.close()
    finally:
        Rasterizer.RenderDebugMode = DebugMode.None_
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
