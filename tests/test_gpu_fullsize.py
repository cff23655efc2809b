"""BASELINE.json's configurations at their STATED sizes on the HIP path, against the serial oracle.

tests/golden/fullsize_golden.json holds what oracle/swr_oracle.c (one thread, the schedule of record) produced for
cfg2 (1920x1080, 10k Gouraud), cfg3 (4096^2, 1M textured), cfg4 (4096^2, 1M, 4-light Phong) and cfg5 (8192^2, 1M):
sha256 of colour and depth, per-band hashes, the six counters and 64 sampled pixels (tests/golden/make_fullsize_golden.py).
A matching colour hash means 0 ULP; if it ever differs the test falls back to running the oracle on the box and applies
the north-star bar itself (depth words bit-exact, colour <= 1 ULP per channel).  cfg5 is additionally rendered as the
eight tile-row bands of the 8-GPU partition (multigpu.band_partition(8192, 8)) on one GPU and the union compared.
Size-independent properties (determinism, flush boundaries, bands) ride on the same frames."""
import hashlib
import json
import os

import numpy as np
import pytest

from softwarerenderer_amd import MainWindow, multigpu, scenes
from softwarerenderer_amd.rasterizer import Rasterizer
from util import assert_frame_parity, render_oracle

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_golden.json")))
MAKERS = {"cfg2": scenes.cfg2, "cfg3": scenes.cfg3, "cfg4": scenes.cfg4, "cfg5": scenes.cfg5,
          "cfg3_bilinear": lambda: scenes.cfg3(bilinear=True)}      # build-defined filter: parity against the build's own oracle only
COUNTERS = ("triangles_in", "triangles_setup", "triangles_clipped", "fragments_tested", "fragments_shaded", "fragments_written")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest(c, d):
    return sha(c), sha(d)


def check_against_golden(name, scene, c, d, st=None):
    g = GOLDEN[name]
    assert (g["scene"], g["width"], g["height"], g["triangles"]) == (scene.name, scene.width, scene.height, scene.n_triangles)
    if st is not None:
        assert {k: st[k] for k in COUNTERS} == g["stats"], f"{name}: counters differ from the serial oracle's"
    for s in g["samples"]:                                      # cheap localisation before the hashes
        assert int(d[s["y"], s["x"]].view(np.uint32)) == s["depth_bits"], f"{name}: depth at ({s['x']},{s['y']})"
    assert sha(d) == g["depth_sha256"], f"{name}: depth words differ from the oracle (bands: " \
        f"{[i for i, h in enumerate(g['band_depth_sha256']) if sha(d[i * g['band_rows']:(i + 1) * g['band_rows']]) != h]})"
    if sha(c) != g["color_sha256"]:
        # not bit-identical: apply the stated bar (<= 1 ULP per channel) against a fresh oracle frame
        rc, rd, _ = render_oracle(scene)
        assert_frame_parity(c, d, rc, rd, color_ulp=1, what=name)


@pytest.fixture(scope="module")
def full(device):
    scene = scenes.cfg3()
    r = scenes.SceneRenderer(device, scene)
    device.reset_stats()
    c, d = r.render()
    st = device.stats()
    yield scene, r, (c, d), digest(c, d), st
    r.close()


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5", "cfg3_bilinear"])
def test_fullsize_golden(device, full, name):
    if name == "cfg3":
        scene, _, (c, d), _, st = full
    else:
        scene = MAKERS[name]()
        r = scenes.SceneRenderer(device, scene)
        device.reset_stats()
        c, d = r.render()
        st = device.stats()
        r.close()
    check_against_golden(name, scene, c, d, st)


def test_cfg5_eight_bands_union(device):
    """cfg5 as the 8-GPU run shards it: eight bands of 64 tile rows, each rendered by its own window on this one GPU
    (geometry replicated, band-aware mesh rejection on), concatenated in rank order = the gathered frame."""
    scene = scenes.cfg5()
    g = GOLDEN["cfg5"]
    bands = multigpu.band_partition(scene.height, 8)
    assert [b[1] for b in bands] == [64] * 8 and g["band_rows"] == 1024
    parts_c, parts_d = [], []
    for i, band in enumerate(bands):
        win = MainWindow(device, scene.width, scene.height)
        win.SetBand(*band)
        rb = scenes.SceneRenderer(device, scene, window=win)
        c, d = rb.render()
        rb.close()
        assert c.shape == (1024, scene.width, 4)
        assert sha(d) == g["band_depth_sha256"][i], f"band {i}: depth words differ from the oracle's rows"
        parts_c.append(c); parts_d.append(d)
    MainWindow(device, scene.width, scene.height).SetBand(-1, -1)
    check_against_golden("cfg5", scene, np.concatenate(parts_c), np.concatenate(parts_d))


def test_full_frame_counters_and_determinism(device, full):
    scene, r, _, ref, st = full
    assert st["triangles_in"] == 1_000_000 and st["triangles_clipped"] == 0
    assert st["fragments_tested"] >= st["fragments_shaded"] >= st["fragments_written"] > 25_000_000
    assert {k: st[k] for k in COUNTERS} == GOLDEN["cfg3"]["stats"]
    for _ in range(2):
        assert digest(*r.render()) == ref


def test_full_frame_is_independent_of_flush_boundaries(device, full):
    scene, r, _, ref, _ = full
    w = r.window
    w.ClearDepthBuffer(); w.ClearColorBuffer(scene.clear_color)
    for i, (d, prog, mesh) in enumerate(zip(scene.draws, r.programs, r.meshes)):
        Rasterizer.RenderMesh(w, mesh, None, d.model, d.view, d.projection, prog.VertexShader, prog.FragmentShader,
                              d.cull, d.depth_test, d.blend)
        if i % 5 == 4:
            device.flush()
    assert digest(*w._read()) == ref


def test_full_frame_two_bands_equal_single_gpu_frame(device, full):
    scene, r, _, ref, _ = full
    parts_c, parts_d = [], []
    for band in multigpu.band_partition(scene.height, 2):
        win = MainWindow(device, scene.width, scene.height)
        win.SetBand(*band)
        rb = scenes.SceneRenderer(device, scene, window=win)
        c, d = rb.render()
        rb.close()
        parts_c.append(c); parts_d.append(d)
    MainWindow(device, scene.width, scene.height).SetBand(-1, -1)
    assert digest(np.concatenate(parts_c), np.concatenate(parts_d)) == ref
