"""Host-side logic: enum ordinals, scene generators, matrix factories, band partition, bench.py's rank launcher."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from softwarerenderer_amd import hostmath as hm, multigpu, scenes
from softwarerenderer_amd.rasterizer import (BlendMode, CullMode, DebugMode, DepthTest, Rasterizer, VERTEX_DTYPE,
                                              as_vertex_array, default_uniforms)


def test_enum_ordinals_match_the_reference():
    assert [int(x) for x in DebugMode] == [0, 1]                       # Rasterizer.cs:14-18
    assert [x.name for x in BlendMode] == ["None_", "Alpha", "Additive", "Multiply"]      # :25-31
    assert [x.name for x in DepthTest] == ["Disabled", "Less", "LessEqual", "Greater", "GreaterEqual", "Equal", "NotEqual", "Always"]
    assert [x.name for x in CullMode] == ["None_", "Back", "Front"]    # :45-50
    assert Rasterizer.NearClip == 0.1 and Rasterizer.FarClip == 1000.0  # :20-21


def test_default_uniforms_are_the_reference_defaults():
    u = default_uniforms()                                             # Renderer.cs:39-44
    assert (u.fog_start, u.fog_end) == (1.0, 25.0)
    assert np.allclose(list(u.fog_color), [1.0, 0.62, 0.5, 1.0])
    assert np.allclose(list(u.light_direction), [0.5, -np.sqrt(0.5), -0.5], atol=1e-6)   # EulerToDirection(-45,-45,0)


def test_vertex_array_views():
    v = np.zeros((5, 12), dtype=np.float32)
    assert as_vertex_array(v).dtype == VERTEX_DTYPE and as_vertex_array(v).shape == (5,)
    with pytest.raises(ValueError):
        as_vertex_array(np.zeros((5, 11), dtype=np.float32))


def test_scenes_are_deterministic_and_fit_u16():
    a, b = scenes.cfg2(320, 200, 500), scenes.cfg2(320, 200, 500)
    assert a.draws[0].vertices.tobytes() == b.draws[0].vertices.tobytes()
    s = scenes.cfg3(256, 256, (2, 2), (16, 8), tex_size=32)
    assert s.n_triangles == 2 * 2 * 16 * 8 * 2
    for d in s.draws:
        assert d.vertices.shape[0] <= 65535 and d.indices.dtype == np.uint16 and int(d.indices.max()) < d.vertices.shape[0]


def test_full_size_cfg3_shape():
    s = scenes.cfg3()
    assert (s.width, s.height, s.n_triangles, len(s.draws)) == (4096, 4096, 1_000_000, 16)
    assert s.textures[0].shape == (2048, 2048, 4) and int(s.textures[0][..., 3].min()) == 255


def test_perspective_matches_system_numerics_layout():
    p = hm.create_perspective_fov(np.pi / 2, 2.0, 0.1, 1000.0)
    assert p[2, 3] == -1.0 and p[3, 3] == 0.0
    assert np.isclose(p[0, 0], 0.5) and np.isclose(p[1, 1], 1.0)
    assert np.isclose(p[2, 2], 1000.0 / (0.1 - 1000.0)) and np.isclose(p[3, 2], 0.1 * 1000.0 / (0.1 - 1000.0))
    v = hm.create_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0))
    assert np.allclose(v, np.eye(4))


@pytest.mark.parametrize("height,world", [(4096, 8), (8192, 8), (1080, 4), (100, 8), (16, 4), (1, 2), (333, 3)])
def test_band_partition_covers_every_tile_row_once(height, world):
    bands = multigpu.band_partition(height, world)
    assert len(bands) == world
    rows = sum(n for _, n in bands)
    assert rows == multigpu.tile_rows(height)
    nxt = 0
    for first, n in bands:
        assert first == nxt
        nxt += n
    assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    assert sum(multigpu.band_pixel_rows(height, b)[1] for b in bands) == height


# ---- bench.py started directly with --gpus N launches its own ranks (the driver calls `python bench.py --gpus N ...`) ----
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("swr_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launches_its_own_ranks_when_started_directly(monkeypatch):
    import argparse
    import types
    bench = _bench_module()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    def fake_run(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"], seen["cwd"] = cmd, env, cwd
        return types.SimpleNamespace(returncode=7)

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    rc = bench.launch_ranks_if_needed(argparse.Namespace(gpus=4, fake_world=0), argv, run=fake_run)
    assert rc == 7                                                   # the children's exit code is the launcher's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                       # same arguments, after the script
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["cwd"] == ROOT


def test_bench_does_not_launch_when_it_is_a_rank_or_single_gpu(monkeypatch):
    import argparse
    bench = _bench_module()

    def boom(*a, **k):
        raise AssertionError("must not launch")

    for k in ("WORLD_SIZE", "RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.launch_ranks_if_needed(argparse.Namespace(gpus=1, fake_world=0), [], run=boom) is None
    assert bench.launch_ranks_if_needed(argparse.Namespace(gpus=2, fake_world=2), [], run=boom) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.launch_ranks_if_needed(argparse.Namespace(gpus=2, fake_world=0), [], run=boom) is None


def test_multi_gpu_model_predicts_from_the_committed_single_gpu_stage_times():
    """the N > 1 JSON carries what DESIGN.md section 6 predicts for that N (vertex + setup replicated, the rest divided by
    N, one 64 GB/s xGMI link per band into rank 0), so that a SCALE run can be read against it line by line."""
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    m1 = bench.multi_gpu_model("cfg3", 1, 4096, 4096, 3)
    assert m1["gather_ms"] == 0.0
    prev = None
    for n in (2, 4, 8):
        m = bench.multi_gpu_model("cfg3", n, 4096, 4096, 3)
        assert abs(m["gather_ms"] - 4096 * 4096 * 12 / n / 64e9 * 1e3) < 1e-3          # 201 MB / N over one link
        assert m["source"] and m["source"].startswith("r") and m["render_ms"] > 0       # a committed profiles/r*_final_bench_with_cpu.json of cfg3
        assert m["step_ms"] == max(m["render_ms"], m["gather_ms"])
        assert m["front_ms"] > m["raster_ms"] * 0 and (prev is None or m["render_ms"] < prev)   # rendering scales, the replicated front end does not
        prev = m["render_ms"]
    assert bench.multi_gpu_model("cfg2", 2, 1080, 1920, 3)["render_ms"] is None          # no committed single-GPU line of that config: gather only
