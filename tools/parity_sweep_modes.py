#!/usr/bin/env python3
"""Extended parity sweep, part 2 (test infrastructure, not part of the suite): DebugMode.Wireframe scenes and frames rendered
as 2-6 tile-row bands (with the band-aware mesh rejection active), HIP path against the oracle, SECONDS of wall time each.
usage: parity_sweep_modes.py [SECONDS=180]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
import numpy as np
import test_gpu_parity as T
from util import assert_frame_parity, render_oracle
from softwarerenderer_amd import Device, scenes, multigpu
from softwarerenderer_amd.rasterizer import DepthTest, Program, BlendMode, CullMode, DebugMode, Rasterizer, MainWindow
from oracle.binding import OracleRenderer
dev = Device(0)
rng = np.random.default_rng(11)
def make(n):
    seed = int(rng.integers(1, 1 << 30)); W, H = int(rng.integers(60, 500)), int(rng.integers(60, 400))
    k = n % 3
    if k == 0: return scenes.cfg2(W, H, int(rng.integers(50, 1500)), seed=seed, min_area=float(rng.uniform(1, 50)), max_area=float(rng.uniform(100, 30000)))
    if k == 1: return scenes.cfg3(W, H, (int(rng.integers(1, 4)), int(rng.integers(1, 4))), (int(rng.integers(4, 30)), int(rng.integers(4, 20))), tex_size=int(rng.integers(8, 100)), seed=seed)
    return scenes.near_clip_scene(W, H, int(rng.integers(50, 600)), seed=seed)
bad = n = 0; t0 = time.time()
Rasterizer.RenderDebugMode = DebugMode.Wireframe
try:
    while time.time() - t0 < SECONDS:
        s = make(n)
        o = OracleRenderer(s.width, s.height); rc, rd = o.render_scene(s, debug_mode=1); rst = o.stats(); o.close()
        dev.reset_stats(); r = scenes.SceneRenderer(dev, s); c, d = r.render(); st = dev.stats(); r.close()
        try:
            assert_frame_parity(c, d, rc, rd, 1, "wf")
            for k in ("fragments_tested", "fragments_shaded", "fragments_written"): assert st[k] == rst[k], k
        except AssertionError as e:
            bad += 1; print("WF MISMATCH", n, s.name, str(e)[:200], flush=True)
        n += 1
finally:
    Rasterizer.RenderDebugMode = DebugMode.None_
print("wireframe scenes", n, "mismatches", bad, flush=True)
bad2 = m = 0; t1 = time.time()
while time.time() - t1 < SECONDS:
    s = make(m); world = int(rng.integers(2, 7))
    rc, rd, rst = render_oracle(s)
    cols, deps, frag = [], [], 0
    for band in multigpu.band_partition(s.height, world):
        win = MainWindow(dev, s.width, s.height); win.SetBand(*band)
        dev.reset_stats(); r = scenes.SceneRenderer(dev, s, window=win); c, d = r.render(); frag += dev.stats()["fragments_written"]; r.close()
        cols.append(c); deps.append(d)
    win = MainWindow(dev, s.width, s.height); win.SetBand(-1, -1)
    try:
        assert_frame_parity(np.concatenate(cols, 0), np.concatenate(deps, 0), rc, rd, 1, "bands")
        assert frag == rst["fragments_written"]
    except AssertionError as e:
        bad2 += 1; print("BAND MISMATCH", m, s.name, world, str(e)[:200], flush=True)
    m += 1
print("banded scenes", m, "mismatches", bad2)
