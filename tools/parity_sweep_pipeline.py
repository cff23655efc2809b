#!/usr/bin/env python3
"""Extended parity sweep, part 3 (test infrastructure, not part of the suite): FRAMES IN FLIGHT.  Random sequences of 3-6 order-dependent
frames (the first clears, the others accumulate; random generator families, states and sizes) are flushed back to back without any
synchronisation under a random pipelining mode (0 / 1 / 2, changed between sequences on one context), then read once and compared with
the oracle's result of the same sequence: depth bit-exact, colour <= 1 ULP, identical counters.  Every hand-over of the pipelined
path is on the line here: the two RasterSets, front_done / raster_done, the per-batch poison protocol (growing pair buffers replay
in the middle of a sequence), the event-ordered fallback to one stream for big batches.
usage: parity_sweep_pipeline.py [SECONDS=240] [RNG_SEED=5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                       # noqa: E402
from util import assert_frame_parity                                    # noqa: E402
from softwarerenderer_amd import Device, scenes                         # noqa: E402
from softwarerenderer_amd.rasterizer import BlendMode, CullMode, DepthTest, Program      # noqa: E402
from oracle.binding import OracleRenderer                               # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = Device(0)
fresh_every = 25                      # a new context now and then: its pair buffers start empty, so sequences replay again


def make(kind, W, H, seed, n):
    if kind == 0:
        return scenes.cfg2(W, H, int(rng.integers(100, 4000)), seed=seed, min_area=float(rng.uniform(1, 50)), max_area=float(rng.uniform(100, 40000)))
    if kind == 1:
        return scenes.cfg3(W, H, (int(rng.integers(1, 4)), int(rng.integers(1, 4))), (int(rng.integers(4, 50)), int(rng.integers(4, 30))),
                           tex_size=int(rng.integers(8, 200)), seed=seed, program=[Program.Dust2LambertFog, Program.Phong4Point, Program.Gouraud][n % 3])
    if kind == 2:
        return scenes.near_clip_scene(W, H, int(rng.integers(50, 800)), seed=seed)
    return scenes.state_scene(W, H, int(rng.integers(100, 2500)), seed=seed, cull=list(CullMode)[n % 3], depth_test=list(DepthTest)[n % 8],
                              blend=list(BlendMode)[n % 4])


bad = n = frames = 0
t0 = time.time()
while time.time() - t0 < SECONDS:
    if n and n % fresh_every == 0:
        dev.close(); dev = Device(0)
    W, H = int(rng.integers(100, 800)), int(rng.integers(100, 600))
    mode = int(rng.integers(0, 3))
    dev.set_pipelining(mode)
    seq = []
    for k in range(int(rng.integers(3, 7))):
        s = make(int(rng.integers(0, 4)), W, H, int(rng.integers(1, 1 << 30)), n + k)
        if k:
            s.clear_color = None; s.clear_depth = False
        seq.append(s)
    o = OracleRenderer(W, H)
    for s in seq:
        rc, rd = o.render_scene(s)
    ost = o.stats(); o.close()
    rs, win = [], None
    for s in seq:
        rs.append(scenes.SceneRenderer(dev, s, window=win)); win = rs[-1].window
    dev.reset_stats()
    for r in rs:
        r.submit_frame(); dev.flush()
    c, d = win._read()
    st = dev.stats()
    try:
        assert_frame_parity(c, d, rc, rd, 1, f"sequence {n} mode {mode}")
        for key in ("triangles_in", "triangles_setup", "fragments_tested", "fragments_shaded", "fragments_written"):
            assert st[key] == ost[key], (key, st[key], ost[key])
    except AssertionError as e:
        bad += 1; print("MISMATCH", n, mode, W, H, [s.name for s in seq], str(e)[:300], flush=True)
    for r in rs:
        r.close()
    n += 1; frames += len(seq)
    if n % 25 == 0:
        print("progress", n, frames, round(time.time() - t0), flush=True)
print("sequences", n, "frames", frames, "mismatches", bad)
