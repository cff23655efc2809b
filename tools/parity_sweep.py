#!/usr/bin/env python3
"""Extended parity sweep (test infrastructure, not part of the suite): random scenes of the four generator families at random
sizes / states, HIP path against the oracle (depth bit-exact, colour <= 1 ULP, identical counters), for SECONDS of wall time.
usage: parity_sweep.py [SECONDS=420] [RNG_SEED=7]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from softwarerenderer_amd import Device, scenes
from softwarerenderer_amd.rasterizer import DepthTest, Program, BlendMode, CullMode
dev = Device(0)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = n = 0; t0 = time.time()
while time.time() - t0 < SECONDS:
    kind = n % 4
    seed = int(rng.integers(1, 1 << 30))
    W, H = int(rng.integers(100, 900)), int(rng.integers(100, 700))
    if kind == 0:
        s = scenes.cfg2(W, H, int(rng.integers(200, 6000)), seed=seed, min_area=float(rng.uniform(1, 50)), max_area=float(rng.uniform(100, 60000)))
    elif kind == 1:
        s = scenes.cfg3(W, H, (int(rng.integers(1, 5)), int(rng.integers(1, 5))), (int(rng.integers(4, 60)), int(rng.integers(4, 40))), tex_size=int(rng.integers(8, 300)), seed=seed,
                        program=[Program.Dust2LambertFog, Program.Phong4Point, Program.Gouraud][n % 3])
    elif kind == 2:
        s = scenes.near_clip_scene(W, H, int(rng.integers(50, 1500)), seed=seed)
    else:
        s = scenes.state_scene(W, H, int(rng.integers(100, 3000)), seed=seed, cull=list(CullMode)[n % 3], depth_test=list(DepthTest)[n % 8], blend=list(BlendMode)[n % 4])
    try:
        T.run_both(dev, s)
    except AssertionError as e:
        bad += 1; print("MISMATCH", kind, seed, W, H, str(e)[:300], flush=True)
    n += 1
    if n % 50 == 0: print("progress", n, round(time.time() - t0), flush=True)
print("scenes", n, "mismatches", bad)
