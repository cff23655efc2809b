#!/usr/bin/env python3
"""Wall-clock frame time of one library build, event-free: frames.py <lib path> <config> <pipelining 0|1|2> [frames]  (same-box A/B of builds
kept in build_ab/; one line per call)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from softwarerenderer_amd import _native      # noqa: E402
_native.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from softwarerenderer_amd import Device, scenes      # noqa: E402

scene = scenes.cfg3(bilinear=True) if sys.argv[2] == "cfg3_bilinear" else getattr(scenes, sys.argv[2])()
dev = Device(0)
dev.set_pipelining(int(sys.argv[3]))
r = scenes.SceneRenderer(dev, scene)
for _ in range(40):
    r.submit_frame(); dev.flush()
dev.sync()
N = int(sys.argv[4]) if len(sys.argv) > 4 else 100
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        r.submit_frame(); dev.flush()
    dev.sync()
    best = min(best, 1e3 * (time.perf_counter() - t0) / N)
print(f"{sys.argv[1]:34s} {sys.argv[2]} pipelining {sys.argv[3]}  ms/frame {best:.4f}", flush=True)
