#!/usr/bin/env python3
"""Wall-clock frame time with an event pair around every raster launch (profile mode 2): evframes.py <lib> <config> <pipelining> [mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from softwarerenderer_amd import _native
_native.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from softwarerenderer_amd import Device, scenes
import numpy as np
scene = getattr(scenes, sys.argv[2])()
dev = Device(0); dev.set_pipelining(int(sys.argv[3]))
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
r = scenes.SceneRenderer(dev, scene)
for _ in range(40):
    r.submit_frame(); dev.flush()
dev.sync(); dev.profile_reset(); dev.profile_enable(mode)
N = 40
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        r.submit_frame(); dev.flush()
    dev.sync()
    best = min(best, 1e3 * (time.perf_counter() - t0) / N)
s = np.sort(dev.raster_samples()) if mode else np.zeros(1)
print(f"{sys.argv[1]:30s} {sys.argv[2]} pipelining {sys.argv[3]} events {mode}  ms/frame {best:.4f}  raster median {np.median(s):.4f}", flush=True)
