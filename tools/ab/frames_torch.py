#!/usr/bin/env python3
"""tools/ab/frames.py with torch imported first (the library then binds to the HIP runtime torch bundles, as in bench.py):
frames_torch.py <lib> <config> <pipelining> [frames]"""
import os, sys, time
import torch
torch.cuda.is_available()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from softwarerenderer_amd import _native
_native.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from softwarerenderer_amd import Device, scenes
scene = getattr(scenes, sys.argv[2])()
dev = Device(0); dev.set_pipelining(int(sys.argv[3]))
r = scenes.SceneRenderer(dev, scene)
for _ in range(40):
    r.submit_frame(); dev.flush()
dev.sync()
N = int(sys.argv[4]) if len(sys.argv) > 4 else 100
best = 1e9; allr = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        r.submit_frame(); dev.flush()
    dev.sync()
    allr.append(1e3 * (time.perf_counter() - t0) / N)
import subprocess
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
print(f"torch+ {sys.argv[2]} pipelining {sys.argv[3]} frames {N}: ms/frame {min(allr):.4f} (reps {[round(x,4) for x in allr]})  hip runtime: {libs}", flush=True)
