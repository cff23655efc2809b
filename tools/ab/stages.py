#!/usr/bin/env python3
"""Per-stage hipEvent times of one library build on one stream (pipelining off): stages.py <lib path> <config> [frames]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from softwarerenderer_amd import _native      # noqa: E402
_native.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from softwarerenderer_amd import Device, scenes      # noqa: E402

scene = scenes.cfg3(bilinear=True) if sys.argv[2] == "cfg3_bilinear" else getattr(scenes, sys.argv[2])()
dev = Device(0)
try:
    dev.set_pipelining(0)
except Exception:
    pass
r = scenes.SceneRenderer(dev, scene)
for _ in range(20):
    r.submit_frame(); dev.flush()
dev.sync(); dev.profile_reset(); dev.profile_enable(1)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for _ in range(N):
    r.submit_frame(); dev.flush()
p = dev.profile()
print(f"{sys.argv[1]:34s} {sys.argv[2]}", json.dumps({k[:-3]: round(1e3 * v / N, 1) for k, v in p.items() if k.endswith("_ms")}), flush=True)
