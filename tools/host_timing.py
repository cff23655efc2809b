import sys, time, os
if os.environ.get('WITH_TORCH'): import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softwarerenderer_amd import Device, scenes
scene = scenes.cfg3()
dev = Device(0); r = scenes.SceneRenderer(dev, scene)
for _ in range(3): r.submit_frame(); dev.flush()
dev.sync()
if os.environ.get('WITH_PROF'): dev.profile_reset(); dev.profile_enable(True)
ts = []
t0 = time.perf_counter()
for i in range(20):
    a = time.perf_counter(); r.submit_frame(); b = time.perf_counter(); dev.flush(); c = time.perf_counter()
    ts.append((b - a, c - b))
dev.sync(); t1 = time.perf_counter()
print("total ms/step", (t1 - t0) / 20 * 1e3)
print("submit ms", [round(x[0] * 1e3, 3) for x in ts[:8]])
print("flush ms", [round(x[1] * 1e3, 3) for x in ts[:8]])
print(dev.stats())
