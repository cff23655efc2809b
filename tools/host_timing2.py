import sys, time, os
if os.environ.get('WITH_TORCH'): import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softwarerenderer_amd import Device, scenes
scene = scenes.cfg3()
dev = Device(0); r = scenes.SceneRenderer(dev, scene)
ts = []
for i in range(40):
    a = time.perf_counter(); r.submit_frame(); dev.flush(); dev.sync(); ts.append(time.perf_counter() - a)
print("per-frame ms (synced each frame):", [round(x * 1e3, 2) for x in ts])
