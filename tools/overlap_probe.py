"""Upper bound of what overlapping one frame's front end with another frame's raster can give: two independent contexts
(own streams, own buffers) render cfg3 frames alternately without syncing in between, against one context rendering the same number."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from softwarerenderer_amd import Device, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = 40
scene = getattr(scenes, cfg)()
devs = [Device(0), Device(0)]
rs = [scenes.SceneRenderer(d, scene) for d in devs]
for r, d in zip(rs, devs):
    for _ in range(3): r.submit_frame(); d.flush()
    d.sync()
def run(which, frames):
    t0 = time.perf_counter()
    for i in range(frames):
        k = which[i % len(which)]
        rs[k].submit_frame(); devs[k].flush()
    for d in devs: d.sync()
    return (time.perf_counter() - t0) / frames * 1e3
for rep in range(3):
    a = run([0], N); b = run([0, 1], N); c = run([1], N)
    print(f"{cfg}: one context {a:.4f} ms/frame, two contexts alternating {b:.4f} ms/frame, other context alone {c:.4f}")
