#!/usr/bin/env python3
"""PCIe-inclusive frame rates of cfg3 (never bench.py's `value`): the frame plus the read-back of its present payload
(swr_readback_rgb, 12 B/pixel) into host memory, with retained meshes (inputs resident in HBM) and with the array
signature (vertices and indices cross PCIe on every RenderMesh call, as the reference's per-frame arguments would)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from softwarerenderer_amd import Device, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scene = getattr(scenes, cfg)()
dev = Device(0)
out = {}
for retained in (True, False):
    r = scenes.SceneRenderer(dev, scene, retained=retained)
    for _ in range(3):
        r.submit_frame(); dev.flush()
    dev.sync()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        r.submit_frame(); dev.flush()
    dev.sync()
    t_frame = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        r.submit_frame()
        rgb = r.window.FlatColorBuffer()
    t_rb = (time.perf_counter() - t0) / n
    pinned = np.empty_like(rgb)
    dev.pin(pinned)
    r.submit_frame(); r.window.FlatColorBuffer(out=pinned)
    t0 = time.perf_counter()
    for _ in range(n):
        r.submit_frame()
        r.window.FlatColorBuffer(out=pinned)
    t_pin = (time.perf_counter() - t0) / n
    assert np.array_equal(pinned, rgb)
    # asynchronous, double-buffered present (swr_present_rgb_async): frame i's copy overlaps frame i + 1's rendering, so a frame
    # costs max(render, copy) instead of their sum
    pinned2 = np.empty_like(rgb)
    dev.pin(pinned2)
    bufs, tickets = [pinned, pinned2], [None, None]
    for i in range(4):                                   # warm-up: both staging buffers exist afterwards
        r.submit_frame(); k = i & 1
        if tickets[k] is not None: r.window.PresentWait(tickets[k])
        tickets[k] = r.window.PresentAsync(bufs[k])
    t0 = time.perf_counter()
    for i in range(n):
        r.submit_frame(); k = i & 1
        if tickets[k] is not None: r.window.PresentWait(tickets[k])
        tickets[k] = r.window.PresentAsync(bufs[k])
    for k in (0, 1):
        r.window.PresentWait(tickets[k])
    t_async = (time.perf_counter() - t0) / n
    assert np.array_equal(pinned, rgb) and np.array_equal(pinned2, rgb)
    dev.unpin(pinned2)
    dev.unpin(pinned)
    key = "retained_meshes" if retained else "host_arrays_every_call"
    out[key] = {"frame_ms": round(1e3 * t_frame, 3), "frame_plus_rgb_readback_ms": round(1e3 * t_rb, 3),
                "frame_plus_rgb_readback_pinned_ms": round(1e3 * t_pin, 3),
                "frame_with_async_double_buffered_present_ms": round(1e3 * t_async, 3), "readback_mb": round(rgb.nbytes / 1e6, 1)}
    r.close()
print(json.dumps(out))
