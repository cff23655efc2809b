#!/usr/bin/env python3
"""Builds a -DSWR_DEBUG_COUNTERS copy of the library into /tmp, renders one cfg frame and prints the raster kernel's work counters (historical: the counters lived in k_raster_b, removed since)."""
import ctypes as C, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
lib = os.path.join(ROOT, "softwarerenderer_amd", "libswr_hip.so")
bak = lib + ".bak"
shutil.copy(lib, bak)
try:
    subprocess.run(["make", "-C", csrc, "-s", "-B", "EXTRA=-DSWR_DEBUG_COUNTERS"], check=True)
    from softwarerenderer_amd import Device, scenes
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    scene = getattr(scenes, cfg)()
    dev = Device(0)
    r = scenes.SceneRenderer(dev, scene)
    r.render()
    out = (C.c_uint64 * 8)()
    dev._lib.swr_debug_counters(dev._ctx, out)
    r.render()
    dev._lib.swr_debug_counters(dev._ctx, out)
    st = dev.stats()
    names = ["batches", "chunks", "p1_max_iters", "chain_max_iters", "chunk_lanes", "tris"]
    d = dict(zip(names, [int(v) for v in out]))
    print(cfg, d)
    print("pairs", d["tris"], "tris/batch", d["tris"] / max(d["batches"], 1), "p1 iters/batch", d["p1_max_iters"] / max(d["batches"], 1),
          "frags/chunk", d["chunk_lanes"] / max(d["chunks"], 1), "chain iters/chunk", d["chain_max_iters"] / max(d["chunks"], 1))
finally:
    shutil.move(bak, lib)
