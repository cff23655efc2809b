#!/usr/bin/env python3
"""Builds a -DSWR_DEBUG_COUNTERS copy of the library into /tmp, renders one cfg frame and prints k_raster_c's work counters (batches, chunks, chunk fill)."""
import ctypes as C, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
lib = os.path.join(ROOT, "softwarerenderer_amd", "libswr_hip.so")
bak = lib + ".bak"
shutil.copy(lib, bak)
try:
    argv = sys.argv[1:]
    # (the variant carries its switches in swr_build_info's `extra=`: left in the product's place it fails the identity test)
    subprocess.run(["make", "-C", csrc, "-s", "-B", "EXTRA=-DSWR_DEBUG_COUNTERS"], check=True)
    from softwarerenderer_amd import Device, scenes
    cfg = argv[0] if argv else "cfg3"
    scene = getattr(scenes, cfg)()
    dev = Device(0)
    r = scenes.SceneRenderer(dev, scene)
    r.render()
    out = (C.c_uint64 * 8)()
    dev._lib.swr_debug_counters(dev._ctx, out)
    r.render()
    dev._lib.swr_debug_counters(dev._ctx, out)
    st = dev.stats()
    names = ["batches", "chunks", "max_col_steps", "max_row_steps", "chunk_lanes", "sum_row_steps", "sum_col_steps", "hiz_hidden_fragments"]
    d = dict(zip(names, [int(v) for v in out]))
    print(cfg, d)
    print("pairs", st["tile_pairs"] // max(st["flushes"], 1), "batches", d["batches"], "chunks", d["chunks"],
          "fragments per chunk", d["chunk_lanes"] / max(d["chunks"], 1),
          "wave-max row steps per chunk", d["max_row_steps"] / max(d["chunks"], 1), "col", d["max_col_steps"] / max(d["chunks"], 1),
          "fragments dropped by hi-Z", d.get("hiz_hidden_fragments", 0), "of depth-failing", st["fragments_tested"] // max(st["flushes"], 1) - st["fragments_shaded"] // max(st["flushes"], 1),
          "mean row steps per fragment", d["sum_row_steps"] / max(d["chunk_lanes"], 1), "col", d["sum_col_steps"] / max(d["chunk_lanes"], 1))
finally:
    shutil.move(bak, lib)
