#!/usr/bin/env python3
"""Builds a -DSWR_DEBUG_COUNTERS copy of the library into /tmp, renders one cfg frame and prints k_raster_c's work counters (batches, chunks, chunk fill)."""
import ctypes as C, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
lib = os.path.join(ROOT, "softwarerenderer_amd", "libswr_hip.so")
bak = lib + ".bak"
shutil.copy(lib, bak)
try:
    cover = "--cover" in sys.argv          # k_cover's walk instead: useful pixel steps against executed ones (extra flags may follow)
    argv = [x for x in sys.argv[1:] if x != "--cover"]
    extra = " ".join(argv[1:])
    subprocess.run(["make", "-C", csrc, "-s", "-B", "EXTRA=" + ("-DSWR_DEBUG_COVER " + extra if cover else "-DSWR_DEBUG_COUNTERS")], check=True)
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
    if cover:
        useful, executed, waves = int(out[0]), int(out[1]), int(out[2])
        print(cfg, extra, "k_cover walk: lane steps useful", useful, "executed", executed, "ratio", round(executed / max(useful, 1), 3),
              "steps per wave", round(executed / 64 / max(waves, 1), 1), "useful per lane", round(useful / 64 / max(waves, 1), 1))
        raise SystemExit(0)
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
