#!/usr/bin/env python3
"""Regenerates tests/golden/fullsize_golden.json: the serial CPU oracle (oracle/swr_oracle.c) run on BASELINE.json's
configurations at their STATED sizes -- cfg2 1920x1080/10k, cfg3 4096^2/1M, cfg4 4096^2/1M Phong, cfg5 8192^2/1M.

Data only (no reference text): per configuration the sha256 of the colour and depth buffers, the sha256 of every band of
`band_rows` pixel rows (cfg5: 1024 = the eight tile-row bands of multigpu.band_partition(8192, 8)), the oracle's six
counters and 64 sampled pixels.  The reference holds no fixtures and cannot be run here: these pin
the HIP path to the build's own restatement at full size ("parity unpinned" against the C# itself).
Takes a few minutes of one CPU core and ~6 GB of memory (cfg5)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob                      # noqa: E402
from softwarerenderer_amd import scenes               # noqa: E402

CONFIGS = {"cfg2": (scenes.cfg2, 120), "cfg3": (scenes.cfg3, 512), "cfg4": (scenes.cfg4, 512), "cfg5": (scenes.cfg5, 1024),
           # cfg3 through the BUILD-DEFINED bilinear filter (BASELINE.json says "bilinear-textured"; the reference samples nearest):
           # pinned to the build's own definition only (oracle/swr_oracle.c:oswr_texture_sample_bilinear)
           "cfg3_bilinear": (lambda: scenes.cfg3(bilinear=True), 512)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_points(width, height, n=64, seed=20261004):
    rng = np.random.default_rng(seed)
    return rng.integers(0, width, n), rng.integers(0, height, n)


def digest(name):
    make, band_rows = CONFIGS[name]
    scene = make()
    t0 = time.time()
    o = ob.OracleRenderer(scene.width, scene.height)          # threads=1: the serial schedule is the oracle of record
    c, d = o.render_scene(scene)
    st = o.stats()
    o.close()
    xs, ys = sample_points(scene.width, scene.height)
    out = {
        "scene": scene.name, "width": scene.width, "height": scene.height, "triangles": scene.n_triangles,
        "color_sha256": sha(c), "depth_sha256": sha(d), "stats": st, "band_rows": band_rows,
        "band_color_sha256": [sha(c[y:y + band_rows]) for y in range(0, scene.height, band_rows)],
        "band_depth_sha256": [sha(d[y:y + band_rows]) for y in range(0, scene.height, band_rows)],
        "samples": [{"x": int(x), "y": int(y), "color_bits": [int(v) for v in c[y, x].view(np.uint32)],
                     "depth_bits": int(d[y, x].view(np.uint32))} for x, y in zip(xs, ys)],
    }
    print(f"{name}: {scene.name} {st} in {time.time() - t0:.1f} s", flush=True)
    return out


if __name__ == "__main__":
    ob.build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_golden.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name in (sys.argv[1:] or list(CONFIGS)):
        out[name] = digest(name)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(f"wrote {path}: {sorted(out)}")
