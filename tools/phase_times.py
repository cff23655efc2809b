#!/usr/bin/env python3
"""Where a k_raster_c wave spends its LIFE (not its instructions): builds the library with -DSWR_DEBUG_PHASES into build_ab/, renders a
few frames of a config and prints the shader-clock ticks per phase, summed over all waves and per tile / batch / chunk
(swr_raster_c.hip.h, SWR_PHASE).  usage: python tools/phase_times.py [cfg3] [extra hipcc flags ...]   (needs a GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
NAMES = ["tile init", "batch: window + hi-Z + selection", "batch: staging", "chunk: lookup", "chunk: election + cut",
         "chunk: replay + depth", "chunk: shade + blend + next lookup", "write-back + statistics"]


def build(extra, out):
    flags = subprocess.run(["make", "-s", "-C", csrc, "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    subprocess.run(["/opt/rocm/bin/hipcc"] + extra + flags + ["swr_api.hip", "-o", out], cwd=csrc, check=True)


def counters(lib, cfg, dbg):
    from softwarerenderer_amd import _native
    _native.LIB_PATH = lib
    from softwarerenderer_amd import Device, scenes
    scene = scenes.cfg3(bilinear=True) if cfg == "cfg3_bilinear" else getattr(scenes, cfg)()
    dev = Device(0)
    dev.set_pipelining(0)
    r = scenes.SceneRenderer(dev, scene)
    for _ in range(3):
        r.render()
    out = (C.c_uint64 * 8)()
    dev._lib.swr_debug_counters(dev._ctx, out)          # reads and clears
    N = 5
    for _ in range(N):
        r.submit_frame(); dev.flush()
    dev.sync()
    dev._lib.swr_debug_counters(dev._ctx, out)
    r.close(); dev.close()
    return [int(v) / N for v in out]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    extra = sys.argv[2:]
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    ph = os.path.join(ROOT, "build_ab", "phases.so")
    cn = os.path.join(ROOT, "build_ab", "counters.so")
    build(["-DSWR_DEBUG_PHASES"] + extra, ph)
    build(["-DSWR_DEBUG_COUNTERS"] + extra, cn)
    import json
    child = "import sys, json; sys.path.insert(0, %r); from tools.phase_times import counters; print(json.dumps(counters(sys.argv[1], sys.argv[2], 0)))" % ROOT
    t = [64.0 * v for v in json.loads(subprocess.run([sys.executable, "-c", child, ph, cfg], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])]     # every 64th wave reports
    c = json.loads(subprocess.run([sys.executable, "-c", child, cn, cfg], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    batches, chunks = c[0], c[1]
    total = sum(t)
    print(f"{cfg} {' '.join(extra)}: {batches:.0f} batches, {chunks:.0f} chunks per frame; wave ticks per frame {total:.3e}")
    for i, name in enumerate(NAMES):
        per = ""
        if i in (1, 2):
            per = f"{t[i] / max(batches, 1):9.0f} per batch"
        elif 3 <= i <= 6:
            per = f"{t[i] / max(chunks, 1):9.0f} per chunk"
        print(f"  {i} {name:36s} {100 * t[i] / total:5.1f} %  {per}")


if __name__ == "__main__":
    main()
