#!/usr/bin/env python3
"""Blast radius of the two open System.Numerics ambiguities on BASELINE's frames, CPU only:
renders cfg2 / cfg3 (and the near-clip scene, which runs Shaders.Lerp) with every oracle build of oracle/Makefile and
counts, against the default build (unfused, sequential dot), the depth words that differ and the colour channels that
differ (with the largest ULP distance).  The table goes into DESIGN.md section 3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob
from softwarerenderer_amd import scenes
from util import ulp_distance

ob.build()
SCENES = [("cfg2 1920x1080", scenes.cfg2), ("cfg3 4096x4096", scenes.cfg3), ("near-clip 320x200", scenes.near_clip_scene)]
if len(sys.argv) > 1 and sys.argv[1] == "--small":
    SCENES = [("cfg2 480x270", lambda: scenes.cfg2(480, 270, 2000)), ("cfg3 512x512", lambda: scenes.cfg3(512, 512, (4, 4), (32, 16), tex_size=256)),
              ("near-clip 320x200", scenes.near_clip_scene)]
print("| scene | build | depth words that differ | colour channels that differ | max colour ULP | written fragments |")
print("|---|---|---|---|---|---|")
for label, make in SCENES:
    scene = make()
    base = None
    for variant, what in (("", "default: unfused, sequential dot"), ("fma", "SWR_NUMERICS_FMA=1"), ("dpps", "SWR_DOT_PAIRWISE=1 (dpps order)"),
                          ("dotpw", "SWR_DOT_PAIRWISE=2 (shuffle-adds)"), ("fma_dotpw", "FMA=1 + DOT_PAIRWISE=2")):
        o = ob.OracleRenderer(scene.width, scene.height, variant=variant)
        c, d = o.render_scene(scene); st = o.stats(); o.close()
        if base is None:
            base = (c, d)
            print(f"| {label} | {what} | - | - | - | {st['fragments_written']} |", flush=True)
            continue
        dz = int((d.view(np.uint32) != base[1].view(np.uint32)).sum())
        u = ulp_distance(c, base[0])
        print(f"| {label} | {what} | {dz} of {d.size} | {int((u > 0).sum())} of {u.size} | {int(u.max())} | {st['fragments_written']} |", flush=True)
