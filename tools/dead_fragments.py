#!/usr/bin/env python3
"""How many written fragments of a BASELINE frame are dead, bit for bit?  (Next #2a -- CPU only.)

A written fragment is DEAD when a later written fragment of the same pixel has alpha exactly 1.0f under BlendMode.Alpha
(Rasterizer.cs:58-65): src * 1 + dst * 0 == src for finite dst (and no zero component in src whose sign dst * 0 could flip), so
nothing of the earlier colour survives.  Counted on a build of the oracle of its own (-DOSWR_DEAD_COUNT, oracle/liboswr_deadcount.so), serial order.
Prints one JSON line per scene: written, dead, the fraction, how many written fragments have alpha == 1.0f at all.
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    so = os.path.join(ROOT, "oracle", "liboswr_deadcount.so")
    subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fexcess-precision=standard", "-msse2",
                    "-mfpmath=sse", "-DSWR_NUMERICS_FMA=0", "-DOSWR_DEAD_COUNT", os.path.join(ROOT, "oracle", "swr_oracle.c"), "-o", so,
                    "-shared", "-lm", "-lpthread"], check=True)
    from oracle import binding
    from softwarerenderer_amd import scenes
    names = sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]
    for name in names:
        scene = getattr(scenes, name)()
        o = binding.OracleRenderer(scene.width, scene.height, threads=1, variant="deadcount")
        o.lib.oswr_dead_count_reset()
        o.render_scene(scene)
        g = lambda n: C.c_ulonglong.in_dll(o.lib, n).value
        w, d, a1, k = g("oswr_dc_written"), g("oswr_dc_dead"), g("oswr_dc_alpha_one"), g("oswr_dc_killers")
        st = o.stats()
        assert w == st["fragments_written"], (w, st)
        print(json.dumps({"scene": name, "written": w, "pixels_touched_upper_bound": scene.width * scene.height, "alpha_exactly_one": a1,
                          "alpha_exactly_one_frac": round(a1 / w, 4), "killers": k, "dead": d, "dead_frac_of_written": round(d / w, 4),
                          "shading_share_of_chunk": "215/375", "bound_on_raster_time": round(d / w * 215 / 375, 4)}), flush=True)
        o.close()


if __name__ == "__main__":
    main()
