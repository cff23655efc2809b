#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_api.py::test_build_identity_matches_the_verified_pair > gpurun_out/r4_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r4_tests.log
[ $rc -ne 0 ] && exit $rc
# where is the crossover?  cfg3-like frames at 1920x1080 with 32 k .. 1 M triangles, one stream against every batch pipelined
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from softwarerenderer_amd import Device, scenes
for quads in [(45, 22), (64, 32), (90, 45), (128, 64), (180, 90), (250, 125)]:
    scene = scenes.cfg3(1920, 1080, (4, 4), quads, tex_size=1024)
    res = {}
    for mode in (0, 2, 0, 2):
        dev = Device(0); dev.set_pipelining(mode); r = scenes.SceneRenderer(dev, scene)
        for _ in range(40): r.submit_frame(); dev.flush()
        dev.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(100): r.submit_frame(); dev.flush()
            dev.sync()
            best = min(best, 1e3 * (time.perf_counter() - t0) / 100)
        res.setdefault(mode, []).append(best)
        r.close(); dev.close()
    print(f"1920x1080 {scene.n_triangles:8d} triangles: one stream {min(res[0]):.4f} ms, frames in flight {min(res[2]):.4f} ms ({100 * (min(res[2]) / min(res[0]) - 1):+.1f} %)", flush=True)
PY
