#!/bin/bash
# raster kernel at 96 VGPRs (dynamic LDS) against 106 (static, build_ab/r4_static106.so): isolated kernel time, then frames in flight
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/ablate.py cfg3 "lib:build_ab/r4_static106.so" "lib:softwarerenderer_amd/libswr_hip.so" "lib:build_ab/r4_static106.so" "lib:softwarerenderer_amd/libswr_hip.so" 2>&1 | tee gpurun_out/r4_dyn_ablate.txt
for lib in build_ab/r4_static106.so softwarerenderer_amd/libswr_hip.so; do
for cfg in cfg3 cfg2; do
for p in 1 0; do
  SWR_LIB_PATH=$lib timeout -k 10 200 python - "$cfg" "$p" "$lib" <<'PY' || exit 1
import sys, os, json, time
sys.path.insert(0, os.getcwd())
from softwarerenderer_amd import _native
_native.LIB_PATH = os.path.join(os.getcwd(), sys.argv[3])
from softwarerenderer_amd import Device, scenes
scene = getattr(scenes, sys.argv[1])()
dev = Device(0); dev.set_pipelining(int(sys.argv[2])); r = scenes.SceneRenderer(dev, scene)
for _ in range(40): r.submit_frame(); dev.flush()
dev.sync()
N = 100
t0 = time.perf_counter()
for _ in range(N): r.submit_frame(); dev.flush()
dev.sync()
print(sys.argv[3], sys.argv[1], "pipelining", sys.argv[2], "ms/frame %.4f" % (1e3 * (time.perf_counter() - t0) / N), flush=True)
PY
done; done; done
rm -rf gpurun_out/prof_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pipe -- python3 bench.py --steps 8 --warmup 3 --prime 6 --no-cpu-baseline --no-profile-events --pipelining 1 > gpurun_out/r4_trace4_run.log 2>&1 || exit 1
find gpurun_out/prof_pipe -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_pipe4_kernel_trace.csv
rm -rf gpurun_out/prof_pipe
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4_pipe4_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
rows=[r for r in rows if "swr::" in r["Kernel_Name"]]
for r in rows[-22:]:
    n=r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","?")} {n}')
PY
