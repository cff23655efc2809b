#!/bin/bash
# does a front-end kernel WITHOUT LDS run beside the raster kernel?  (k_vertex with direct loads / stores: 0 B of LDS)
set -o pipefail
mkdir -p gpurun_out build_ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -C softwarerenderer_amd/csrc -s -B EXTRA="-DSWR_VERTEX_DIRECT_STORES -DSWR_VERTEX_DIRECT_LOADS" ../libswr_hip.so || exit 1
rm -rf gpurun_out/prof_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pipe -- python3 bench.py --steps 8 --warmup 3 --prime 6 --no-cpu-baseline --no-profile-events --pipelining 1 > gpurun_out/r4_trace2_run.log 2>&1 || exit 1
find gpurun_out/prof_pipe -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_pipe2_kernel_trace.csv
rm -rf gpurun_out/prof_pipe
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4_pipe2_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
rows=[r for r in rows if "swr::" in r["Kernel_Name"]]
for r in rows[-40:]:
    n=r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","?")} {n}')
PY
