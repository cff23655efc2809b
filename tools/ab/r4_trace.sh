#!/bin/bash
# kernel timeline of pipelined frames (do front end and raster kernel overlap?) + event-free frame times
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 1 0 1 0; do
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-profile-events --pipelining $p > gpurun_out/r4_ne_cfg3_p$p.json 2>gpurun_out/r4_ne_cfg3_p$p.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ne_cfg3_p$p.json") if l.startswith("{")][-1])
print("cfg3 no-events p=$p ms/step", j["ms_per_step"])
PY
done
rm -rf gpurun_out/prof_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pipe -- python3 bench.py --steps 8 --warmup 3 --prime 6 --no-cpu-baseline --no-profile-events --pipelining 1 > gpurun_out/r4_trace_run.log 2>&1 || exit 1
find gpurun_out/prof_pipe -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_pipe_kernel_trace.csv
rm -rf gpurun_out/prof_pipe
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4_pipe_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
tail=rows[-70:]
for r in tail:
    n=r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","?")} {n}')
PY
