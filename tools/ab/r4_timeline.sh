#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace) of the final build with frames in flight: cfg2 (pipelined by the default mode) and cfg3 under mode 2
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for run in "cfg2 1" "cfg3 2"; do
  set -- $run
  rm -rf gpurun_out/prof_tl
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --config $1 --steps 8 --warmup 3 --prime 6 --no-cpu-baseline --no-profile-events --pipelining $2 > gpurun_out/r4_tl_$1.log 2>&1 || exit 1
  find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_timeline_$1.csv
  rm -rf gpurun_out/prof_tl
done
python - <<'PY'
import csv
for cfg in ("cfg2", "cfg3"):
    rows=list(csv.DictReader(open(f"gpurun_out/r4_timeline_{cfg}.csv")))
    rows=[r for r in rows if "swr::" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # the timed region = the last burst of 8 raster kernels that follow each other closely
    ras=[i for i,r in enumerate(rows) if "k_raster_c" in r["Kernel_Name"]]
    first=ras[-6]
    t0=int(rows[first]["Start_Timestamp"])
    print("###", cfg)
    for r in rows[first: ras[-3] + 1]:
        n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("swr::","")[:28]
        print(f'| {(int(r["Start_Timestamp"])-t0)/1e3:8.1f} | {(int(r["End_Timestamp"])-t0)/1e3:8.1f} | {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:6.1f} | q{r.get("Queue_Id","?")} | `{n}` |')
PY
