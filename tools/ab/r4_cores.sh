#!/bin/bash
# front-end kernels retuned to run beside the raster kernel: suite, kernel timeline, frame times with frames in flight on / off
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_api.py::test_build_identity_matches_the_verified_pair > gpurun_out/r4_tests2.log 2>&1; rc=$?
tail -4 gpurun_out/r4_tests2.log
[ $rc -ne 0 ] && exit $rc
for cfg in cfg3 cfg2; do
for p in 1 0 1 0; do
  timeout -k 10 200 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline --no-profile-events --pipelining $p > gpurun_out/r4c_ne_${cfg}_p$p.json 2>gpurun_out/r4c_ne_${cfg}_p$p.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4c_ne_${cfg}_p$p.json") if l.startswith("{")][-1])
print("$cfg no-events p=$p ms/step", j["ms_per_step"], "value", j["value"])
PY
done
done
rm -rf gpurun_out/prof_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pipe -- python3 bench.py --steps 8 --warmup 3 --prime 6 --no-cpu-baseline --no-profile-events --pipelining 1 > gpurun_out/r4_trace3_run.log 2>&1 || exit 1
find gpurun_out/prof_pipe -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_pipe3_kernel_trace.csv
rm -rf gpurun_out/prof_pipe
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4_pipe3_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
rows=[r for r in rows if "swr::" in r["Kernel_Name"]]
for r in rows[-32:]:
    n=r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","?")} {n}')
PY
