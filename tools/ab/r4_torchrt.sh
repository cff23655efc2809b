#!/bin/bash
set -o pipefail
for p in 0 2 0 2; do timeout -k 10 200 python tools/ab/frames_torch.py softwarerenderer_amd/libswr_hip.so cfg3 $p 100 2>&1 | tail -1 || exit 1; done
for p in 0 2; do timeout -k 10 200 python tools/ab/frames_torch.py softwarerenderer_amd/libswr_hip.so cfg3 $p 20 2>&1 | tail -1 || exit 1; done
for p in 0 2; do timeout -k 10 200 python tools/ab/frames.py softwarerenderer_amd/libswr_hip.so cfg3 $p 20 2>&1 | tail -1 || exit 1; done
python - <<'PY'
import os
print([l.split()[-1] for l in open("/proc/self/maps") if "hip" in l][:2])
PY
