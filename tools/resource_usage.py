#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the product build (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/resource_usage.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "softwarerenderer_amd", "csrc")


def main():
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    flags = [f for f in flags if f != "-shared" and not f.startswith("-Wl,")]
    cmd = ["/opt/rocm/bin/hipcc"] + sys.argv[1:] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "swr_api.hip", "-o", "/tmp/swr_api_ru.o"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", name).replace("void ", "")
            rows[cur] = {}
            continue
        m = re.search(r"remark: +(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'LDS/blk':>8s} {'waves/SIMD':>10s}")
    for k, v in rows.items():
        print(f"{k[:58]:58s} {v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('TotalSGPRs', 0):5d} {v.get('ScratchSize', 0):7d} {v.get('LDS', 0):8d} {v.get('Occupancy', 0):10d}")


if __name__ == "__main__":
    main()
