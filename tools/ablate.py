#!/usr/bin/env python3
"""Timing-only A/B of kernel variants: for each EXTRA flag set, rebuild the library, run N frames of a config and print
per-stage hipEvent times.  Outputs of ablated builds are wrong by design; the tracked library is restored afterwards.
usage: ablate.py cfg3 "" "-DSWR_ABLATE_PHASE2" "lib:build_ab/ref.so" ...
A variant "env:NAME=VALUE[,..]" times the in-tree library as it is under those environment variables.  A variant "lib:PATH" times a prebuilt library instead (boxes differ by a few percent: keep a reference library in
build_ab/ -- git-ignored, but it travels with gpurun -- and compare inside ONE call)."""
import os, shutil, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
lib = os.path.join(ROOT, "softwarerenderer_amd", "libswr_hip.so")
cfg = sys.argv[1]; variants = sys.argv[2:] or [""]
child = r'''
import sys, json, os; sys.path.insert(0, %r)
from softwarerenderer_amd import _native
if os.environ.get("SWR_ABLATE_LIB"): _native.LIB_PATH = os.environ["SWR_ABLATE_LIB"]
from softwarerenderer_amd import Device, scenes
scene = getattr(scenes, %r)()
dev = Device(0); r = scenes.SceneRenderer(dev, scene)
for _ in range(3): r.submit_frame(); dev.flush()
dev.sync(); dev.profile_reset(); dev.profile_enable(True)
N = int(os.environ.get('ABLATE_N', '10'))
for _ in range(N): r.submit_frame(); dev.flush()
p = dev.profile(); print(json.dumps({k: round(v / N, 4) for k, v in p.items() if k.endswith("_ms")}))
''' % (ROOT, cfg)
shutil.copy(lib, lib + ".bak")
try:
    for v in variants:
        env = dict(os.environ)
        if v.startswith("lib:"): env["SWR_ABLATE_LIB"] = os.path.join(ROOT, v[4:])
        elif v.startswith("env:"):       # "env:NAME=VALUE,NAME2=VALUE2": the in-tree library as it is, run under these variables
            env.update(kv.split("=", 1) for kv in v[4:].split(","))
        else: subprocess.run(["make", "-C", csrc, "-s", "-B", f"EXTRA={v}"], check=True, stderr=subprocess.DEVNULL)
        out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env)
        print(f"{v or '(release)':40s}", out.stdout.strip() or out.stderr[-400:])
finally:
    shutil.move(lib + ".bak", lib)
