#!/usr/bin/env python3
"""Writes profiles/verified_build.json = the (compiler, kernel source) pair of the in-tree libswr_hip.so.  Run it after the
parity suite and the sweeps (tools/parity_sweep.py, tools/parity_sweep_modes.py) have been green ON THIS BUILD: the unfenced
wave-synchronous LDS hand-offs of k_cover / k_raster_c are only claimed for that pair
(tests/test_gpu_api.py::test_build_identity_matches_the_verified_pair)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "softwarerenderer_amd", "libswr_hip.so"))
lib.swr_build_info.restype = ctypes.c_char_p
fields = dict(kv.split("=", 1) for kv in lib.swr_build_info().decode().split("; "))
out = {"hipcc": fields["hipcc"], "csrc_sha256": fields["csrc_sha256"],
       "verified_by": sys.argv[1] if len(sys.argv) > 1 else "pytest -m gpu + tools/parity_sweep.py + tools/parity_sweep_modes.py"}
with open(os.path.join(ROOT, "profiles", "verified_build.json"), "w") as f:
    json.dump(out, f, indent=1); f.write("\n")
print(json.dumps(out))
