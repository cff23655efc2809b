"""The co-residency budget of frames in flight, checked in what the compiler actually allocated (hipcc cross-compiles without a GPU).

The front end of flush N+1 runs BESIDE the raster kernel of flush N only if each of its blocks fits what four raster waves per SIMD
leave on a CU (csrc/swr_device.h): the raster kernels hold all 160 KB of LDS in slots of 10,240 B and at most 104 allocated VGPRs
each (4 x 104 = 416 of 512), so a front-end block needs <= 10,240 B of LDS, <= 96 VGPRs per wave and <= 4 waves.  One register more
in the raster kernel (105 -> 112 allocated: 64 left) silently sends k_bin (69 / 79) and k_setup (78) back to waiting for the raster
kernel's tail: a 5 % slower frame and no failing pixel.  Hence this test (profiles/r04_frames_in_flight.md)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _defines():
    txt = open(os.path.join(CSRC, "swr_device.h")).read()
    return {k: int(v) for k, v in re.findall(r"#define (SWR_FRONT_MAX_LDS|SWR_FRONT_MAX_VGPRS|SWR_RASTER_MAX_VGPRS|SWR_GEOM_BLOCK) (\d+)", txt)}


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this machine")
    mk = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], check=True, capture_output=True, text=True).stdout.split()
    flags = [f for f in mk if f != "-shared" and not f.startswith("-Wl,")]
    obj = str(tmp_path_factory.mktemp("ru") / "swr_api.o")
    err = subprocess.run([HIPCC] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "swr_api.hip", "-o", obj], cwd=CSRC,
                         capture_output=True, text=True, check=True).stderr
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("void ", "")
            cur = re.sub(r"\(.*", "", cur)
            out[cur] = {}
            continue
        m = re.search(r"remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            out[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def _alloc(v):            # VGPRs are allocated in granules of 8 on gfx950 (unified file of 512 per SIMD lane)
    return (v + 7) // 8 * 8


def test_the_raster_kernels_that_matter_leave_96_registers_per_simd(usage):
    d = _defines()
    assert d["SWR_RASTER_MAX_VGPRS"] == 104 and d["SWR_FRONT_MAX_VGPRS"] == 512 - 4 * d["SWR_RASTER_MAX_VGPRS"]
    ras = {k: v for k, v in usage.items() if "k_raster_c<" in k}
    assert len(ras) >= 9
    # the staged-varyings kernels (DUST2 = the reference's own frame, Gouraud, generic): four waves per SIMD by LDS (10,208 B per wave)
    for k, v in ras.items():
        if "k_raster_c<false, false" in k:
            assert v["LDS"] <= 10240 and v["ScratchSize"] == 0, (k, v)
            assert _alloc(v["VGPRs"] + v["AGPRs"]) <= d["SWR_RASTER_MAX_VGPRS"], (k, v, "one register too many: the front end of the next frame no longer fits beside this kernel")
    # the 4-light kernel runs four waves of up to 128 registers on purpose (capped at 96 it spilled: +14 %): nothing runs beside it but
    # the tail overlap; every other kernel of that family stays at five waves (96)
    for k, v in ras.items():
        if "k_raster_c<false, true, 3, 1, 2" in k:
            assert _alloc(v["VGPRs"]) <= 128 and v["ScratchSize"] == 0, (k, v)


def test_every_front_end_kernel_fits_the_slot_of_a_retiring_raster_wave(usage):
    d = _defines()
    front = ("swr::k_vertex", "swr::k_setup", "swr::k_bin<false>", "swr::k_bin<true>", "swr::k_scan_sums", "swr::k_scan_apply",
             "swr::k_sort_tiles", "swr::k_cover<false>", "swr::k_cover<true>", "swr::k_frustum_cull")
    for k in front:
        v = usage[k]
        assert v["LDS"] <= d["SWR_FRONT_MAX_LDS"], (k, v)
        assert _alloc(v["VGPRs"] + v["AGPRs"]) <= d["SWR_FRONT_MAX_VGPRS"], (k, v)
    # <= 4 waves per block: the block sizes are compile-time constants of the headers
    txt = open(os.path.join(CSRC, "swr_binning.hip.h")).read() + open(os.path.join(CSRC, "swr_raster_c.hip.h")).read()
    assert d["SWR_GEOM_BLOCK"] <= 256
    assert re.search(r"#define SWR_SCAN_BLOCK 256\b", txt) and re.search(r"#define SWR_COVER_BLOCK 256\b", txt) and re.search(r"#define SWR_SORT_TPB 1\b", txt)
    assert "__launch_bounds__(256) void k_bin" in txt
