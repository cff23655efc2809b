"""The two open System.Numerics ambiguities as TESTED switches.

.NET 9's SIMD Vector4.Transform / Lerp may fuse multiply-add on FMA-capable x64 (SWR_NUMERICS_FMA) and Vector3.Dot may sum
its lanes pairwise (SWR_DOT_PAIRWISE); neither can be settled without a .NET 9 box.  The HIP backend is therefore built
in every mode (csrc/Makefile `variants`) and each build must match the oracle built with the SAME switches: depth words
bit-exact, colour <= 1 ULP -- so that a maintainer who determines the true mode finds a backend that already passes in it.
Product default = unfused + sequential; the other builds are sensitivity builds, selected with Device(lib=...)."""
import os

import numpy as np
import pytest

from softwarerenderer_amd import Device, _native, scenes
from softwarerenderer_amd.rasterizer import Program
from util import assert_frame_parity

pytestmark = pytest.mark.gpu

MODES = [("libswr_hip_fma.so", "fma", 1, 0), ("libswr_hip_dotpw.so", "dotpw", 0, 2), ("libswr_hip_fma_dotpw.so", "fma_dotpw", 1, 2),
         ("libswr_hip_dpps.so", "dpps", 0, 1), ("libswr_hip_fma_dpps.so", "fma_dpps", 1, 1)]


def mode_scenes():
    yield scenes.cfg1()
    yield scenes.cfg2(640, 360, 3000)
    yield scenes.cfg3(768, 512, (3, 3), (40, 24), tex_size=256)
    yield scenes.near_clip_scene()                                  # Shaders.Lerp (the clipper's fused / unfused lerps)
    yield scenes.near_clip_scene(program=Program.Gouraud)
    yield scenes.cfg3(400, 300, (2, 2), (30, 20), tex_size=64, seed=11)


@pytest.fixture(scope="module", params=MODES, ids=[m[1] for m in MODES])
def mode(request):
    lib, variant, fma, dot = request.param
    if not os.path.exists(os.path.join(os.path.dirname(_native.LIB_PATH), lib)):
        pytest.fail(f"{lib} is missing: __graft_entry__.build() makes it (make -C softwarerenderer_amd/csrc variants)")
    dev = Device(0, lib=lib)
    yield dev, variant, fma, dot
    dev.close()


def test_mode_build_matches_the_oracle_built_with_the_same_switches(mode):
    from oracle import binding as ob
    dev, variant, fma, dot = mode
    olib = ob.load(variant=variant)
    assert (olib.oswr_numerics_fma(), olib.oswr_dot_pairwise()) == (fma, dot)
    differs_from_default = 0
    for scene in mode_scenes():
        r = scenes.SceneRenderer(dev, scene)
        c, d = r.render()
        st = dev.stats()
        r.close()
        o = ob.OracleRenderer(scene.width, scene.height, variant=variant)
        rc, rd = o.render_scene(scene)
        ost = o.stats(); o.close()
        assert_frame_parity(c, d, rc, rd, color_ulp=1, what=f"{variant}/{scene.name}")
        o0 = ob.OracleRenderer(scene.width, scene.height)
        c0, d0 = o0.render_scene(scene); o0.close()
        differs_from_default += int((d0.view(np.uint32) != rd.view(np.uint32)).sum()) + int((c0.view(np.uint32) != rc.view(np.uint32)).sum())
    # the switch is not a no-op: somewhere in these frames the mode changes bits, and the HIP build follows it
    # (the dpps order (xx + yy) + (zz + 0) differs from the sequential sum only in the sign of a zero: no pixel of these scenes)
    assert differs_from_default > 0 or variant == "dpps"
    assert dev.numerics_mode() == (fma, dot)
    assert dev.transform_fma() == (bool(fma), bool(fma))             # the run-time half defaults to the compile-time one


# (library, oracle variant, Transform fused?, TransformNormal fused?): the run-time half of the model set AGAINST the library's
# compile-time Lerp switch -- the combinations round 3's probe had to refuse (Missing #2)
MIXED = [("libswr_hip.so", "", 1, 0), ("libswr_hip.so", "", 0, 1), ("libswr_hip.so", "", 1, 1),
         ("libswr_hip_fma.so", "fma", 0, 0), ("libswr_hip_fma.so", "fma", 1, 0), ("libswr_hip_dotpw.so", "dotpw", 1, 1),
         ("libswr_hip_fma_dpps.so", "fma_dpps", 0, 1)]


@pytest.mark.parametrize("lib,variant,tr,tn", MIXED, ids=[f"{m[1] or 'default'}-t{m[2]}n{m[3]}" for m in MIXED])
def test_run_time_transform_flags_against_the_oracle_set_alike(lib, variant, tr, tn):
    from oracle import binding as ob
    dev = Device(0, lib=lib)
    try:
        dev.set_transform_fma(tr, tn)
        assert dev.transform_fma() == (bool(tr), bool(tn))
        moved = 0
        for scene in mode_scenes():
            r = scenes.SceneRenderer(dev, scene)
            c, d = r.render()
            r.close()
            o = ob.OracleRenderer(scene.width, scene.height, variant=variant, transform_fma=(tr, tn))
            rc, rd = o.render_scene(scene); o.close()
            assert_frame_parity(c, d, rc, rd, color_ulp=1, what=f"{variant or 'default'}/t{tr}n{tn}/{scene.name}")
            o0 = ob.OracleRenderer(scene.width, scene.height, variant=variant)          # the library's own default flags
            c0, d0 = o0.render_scene(scene); o0.close()
            moved += int((d0.view(np.uint32) != rd.view(np.uint32)).sum()) + int((c0.view(np.uint32) != rc.view(np.uint32)).sum())
        assert moved > 0                      # the flags are not a no-op on these frames, and the HIP build follows them
    finally:
        dev.close()
