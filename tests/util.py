"""Comparison helpers shared by the parity tests."""
import numpy as np


def ulp_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in units in the last place between float32 arrays; NaN vs NaN counts as 0."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)     # sign-magnitude -> monotone integer line
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    one_nan = np.isnan(a) ^ np.isnan(b)
    d = np.where(both_nan, 0, d)
    d = np.where(one_nan, 1 << 40, d)
    return d


def assert_frame_parity(color, depth, ref_color, ref_depth, color_ulp=1, what=""):
    """The north-star bar: depth words bit-exact, colour within `color_ulp` ULP per channel."""
    assert color.shape == ref_color.shape and depth.shape == ref_depth.shape, what
    dz = depth.view(np.uint32) != ref_depth.view(np.uint32)
    if dz.any():
        ys, xs = np.nonzero(dz)
        y, x = int(ys[0]), int(xs[0])
        raise AssertionError(f"{what}: {int(dz.sum())} depth words differ; first at (x={x}, y={y}): "
                             f"got {depth[y, x]!r} ({depth.view(np.uint32)[y, x]:#010x}) "
                             f"want {ref_depth[y, x]!r} ({ref_depth.view(np.uint32)[y, x]:#010x})")
    d = ulp_distance(color, ref_color)
    bad = d > color_ulp
    if bad.any():
        ys, xs, cs = np.nonzero(bad)
        y, x = int(ys[0]), int(xs[0])
        raise AssertionError(f"{what}: {int(bad.sum())} colour channels differ by more than {color_ulp} ULP "
                             f"(max {int(d.max())}); first at (x={x}, y={y}): got {color[y, x]} want {ref_color[y, x]}")
    return int((d > 0).sum())


def render_oracle(scene, debug_mode=0, threads=1):
    """SWR_ORACLE_VARIANT (fma / dotpw / fma_dotpw / dpps) selects the oracle build that matches a System.Numerics sensitivity
    build of the backend (SWR_LIB=libswr_hip_<variant>.so): used by the sweeps in tools/, never by the suite's own defaults."""
    import os
    from oracle.binding import OracleRenderer
    o = OracleRenderer(scene.width, scene.height, threads=threads, variant=os.environ.get("SWR_ORACLE_VARIANT") or None)
    c, d = o.render_scene(scene, debug_mode)
    st = o.stats()
    o.close()
    return c, d, st
