"""The C-ABI library loads and exports every symbol include/swr.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from softwarerenderer_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "swr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_declare_the_same_entry_points():
    assert header_functions() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, missing
    lib.swr_abi_version.restype = ctypes.c_int
    assert lib.swr_abi_version() == 3


def test_struct_layouts_match_the_reference_types():
    assert ctypes.sizeof(_native.Vertex) == 48                 # Shaders.VertexInput, Shaders.cs:10-24
    assert _native.Vertex.uv.offset == 12 and _native.Vertex.normal.offset == 20 and _native.Vertex.color.offset == 32
    assert ctypes.sizeof(_native.Uniforms) == 4 * (4 + 4 + 4 + 4 + 4) + 4 * 32
    assert ctypes.sizeof(_native.Stats) == 64


def test_no_device_fails_loudly_instead_of_falling_back():
    """On a box without a usable GPU the product path must raise, never compute on the CPU."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    from softwarerenderer_amd import Device
    with pytest.raises(_native.SwrError) as e:
        Device(0)
    assert e.value.code in (_native.SWR_ERR_NO_DEVICE, _native.SWR_ERR_HIP)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "softwarerenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboswr" not in txt, f


# ---- csharp/RasterizerNative.cs: the C# side of the boundary cannot be compiled here (no .NET), so it is checked mechanically ----
CS = open(os.path.join(ROOT, "csharp", "RasterizerNative.cs")).read()


def _split_params(plist):
    plist = plist.strip()
    if plist in ("", "void"):
        return []
    out, depth, cur = [], 0, ""
    for ch in plist:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "swr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return {m.group(1): _split_params(m.group(2)) for m in re.finditer(r"\b(swr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)}


def csharp_imports():
    return {m.group(2): _split_params(m.group(3))
            for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public static extern\s+([\w\*]+)\s+(swr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", CS)}


def test_csharp_binding_declares_exactly_the_entry_points_of_swr_h():
    protos, imports = header_prototypes(), csharp_imports()
    assert sorted(imports) == sorted(protos) == sorted(_native.EXPORTS)
    for name, params in protos.items():
        assert len(imports[name]) == len(params), (name, params, imports[name])
    # pointer-ness agrees argument by argument (a C pointer / array is `*`, `out`, or IntPtr on the C# side; a scalar is neither)
    for name, params in protos.items():
        for cp, sp in zip(params, imports[name]):
            c_is_ptr = "*" in cp or "[" in cp
            s_is_ptr = "*" in sp or sp.startswith("out ") or sp.startswith("IntPtr")
            assert c_is_ptr == s_is_ptr, (name, cp, sp)


CS_SIZES = {"float": 4, "double": 8, "ulong": 8, "int": 4, "Vector2": 8, "Vector3": 12, "Vector4": 16, "SwrPointLight": 32}


def cs_struct(name):
    body = re.search(r"public struct %s\b[^{]*\{(.*?)\n    \}" % name, CS, flags=re.S).group(1)
    body = re.sub(r"//.*", "", body)
    return [(m.group(2), m.group(1)) for m in re.finditer(r"public\s+(\w+)\s+(\w+)\s*;", body)]


def snake(n):
    return re.sub(r"(?<!^)(?=[A-Z])", "_", n).lower()


def test_csharp_struct_layouts_equal_the_ctypes_ones():
    pairs = {"SwrPointLight": _native.PointLight, "SwrUniforms": _native.Uniforms, "SwrStats": _native.Stats, "SwrProfile": _native.Profile}
    for cs_name, ct in pairs.items():
        fields = cs_struct(cs_name)
        assert sum(CS_SIZES[t] for _, t in fields) == ctypes.sizeof(ct), cs_name
        # field order: walk both in parallel by offset
        off, got = 0, []
        for n, t in fields:
            got.append((off, CS_SIZES[t])); off += CS_SIZES[t]
        want = []
        for fname, ftype in ct._fields_:
            f = getattr(ct, fname)
            if cs_name == "SwrUniforms" and fname == "lights":
                want += [(f.offset + 32 * i, 32) for i in range(4)]
            else:
                want.append((f.offset, f.size))
        assert got == want, (cs_name, got, want)
        names_cs = [snake(n).replace("pad", "_pad") for n, _ in fields if not n.startswith("Light") or cs_name != "SwrUniforms" or n in ("LightDirection", "LightColor")]
        names_ct = [n for n, _ in ct._fields_ if n != "lights"]
        assert names_cs == names_ct, (names_cs, names_ct)
    # Shaders.VertexInput is used as swr_vertex directly: 48 sequential bytes (Vector3, Vector2, Vector3, Vector4)
    assert 12 + 8 + 12 + 16 == ctypes.sizeof(_native.Vertex)


def test_csharp_shim_keeps_the_reference_signature_and_enum_ordinals():
    sig = re.search(r"public static unsafe void RenderMesh\(\s*MainWindow window,\s*Shaders\.VertexInput\[\] vertices,\s*ushort\[\] indices,\s*"
                    r"Matrix4x4 model,\s*Matrix4x4 view,\s*Matrix4x4 projection,\s*Shaders\.VertexShader vertexShader,\s*"
                    r"Shaders\.FragmentShader fragmentShader,\s*CullMode cullMode = CullMode\.Back,\s*DepthTest depthTest = DepthTest\.LessEqual,\s*"
                    r"BlendMode blendMode = BlendMode\.Alpha\)", CS)
    assert sig, "RenderMesh must keep the signature and defaults of Rasterizer.cs:163-174"
    hdr = open(os.path.join(ROOT, "include", "swr.h")).read()
    for cs, c in (("FlatColor = 0", "SWR_PROG_FLAT_COLOR = 0"), ("Gouraud = 1", "SWR_PROG_GOURAUD = 1"),
                  ("Dust2LambertFog = 2", "SWR_PROG_DUST2_LAMBERT_FOG = 2"), ("Phong4Point = 3", "SWR_PROG_PHONG_4POINT = 3"),
                  ("DebugVaryings = 4", "SWR_PROG_DEBUG_VARYINGS = 4")):
        assert cs in CS and c in hdr
    for fwd in ("SetPixel", "GetPixel", "ClearColorBuffer", "SetDepth", "GetDepth", "ClearDepthBuffer", "Resize", "Present"):
        assert re.search(r"public static \w+ %s\(" % fwd, CS), fwd          # MainWindow.cs:320-321,382-436 forwards
    assert "class TextureNative : IDisposable" in CS and "public Vector4 Sample(Vector2 uv)" in CS and "public void Dispose()" in CS


# ---- numerics start-up probe (csharp/RasterizerNative.cs NumericsProbe, tools/make_numerics_probe.py) ----
def _probe_models():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_numerics_probe", os.path.join(ROOT, "tools", "make_numerics_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.models()


def _observe(mod, lib, probes, transform_fused, transform_normal_fused):
    """What NumericsProbe.Observe() computes, with ONE oracle build (its Lerp and Dot) and one setting of its run-time Transform
    flags standing in for the running System.Numerics."""
    import struct
    f = lambda b: struct.unpack("<f", struct.pack("<I", b))[0]
    p = probes
    obs = {"lerp": mod.lerp(lib, f(p["lerp"]["a"]), f(p["lerp"]["b"]), f(p["lerp"]["t"])),
           "transform": mod.transform_x(lib, [f(x) for x in p["transform"]["v"]], [f(x) for x in p["transform"]["column"]], transform_fused),
           "transform_normal": mod.transform_normal_x(lib, [f(x) for x in p["transform_normal"]["n"]],
                                                      [f(x) for x in p["transform_normal"]["column"]], transform_normal_fused),
           "dot": mod.dot(lib, [f(x) for x in p["dot"]["a"]], [f(x) for x in p["dot"]["b"]]),
           "dot_zero": mod.dot(lib, [f(x) for x in p["dot_zero"]["a"]], [f(x) for x in p["dot_zero"]["b"]])}
    d = lib.oswr_numerics_fma(); lib.oswr_set_transform_fma(d, d)          # back to the library's default
    return obs


def _select(obs, p):
    """NumericsProbe.Observe() + SelectLibrary() + Configure(): the decision tree of the C# file, restated.  Nothing here can
    refuse a combination: each of the three fused-or-not answers is taken on its own."""
    lerp = 1 if obs["lerp"] == p["lerp"]["fused"] else 0 if obs["lerp"] == p["lerp"]["unfused"] else -1
    tr = 1 if obs["transform"] == p["transform"]["fused"] else 0 if obs["transform"] == p["transform"]["unfused"] else -1
    tn = 1 if obs["transform_normal"] == p["transform_normal"]["fused"] else 0 if obs["transform_normal"] == p["transform_normal"]["unfused"] else -1
    assert lerp >= 0 and tr >= 0 and tn >= 0
    if obs["dot"] == p["dot"]["shuffle"]:
        order = 2
    else:
        assert obs["dot"] == p["dot"]["sequential"]
        order = 1 if obs["dot_zero"] == p["dot_zero"]["dpps"] else 0
        assert order == 1 or obs["dot_zero"] == p["dot_zero"]["sequential"]
    return lerp, tr, tn, order, "libswr_hip" + ("_fma" if lerp else "") + ("_dotpw" if order == 2 else "_dpps" if order == 1 else "") + ".so"


def test_numerics_probe_table_selects_the_matching_build_for_every_model():
    """All 2 x 2 x 2 x 3 = 24 System.Numerics models (Transform, TransformNormal, Lerp fused or not; three dot orders) are
    observed correctly and served: (Lerp, Dot) picks one of six existing libraries, the Transform pair becomes run-time flags."""
    import json
    mod, models = _probe_models()
    table = json.load(open(os.path.join(ROOT, "csharp", "numerics_probe.json")))
    p = table["probes"]
    # the committed operands still separate the models (the expected bits are what the oracle builds compute today)
    for k in ("lerp", "transform", "transform_normal"):
        assert p[k]["fused"] != p[k]["unfused"], k
    assert p["dot"]["sequential"] != p["dot"]["shuffle"] and p["dot_zero"]["sequential"] != p["dot_zero"]["dpps"]
    libs = {(e["fma"], e["dot"]): e["library"] for e in table["libraries"]}
    assert set(libs) == set(models) and len(libs) == 6                     # six builds: every (Lerp, Dot) pair
    served = 0
    for (fma, dot), olib in models.items():
        path = os.path.join(ROOT, "softwarerenderer_amd", libs[(fma, dot)])
        assert os.path.exists(path), f"{libs[(fma, dot)]} is missing: make -C softwarerenderer_amd/csrc variants"
        hip = ctypes.CDLL(path)                                            # loads without a GPU; the query touches no device
        a, b = ctypes.c_int(-1), ctypes.c_int(-1)
        assert hip.swr_numerics_mode(ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (fma, dot)
        for tr in (0, 1):
            for tn in (0, 1):
                sel = _select(_observe(mod, olib, p, tr, tn), p)
                assert sel == (fma, tr, tn, dot, libs[(fma, dot)]), (fma, tr, tn, dot, sel)
                served += 1
    assert served == 24
    cs = open(os.path.join(ROOT, "csharp", "RasterizerNative.cs")).read()
    probe_cs = cs[cs.index("public static class NumericsProbe"):cs.index("public static class SwrContext")]
    assert "lerpFused != trFused" not in probe_cs and "NumericsProbe.Configure(c);" in cs       # no combination is refused any more
    assert probe_cs.count("throw new NotSupportedException") == 2                                 # only bit patterns that are neither model's


def test_csharp_probe_constants_are_the_generated_ones():
    import json
    p = json.load(open(os.path.join(ROOT, "csharp", "numerics_probe.json")))["probes"]
    cs = open(os.path.join(ROOT, "csharp", "RasterizerNative.cs")).read()
    block = re.search(r"// <generated by tools/make_numerics_probe\.py.*?// </generated>", cs, flags=re.S).group(0)
    consts = {k: int(v, 16) for k, v in re.findall(r"(\w+) = (0x[0-9A-F]{8})u", block)}
    arrays = {k: [int(x, 16) for x in re.findall(r"0x[0-9A-F]{8}", v)] for k, v in re.findall(r"(\w+) = \{([^}]*)\}", block)}
    assert (consts["LerpA"], consts["LerpB"], consts["LerpT"], consts["LerpUnfused"], consts["LerpFused"]) == \
        (p["lerp"]["a"], p["lerp"]["b"], p["lerp"]["t"], p["lerp"]["unfused"], p["lerp"]["fused"])
    assert arrays["TransformV"] == p["transform"]["v"] and arrays["TransformColumn"] == p["transform"]["column"]
    assert (consts["TransformUnfused"], consts["TransformFused"]) == (p["transform"]["unfused"], p["transform"]["fused"])
    assert arrays["TransformNormalN"] == p["transform_normal"]["n"] and arrays["TransformNormalColumn"] == p["transform_normal"]["column"]
    assert (consts["TransformNormalUnfused"], consts["TransformNormalFused"]) == (p["transform_normal"]["unfused"], p["transform_normal"]["fused"])
    assert arrays["DotA"] == p["dot"]["a"] and arrays["DotB"] == p["dot"]["b"]
    assert (consts["DotSequential"], consts["DotShuffle"]) == (p["dot"]["sequential"], p["dot"]["shuffle"])
    assert arrays["DotZeroA"] == p["dot_zero"]["a"] and arrays["DotZeroB"] == p["dot_zero"]["b"]
    assert (consts["DotZeroSequential"], consts["DotZeroDpps"]) == (p["dot_zero"]["sequential"], p["dot_zero"]["dpps"])
    # the probe runs before the first P/Invoke, and the resolver maps the DllImport name to the selected file
    assert "NumericsProbe.Install();" in cs and 'SetDllImportResolver' in cs and 'name == "swr_hip"' in cs
    # regenerating gives the committed table (the search is seeded)
    mod, _ = _probe_models()
    assert mod.search() == p
