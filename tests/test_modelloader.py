"""Row N4: asset ingestion (softwarerenderer_amd/modelloader.py) against hand-built glTF / OBJ files.
The expected arrays are computed here independently from ModelLoader.cs's rules (file:line in the loader);
Assimp itself is unavailable, so what it would add (normal generation) is unpinned and not asserted."""
import base64
import json
import os
import struct

import numpy as np
import pytest

from softwarerenderer_amd.modelloader import Model, MAX_VERTS

F = np.float32


def _write_gltf(tmp_path, name, pos, nrm, uv, idx, nodes, idx_type=5123, embed=False, stride_pad=0):
    pos, nrm, uv = (np.asarray(a, dtype=F) for a in (pos, nrm, uv))
    idt = {5121: np.uint8, 5123: np.uint16, 5125: np.uint32}[idx_type]
    # interleave position+normal in ONE strided buffer view to exercise byteStride; uv and indices tightly packed
    inter = np.concatenate([pos, nrm, np.zeros((pos.shape[0], stride_pad), dtype=F)], axis=1).astype(F)
    stride = inter.shape[1] * 4
    blob_inter = inter.tobytes()
    blob_uv = uv.tobytes()
    blob_idx = np.asarray(idx, dtype=idt).tobytes()
    pad = (-len(blob_idx)) % 4
    blob = blob_inter + blob_uv + blob_idx + b"\0" * pad
    n = pos.shape[0]
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes,
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "materials": [{"name": "m0", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
        "textures": [{"source": 0}], "images": [{"uri": "tex/albedo.png"}],
        "accessors": [
            {"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": n, "type": "VEC3"},
            {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": n, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5126, "count": n, "type": "VEC2"},
            {"bufferView": 2, "componentType": idx_type, "count": len(idx), "type": "SCALAR"}],
        "bufferViews": [
            {"buffer": 0, "byteOffset": 0, "byteLength": len(blob_inter), "byteStride": stride},
            {"buffer": 0, "byteOffset": len(blob_inter), "byteLength": len(blob_uv)},
            {"buffer": 0, "byteOffset": len(blob_inter) + len(blob_uv), "byteLength": len(blob_idx)}],
    }
    if embed:
        doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    else:
        doc["buffers"] = [{"byteLength": len(blob), "uri": name + ".bin"}]
        (tmp_path / (name + ".bin")).write_bytes(blob)
    p = tmp_path / (name + ".gltf")
    p.write_text(json.dumps(doc))
    return str(p)


def _ref_transform(p, m):
    """Vector3.Transform in float32, one vertex at a time (independent of the loader's vectorised form)."""
    out = np.zeros((len(p), 3), dtype=F)
    for i, (x, y, z) in enumerate(np.asarray(p, dtype=F)):
        for c in range(3):
            out[i, c] = F(F(F(F(x * m[0, c]) + F(y * m[1, c])) + F(z * m[2, c])) + m[3, c])
    return out


QUAD_POS = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 0]]     # vertices 4,5 duplicate 1,0
QUAD_NRM = [[0, 0, 1]] * 6
QUAD_UV = [[0, 0], [1, 0], [1, 1], [0, 1], [1, 0], [0, 0]]
QUAD_IDX = [0, 1, 2, 5, 2, 3, 4, 2, 0]


def test_gltf_matrix_node_dedup_flipuv_and_material(tmp_path):
    Model._model_cache.clear()
    m = np.array([[2, 0, 0, 0], [0, 0, 3, 0], [0, -1, 0, 0], [5, 6, 7, 1]], dtype=F)   # row-vector matrix
    path = _write_gltf(tmp_path, "a", QUAD_POS, QUAD_NRM, QUAD_UV, QUAD_IDX,
                       nodes=[{"mesh": 0, "matrix": [float(v) for v in m.reshape(-1)]}], stride_pad=2)
    model = Model().LoadModel(path)
    assert len(model.Meshes) == 1 and model.Lights == [] and model.AnimationFrames == []
    mesh = model.Meshes[0]
    assert mesh.Indices.dtype == np.uint16
    # first-seen order of distinct (pos, normal, uv): 0,1,2,3 ; 5->0 ; 4->1
    np.testing.assert_array_equal(mesh.Indices, [0, 1, 2, 0, 2, 3, 1, 2, 0])
    assert mesh.Vertices.shape == (4,)
    np.testing.assert_array_equal(mesh.Vertices["position"], _ref_transform(QUAD_POS[:4], m))
    # normal (0,0,1) through the upper 3x3 = row 3 = (0,-1,0), normalised
    np.testing.assert_array_equal(mesh.Vertices["normal"], np.tile(np.array([0, -1, 0], dtype=F), (4, 1)))
    np.testing.assert_array_equal(mesh.Vertices["uv"], np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=F))   # v -> 1 - v
    np.testing.assert_array_equal(mesh.Vertices["color"], np.ones((4, 4), dtype=F))
    np.testing.assert_array_equal(mesh.BaseVertices, mesh.Vertices)
    assert mesh.Material["TexturePaths"]["Diffuse"] == os.path.join(str(tmp_path), "tex/albedo.png")
    assert mesh.ModelRootPath == str(tmp_path)


def test_gltf_trs_hierarchy_is_node_times_parent(tmp_path):
    Model._model_cache.clear()
    s = float(np.sqrt(0.5))
    nodes = [{"translation": [10, 0, 0], "children": [1]},
             {"mesh": 0, "scale": [2, 2, 2], "rotation": [0, 0, s, s], "translation": [0, 1, 0]}]   # +90 deg about z
    path = _write_gltf(tmp_path, "b", QUAD_POS[:3], QUAD_NRM[:3], QUAD_UV[:3], [0, 1, 2], nodes, idx_type=5125, embed=True)
    mesh = Model().LoadModel(path).Meshes[0]
    # p' = R(S p) + t_child + t_parent ; R(+90 z): (x,y) -> (-y,x)
    expect = np.array([[10, 1, 0], [10, 3, 0], [8, 3, 0]], dtype=F)
    np.testing.assert_allclose(mesh.Vertices["position"], expect, rtol=0, atol=2e-6)
    np.testing.assert_allclose(mesh.Vertices["normal"], np.tile([0, 0, 1], (3, 1)), atol=1e-6)
    np.testing.assert_array_equal(mesh.Indices, [0, 1, 2])


def test_gltf_u8_indices_and_no_indices(tmp_path):
    Model._model_cache.clear()
    path = _write_gltf(tmp_path, "c", QUAD_POS, QUAD_NRM, QUAD_UV, QUAD_IDX, nodes=[{"mesh": 0}], idx_type=5121)
    a = Model().LoadModel(path).Meshes[0]
    np.testing.assert_array_equal(a.Indices, [0, 1, 2, 0, 2, 3, 1, 2, 0])
    doc = json.loads(open(path).read())
    del doc["meshes"][0]["primitives"][0]["indices"]
    p2 = tmp_path / "c2.gltf"
    p2.write_text(json.dumps(doc))
    b = Model().LoadModel(str(p2)).Meshes[0]
    np.testing.assert_array_equal(b.Indices, [0, 1, 2, 3, 1, 0])          # 6 vertices, the last two duplicates
    assert b.Vertices.shape == (4,)


def test_split_instead_of_ushort_wrap(tmp_path):
    """> 65,535 distinct vertices: the reference wraps (ushort)vertices.Count silently; the loader starts a new mesh."""
    Model._model_cache.clear()
    n_tri = 30000                                                         # 90,000 distinct vertices
    pos = np.zeros((n_tri * 3, 3), dtype=F)
    pos[:, 0] = np.arange(n_tri * 3, dtype=F)
    pos[1::3, 1] = 1
    nrm = np.tile(np.array([0, 0, 1], dtype=F), (n_tri * 3, 1))
    uv = np.zeros((n_tri * 3, 2), dtype=F)
    path = _write_gltf(tmp_path, "big", pos, nrm, uv, np.arange(n_tri * 3), nodes=[{"mesh": 0}], idx_type=5125)
    meshes = Model().LoadModel(path).Meshes
    assert len(meshes) == 2
    assert meshes[0].Vertices.size == MAX_VERTS and meshes[0].Indices.size == MAX_VERTS
    assert meshes[1].Vertices.size == n_tri * 3 - MAX_VERTS
    assert int(meshes[0].Indices.max()) == MAX_VERTS - 1 and int(meshes[1].Indices.max()) == meshes[1].Vertices.size - 1
    back = np.concatenate([m.Vertices["position"][m.Indices] for m in meshes])
    np.testing.assert_array_equal(back, pos)


def test_obj_fan_triangulation_negative_indices_groups(tmp_path):
    Model._model_cache.clear()
    obj = """# quad + triangle
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
usemtl first
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl second
f -4/-4/-1 -3/-3/-1 -2/-2/-1
"""
    p = tmp_path / "q.obj"
    p.write_text(obj)
    meshes = Model().LoadModel(str(p)).Meshes
    assert [m.Material["Name"] for m in meshes] == ["first", "second"]
    np.testing.assert_array_equal(meshes[0].Indices, [0, 1, 2, 0, 2, 3])
    np.testing.assert_array_equal(meshes[0].Vertices["position"], np.array(QUAD_POS[:4], dtype=F))
    np.testing.assert_array_equal(meshes[0].Vertices["uv"], np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=F))
    np.testing.assert_array_equal(meshes[1].Indices, [0, 1, 2])
    np.testing.assert_array_equal(meshes[1].Vertices["position"], np.array(QUAD_POS[:3], dtype=F))


def test_directory_is_animation_frames_sorted_and_cached(tmp_path):
    Model._model_cache.clear()
    d = tmp_path / "anim"
    d.mkdir()
    for k, name in enumerate(["frame_002", "frame_000", "frame_001"]):
        pos = np.array(QUAD_POS[:3], dtype=F) + F(int(name[-1]))
        _write_gltf(d, name, pos, QUAD_NRM[:3], QUAD_UV[:3], [0, 1, 2], nodes=[{"mesh": 0}], embed=True)
    (d / "notes.txt").write_text("ignored")
    model = Model().LoadModel(str(d))
    assert len(model.AnimationFrames) == 3
    assert [float(f.Meshes[0].Vertices["position"][0, 0]) for f in model.AnimationFrames] == [0.0, 1.0, 2.0]
    assert model.Meshes is model.AnimationFrames[0].Meshes
    again = Model().LoadModel(str(d))
    assert again.Meshes is model.Meshes                                    # served from the model cache


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        Model().LoadModel(str(tmp_path / "missing.gltf"))
    p = tmp_path / "x.fbx"
    p.write_bytes(b"")
    with pytest.raises(ValueError):
        Model().LoadModel(str(p))


@pytest.mark.skipif(not os.path.isdir("/root/reference/OutputAssets/Assets"), reason="reference assets only exist in the build container")
def test_reference_assets_triangle_and_mesh_counts():
    """SURVEY.md section 5's counts for the shipped scenes (read as data; nothing is copied into the repo)."""
    Model._model_cache.clear()
    expect = {"dust2": (11, 9061), "Gun": (5, 3910), "gordon_freeman": (2, 639)}
    for name, (n_mesh, n_tri) in expect.items():
        m = Model().LoadModel(f"/root/reference/OutputAssets/Assets/{name}/scene.gltf")
        assert len(m.Meshes) == n_mesh
        assert sum(x.Indices.size // 3 for x in m.Meshes) == n_tri
