"""Asset ingestion without Assimp (row N4): glTF 2.0 (+ .bin / data: URIs) and Wavefront OBJ -> the reference's
`Mesh` shape (Shaders.VertexInput[] + ushort[] indices), mirroring `Model` / `Mesh` of ModelLoader.cs.

Host-side: it only PRODUCES inputs of the hot path.  What is reproduced from ModelLoader.cs (file:line):
  * node walk with globalTransform = nodeTransform * parentTransform, row-vector convention   :159-162,296-299
  * positions through Vector3.Transform(pos, global); normals through Normalize(TransformNormal(n, upper 3x3))   :188-191
  * vertex de-duplication on (position, normal, uv) in first-seen order, 16-bit indices        :165-218
  * PostProcessSteps.FlipUVs (v -> 1 - v), white default vertex colour, diffuse texture path  :145-150,184-186,240-270
  * a directory of model files = animation frames sorted by name                              :79-115
Where Assimp's own numerics matter (normal generation when a mesh has none, TRS composition) this loader is
build-defined -- AssimpNet 5.0.0-beta1 is not available, so those results are unpinned.
One deliberate difference: the reference wraps its ushort index silently past 65,535 unique vertices
(ModelLoader.cs:206 `(ushort)vertices.Count`); this loader starts a new Mesh instead, so every mesh is valid.
"""
from __future__ import annotations

import base64
import json
import os
from typing import Dict, List, Optional

import numpy as np

from .rasterizer import VERTEX_DTYPE

F = np.float32
MAX_VERTS = 65535


class Mesh:
    """public class Mesh (ModelLoader.cs:43-60)."""

    def __init__(self, vertices: np.ndarray, indices: np.ndarray, material: Optional[dict] = None, root: str = ""):
        self.Vertices = vertices
        self.BaseVertices = vertices.copy()
        self.Indices = indices
        self.Material = material or {}
        self.ModelRootPath = root
        self._gpu = None

    def Upload(self, device):
        """Retained GPU mesh (softwarerenderer_amd.Mesh); its SphereBounds is FrustumCuller.CalculateBoundingSphere on the GPU."""
        from .rasterizer import Mesh as GpuMesh
        if self._gpu is None:
            self._gpu = GpuMesh(device, self.Vertices, self.Indices)
        return self._gpu


def _transform_points(p: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Vector3.Transform(p, M): ((x*row1 + y*row2) + z*row3) + row4, float32, row-vector convention."""
    x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
    r = x * m[0:1, :3]
    r = r + y * m[1:2, :3]
    r = r + z * m[2:3, :3]
    return (r + m[3:4, :3]).astype(F)


def _transform_normals(n: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Vector3.Normalize(Vector3.TransformNormal(n, rotationOnly)) (ModelLoader.cs:189-191)."""
    x, y, z = n[:, 0:1], n[:, 1:2], n[:, 2:3]
    r = x * m[0:1, :3]
    r = r + y * m[1:2, :3]
    r = (r + z * m[2:3, :3]).astype(F)
    ln = np.sqrt(((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2]).astype(F)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (r / ln[:, None]).astype(F)


def _quat_to_matrix(q) -> np.ndarray:
    """Matrix4x4.CreateFromQuaternion (row-vector convention), q = (x, y, z, w)."""
    x, y, z, w = (F(v) for v in q)
    xx, yy, zz = x * x, y * y, z * z
    xy, wz, xz, wy, yz, wx = x * y, z * w, z * x, y * w, y * z, x * w
    m = np.eye(4, dtype=F)
    m[0, 0] = F(1) - F(2) * (yy + zz); m[0, 1] = F(2) * (xy + wz); m[0, 2] = F(2) * (xz - wy)
    m[1, 0] = F(2) * (xy - wz); m[1, 1] = F(1) - F(2) * (zz + xx); m[1, 2] = F(2) * (yz + wx)
    m[2, 0] = F(2) * (xz + wy); m[2, 1] = F(2) * (yz - wx); m[2, 2] = F(1) - F(2) * (yy + xx)
    return m


def _node_matrix(node: dict) -> np.ndarray:
    if "matrix" in node:
        # glTF stores column-major for column vectors; read row-major it IS the row-vector matrix
        return np.asarray(node["matrix"], dtype=F).reshape(4, 4)
    s = np.eye(4, dtype=F)
    if "scale" in node:
        s[0, 0], s[1, 1], s[2, 2] = (F(v) for v in node["scale"])
    r = _quat_to_matrix(node["rotation"]) if "rotation" in node else np.eye(4, dtype=F)
    t = np.eye(4, dtype=F)
    if "translation" in node:
        t[3, 0], t[3, 1], t[3, 2] = (F(v) for v in node["translation"])
    return ((s @ r).astype(F) @ t).astype(F)          # row vectors: scale, then rotate, then translate


def _face_normals(pos: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """Stand-in for Assimp's GenerateNormals when a mesh has none: area-weighted vertex normals (build-defined)."""
    n = np.zeros_like(pos, dtype=np.float64)
    a, b, c = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
    fn = np.cross(b - a, c - a)
    for k in range(3):
        np.add.at(n, tris[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (n / ln).astype(F)


def _build_meshes(pos, nrm, uv, col, tri_idx, material, root) -> List[Mesh]:
    """De-duplicate on (position, normal, uv) in first-seen order (ModelLoader.cs:165-218); split at 65,535 vertices."""
    meshes: List[Mesh] = []
    keys = np.concatenate([pos, nrm, uv], axis=1).astype(F)
    key_bytes = keys.view(np.uint8).reshape(keys.shape[0], -1)
    table: Dict[bytes, int] = {}
    verts: List[int] = []
    idx: List[int] = []

    def emit():
        if not idx:
            return
        v = np.zeros(len(verts), dtype=VERTEX_DTYPE)
        sel = np.asarray(verts, dtype=np.int64)
        v["position"], v["uv"], v["normal"], v["color"] = pos[sel], uv[sel], nrm[sel], col[sel]
        meshes.append(Mesh(v, np.asarray(idx, dtype=np.uint16), material, root))

    for tri in tri_idx.reshape(-1, 3):
        new = sum(1 for vi in tri if key_bytes[vi].tobytes() not in table)
        if len(verts) + new > MAX_VERTS:               # the reference would wrap here; start a fresh mesh instead
            emit()
            table, verts, idx = {}, [], []
        for vi in tri:
            kb = key_bytes[vi].tobytes()
            j = table.get(kb)
            if j is None:
                j = len(verts)
                table[kb] = j
                verts.append(int(vi))
            idx.append(j)
    emit()
    return meshes


# ------------------------------------------------------------------------------------------ glTF
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class _Gltf:
    def __init__(self, path: str):
        self.root = os.path.dirname(os.path.abspath(path))
        self.doc = json.load(open(path, "r"))
        self.buffers = []
        for b in self.doc.get("buffers", []):
            uri = b.get("uri", "")
            if uri.startswith("data:"):
                self.buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                self.buffers.append(open(os.path.join(self.root, uri), "rb").read())

    def accessor(self, i: int) -> np.ndarray:
        a = self.doc["accessors"][i]
        dt, nc = np.dtype(_COMP[a["componentType"]]), _NCOMP[a["type"]]
        bv = self.doc["bufferViews"][a["bufferView"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        buf = self.buffers[bv["buffer"]]
        raw = np.frombuffer(buf, dtype=np.uint8, count=(a["count"] - 1) * stride + dt.itemsize * nc, offset=off)
        rows = np.lib.stride_tricks.as_strided(raw, shape=(a["count"], dt.itemsize * nc), strides=(stride, 1))
        out = np.ascontiguousarray(rows).view(dt).reshape(a["count"], nc)
        if a.get("normalized") and dt.kind in "iu":
            out = out.astype(np.float32) / float(np.iinfo(dt).max)
        return out

    def material(self, i: Optional[int]) -> dict:
        if i is None:
            return {}
        m = self.doc.get("materials", [])[i]
        pbr = m.get("pbrMetallicRoughness", {})
        out = {"Name": m.get("name", ""), "BaseColor": tuple(pbr.get("baseColorFactor", (1, 1, 1, 1))),
               "Metallic": pbr.get("metallicFactor", 0.0), "Roughness": pbr.get("roughnessFactor", 0.5), "TexturePaths": {}}
        tex = pbr.get("baseColorTexture")
        if tex is not None:
            src = self.doc["textures"][tex["index"]].get("source")
            if src is not None and "uri" in self.doc["images"][src]:
                out["TexturePaths"]["Diffuse"] = os.path.join(self.root, self.doc["images"][src]["uri"])   # ModelLoader.cs:266-268
        return out


def _load_gltf(path: str) -> List[Mesh]:
    g = _Gltf(path)
    doc = g.doc
    meshes: List[Mesh] = []

    def process(node_i: int, parent: np.ndarray):
        node = doc["nodes"][node_i]
        glob = (_node_matrix(node) @ parent).astype(F)                         # ModelLoader.cs:162
        rot = np.eye(4, dtype=F)
        rot[:3, :3] = glob[:3, :3]                                             # :164-168
        if "mesh" in node:
            for prim in doc["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue                                                   # only triangle lists (face.IndexCount != 3 -> skipped, :178)
                att = prim["attributes"]
                pos = g.accessor(att["POSITION"]).astype(F)
                n = pos.shape[0]
                tri = g.accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(n, dtype=np.int64)
                tri = tri[: (tri.size // 3) * 3]
                nrm = g.accessor(att["NORMAL"]).astype(F) if "NORMAL" in att else _face_normals(pos, tri.reshape(-1, 3))
                uv = g.accessor(att["TEXCOORD_0"]).astype(F)[:, :2] if "TEXCOORD_0" in att else np.zeros((n, 2), dtype=F)
                uv = uv.copy()
                if "TEXCOORD_0" in att:
                    uv[:, 1] = F(1.0) - uv[:, 1]                               # PostProcessSteps.FlipUVs, :148
                col = np.ones((n, 4), dtype=F)
                if "COLOR_0" in att:
                    c = g.accessor(att["COLOR_0"]).astype(F)
                    col[:, :c.shape[1]] = c
                meshes.extend(_build_meshes(_transform_points(pos, glob), _transform_normals(nrm, rot), uv, col, tri,
                                            g.material(prim.get("material")), g.root))
        for ch in node.get("children", []):
            process(ch, glob)                                                  # :296-299

    scene = doc["scenes"][doc.get("scene", 0)]
    for ni in scene["nodes"]:
        process(ni, np.eye(4, dtype=F))
    return meshes


# ------------------------------------------------------------------------------------------ OBJ
def _load_obj(path: str) -> List[Mesh]:
    root = os.path.dirname(os.path.abspath(path))
    v, vt, vn = [], [], []
    groups: Dict[str, List[tuple]] = {}
    cur = "default"
    for line in open(path, "r"):
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            vt.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
        elif p[0] == "vn":
            vn.append([float(x) for x in p[1:4]])
        elif p[0] == "usemtl":
            cur = p[1] if len(p) > 1 else "default"
        elif p[0] == "f":
            corners = []
            for tok in p[1:]:
                a = (tok.split("/") + ["", ""])[:3]
                vi = int(a[0]); ti = int(a[1]) if a[1] else 0; ni = int(a[2]) if a[2] else 0
                corners.append((vi - 1 if vi > 0 else len(v) + vi, (ti - 1 if ti > 0 else len(vt) + ti) if ti else -1,
                                (ni - 1 if ni > 0 else len(vn) + ni) if ni else -1))
            for k in range(1, len(corners) - 1):                               # Triangulate: fan
                groups.setdefault(cur, []).extend([corners[0], corners[k], corners[k + 1]])
    V = np.asarray(v, dtype=F).reshape(-1, 3)
    VT = np.asarray(vt, dtype=F).reshape(-1, 2)
    VN = np.asarray(vn, dtype=F).reshape(-1, 3)
    meshes: List[Mesh] = []
    ident = np.eye(4, dtype=F)
    for name, corners in groups.items():
        c = np.asarray(corners, dtype=np.int64)
        pos = V[c[:, 0]]
        tri = np.arange(c.shape[0], dtype=np.int64)
        uv = np.zeros((c.shape[0], 2), dtype=F)
        has_t = c[:, 1] >= 0
        if has_t.any():
            uv[has_t] = VT[c[has_t, 1]]
            uv[has_t, 1] = F(1.0) - uv[has_t, 1]                               # FlipUVs
        if (c[:, 2] >= 0).all() and VN.size:
            nrm = VN[c[:, 2]]
        else:
            nrm = _face_normals(pos, tri.reshape(-1, 3))
        col = np.ones((c.shape[0], 4), dtype=F)
        meshes.extend(_build_meshes(_transform_points(pos, ident), _transform_normals(nrm, ident), uv, col, tri,
                                    {"Name": name, "TexturePaths": {}}, root))
    return meshes


# ------------------------------------------------------------------------------------------ Model
class Model:
    """public class Model (ModelLoader.cs:62-349): Meshes, Lights, AnimationFrames, LoadModel."""

    _model_cache: Dict[str, "Model"] = {}
    SUPPORTED = {".gltf", ".obj"}          # the reference lists .fbx .obj .dae .3ds .blend .gltf .glb (Assimp); here: no Assimp

    def __init__(self):
        self.Meshes: List[Mesh] = []
        self.Lights: list = []             # Light.cs is dead data for the renderer
        self.AnimationFrames: List["Model"] = []

    def LoadModel(self, file_path: str) -> "Model":
        path = os.path.abspath(file_path)
        cached = Model._model_cache.get(path)
        if cached is not None:                                                  # ModelLoader.cs:80-86,118-124
            self.Meshes, self.Lights, self.AnimationFrames = cached.Meshes, cached.Lights, cached.AnimationFrames
            return self
        if os.path.isdir(path):                                                 # a directory = animation frames, :79-115
            files = sorted(f for f in os.listdir(path) if os.path.splitext(f)[1].lower() in Model.SUPPORTED)
            self.AnimationFrames = [Model()._load_single(os.path.join(path, f)) for f in files]
            if self.AnimationFrames:
                self.Meshes, self.Lights = self.AnimationFrames[0].Meshes, self.AnimationFrames[0].Lights
        elif os.path.isfile(path):
            self._load_single(path)
        else:
            raise FileNotFoundError(f"Model path not found: {path}")            # :131-134
        Model._model_cache[path] = self
        return self

    def _load_single(self, path: str) -> "Model":
        ext = os.path.splitext(path)[1].lower()
        if ext == ".gltf":
            self.Meshes = _load_gltf(path)
        elif ext == ".obj":
            self.Meshes = _load_obj(path)
        else:
            raise ValueError(f"unsupported model format {ext} (this loader reads .gltf and .obj)")
        return self
