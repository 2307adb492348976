"""Model files -> triangle soup + materials + images (+ camera).

Plumbing for the benchmark configs.  The reference does this in
driver.c:510-728 on top of codin's obj.h / gltf.h / stb_image, none of which exist in the
reference tree; material defaults follow driver.c:549-568 (OBJ) and :628-639 (glTF) with the
glTF-spec defaults (metallic = roughness = 1) where the file omits a factor.
Image decoding uses PIL.
"""
import io
import json
import os
import struct

import numpy as np

from .scene import Material, build_scene
from .background import procedural_background


# ---------------------------------------------------------------------------------------
# cameras

def quat_to_matrix(q):
    x, y, z, w = [np.float32(v) for v in q]
    one, two = np.float32(1), np.float32(2)
    return np.array([
        [one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
        [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
        [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], np.float32)


def camera_from_trs(translation, rotation=(0, 0, 0, 1), scale=(1, 1, 1)):
    """matrix_4x4_translation_rotation_scale of driver.c:765 (quaternion x,y,z,w)."""
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = quat_to_matrix(rotation) * np.asarray(scale, np.float32)[None, :]
    m[:3, 3] = np.asarray(translation, np.float32)
    return m


def default_camera():
    """driver.c:765-767: T=(0,0,3), R=I, fov 70 degrees."""
    fov = np.float32(np.float32(70.0) / np.float32(360.0)) * np.float32(np.pi) * np.float32(2.0)
    return camera_from_trs((0, 0, 3)), float(fov)


# ---------------------------------------------------------------------------------------
# OBJ / MTL

def _parse_mtl(path):
    mats, order = {}, []
    cur = None
    if not os.path.exists(path):
        return mats, order
    base = os.path.dirname(path)
    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            k = p[0]
            if k == "newmtl":
                cur = {"name": " ".join(p[1:]), "Kd": (0.8, 0.8, 0.8), "Ke": (0.0, 0.0, 0.0), "pbr": False, "base": base}
                mats[cur["name"]] = cur
                order.append(cur["name"])
            elif cur is None:
                continue
            elif k in ("Kd", "Ke"):
                cur[k] = tuple(float(v) for v in p[1:4])
            elif k in ("Pr", "Pm", "Ps", "Pc", "Pcr", "aniso", "anisor"):
                cur[k] = float(p[1])
                cur["pbr"] = True
            elif k in ("map_Kd", "map_Ke", "map_Pr", "map_Pm", "map_Bump", "norm", "bump"):
                cur[k] = p[-1]
    return mats, order


def load_obj(path):
    """Returns dict(positions, normals, uvs, material_ids, materials, images, camera=None)."""
    base = os.path.dirname(path)
    v, vt, vn = [], [], []
    faces, face_mat = [], []
    mtl, mtl_order = {}, []
    cur_mat = None
    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            k = p[0]
            if k == "v":
                v.append((float(p[1]), float(p[2]), float(p[3])))
            elif k == "vt":
                vt.append((float(p[1]), float(p[2]) if len(p) > 2 else 0.0))
            elif k == "vn":
                vn.append((float(p[1]), float(p[2]), float(p[3])))
            elif k == "mtllib":
                m, o = _parse_mtl(os.path.join(base, " ".join(p[1:])))
                mtl.update(m)
                mtl_order += o
            elif k == "usemtl":
                cur_mat = " ".join(p[1:]) if len(p) > 1 else None
            elif k == "f":
                idx = []
                for tok in p[1:]:
                    a = tok.split("/")
                    vi = int(a[0])
                    ti = int(a[1]) if len(a) > 1 and a[1] else 0
                    ni = int(a[2]) if len(a) > 2 and a[2] else 0
                    idx.append((vi, ti, ni))
                for j in range(1, len(idx) - 1):          # fan triangulation
                    faces.append((idx[0], idx[j], idx[j + 1]))
                    face_mat.append(cur_mat)
    V = np.asarray(v, np.float32).reshape(-1, 3)
    VT = np.asarray(vt, np.float32).reshape(-1, 2)
    VN = np.asarray(vn, np.float32).reshape(-1, 3)
    n = len(faces)
    fi = np.asarray(faces, np.int64).reshape(n, 3, 3)

    def resolve(col, count):
        i = fi[:, :, col].copy()
        i = np.where(i < 0, i + count + 1, i)     # negative = relative to the end
        return i

    pi, ti, ni = resolve(0, len(V)), resolve(1, len(VT)), resolve(2, len(VN))
    positions = V[pi - 1]
    uvs = np.where((ti > 0)[..., None], VT[np.maximum(ti, 1) - 1] if len(VT) else np.zeros((n, 3, 2), np.float32), 0.0)
    e1 = positions[:, 1] - positions[:, 0]
    e2 = positions[:, 2] - positions[:, 0]
    fn = np.cross(e1, e2)
    fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
    fn3 = np.repeat(fn[:, None, :], 3, axis=1).astype(np.float32)
    normals = np.where((ni > 0)[..., None], VN[np.maximum(ni, 1) - 1] if len(VN) else fn3, fn3)

    # materials: one per MTL entry (driver.c:549-568); unknown / empty usemtl -> default (F10)
    names = list(mtl_order)
    materials, images = [], []

    def add_image(m, key):
        if key in m:
            from PIL import Image as PILImage
            p_ = os.path.join(m["base"], m[key])
            if os.path.exists(p_):
                images.append(np.asarray(PILImage.open(p_).convert("RGB"), np.uint8))
                return len(images) - 1
        return None

    for nm in names:
        m = mtl[nm]
        mat = Material(base_color=m["Kd"], emission=m["Ke"], roughness=0.5)
        mat.texture_albedo = add_image(m, "map_Kd")
        mat.texture_emission = add_image(m, "map_Ke")
        if m["pbr"]:
            mat.anisotropic_strength = m.get("aniso", 0.0)
            mat.metalness = m.get("Pm", 0.0)
            mat.roughness = m.get("Pr", 0.0)
            mat.sheen = m.get("Ps", 0.0)
            mat.texture_normal = add_image(m, "norm")
            mat.texture_metal_roughness = add_image(m, "map_Pm")
        materials.append(mat)
    default_id = None
    ids = np.zeros(n, np.int64)
    for i, nm in enumerate(face_mat):
        if nm in mtl:
            ids[i] = names.index(nm)
        else:
            if default_id is None:
                default_id = len(materials)
                materials.append(Material(base_color=(0.8, 0.8, 0.8), roughness=0.5, metalness=0.0))
            ids[i] = default_id
    return dict(positions=positions.astype(np.float32), normals=normals.astype(np.float32),
                uvs=uvs.astype(np.float32), material_ids=ids, materials=materials, images=images, camera=None)


# ---------------------------------------------------------------------------------------
# glTF 2.0 (.glb / .gltf)

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _read_gltf(path):
    data = open(path, "rb").read()
    base = os.path.dirname(path)
    if data[:4] == b"glTF":
        _, _, total = struct.unpack("<III", data[:12])
        off, js, bins = 12, None, []
        while off < total:
            clen, ctype = struct.unpack("<II", data[off:off + 8])
            chunk = data[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                js = json.loads(chunk)
            elif ctype == 0x004E4942:
                bins.append(chunk)
            off += 8 + clen
        buffers = []
        for i, b in enumerate(js.get("buffers", [])):
            buffers.append(bins[i] if "uri" not in b else open(os.path.join(base, b["uri"]), "rb").read())
        return js, buffers, base
    js = json.loads(data)
    buffers = [open(os.path.join(base, b["uri"]), "rb").read() for b in js.get("buffers", [])]
    return js, buffers, base


def _accessor(js, buffers, idx):
    a = js["accessors"][idx]
    bv = js["bufferViews"][a["bufferView"]]
    dt = np.dtype(_COMP[a["componentType"]]).newbyteorder("<")
    nc = _NCOMP[a["type"]]
    start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    buf = buffers[bv["buffer"]]
    count = a["count"]
    if stride == dt.itemsize * nc:
        arr = np.frombuffer(buf, dt, count * nc, start).reshape(count, nc)
    else:
        raw = np.frombuffer(buf, np.uint8, (count - 1) * stride + dt.itemsize * nc, start)
        arr = np.lib.stride_tricks.as_strided(raw, (count, dt.itemsize * nc), (stride, 1)).copy().view(dt).reshape(count, nc)
    return arr


def _node_local(node):
    if "matrix" in node:
        return np.asarray(node["matrix"], np.float32).reshape(4, 4).T      # JSON is column-major
    return camera_from_trs(node.get("translation", (0, 0, 0)), node.get("rotation", (0, 0, 0, 1)),
                           node.get("scale", (1, 1, 1)))


def load_gltf(path):
    from PIL import Image as PILImage
    js, buffers, base = _read_gltf(path)

    images = []
    for im in js.get("images", []):
        if "bufferView" in im:
            bv = js["bufferViews"][im["bufferView"]]
            raw = buffers[bv["buffer"]][bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            pil = PILImage.open(io.BytesIO(raw))
        else:
            pil = PILImage.open(os.path.join(base, im["uri"]))
        images.append(np.asarray(pil.convert("RGB"), np.uint8))

    def tex_image(ref):
        if ref is None:
            return None
        return js["textures"][ref["index"]].get("source")

    materials = []
    for m in js.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        sheen = m.get("extensions", {}).get("KHR_materials_sheen", {}).get("sheenColorFactor", (0, 0, 0))
        lum = float(np.float32(0.2126) * np.float32(sheen[0]) + np.float32(0.7152) * np.float32(sheen[1])
                    + np.float32(0.0722) * np.float32(sheen[2]))
        mat = Material(base_color=tuple(pbr.get("baseColorFactor", (1, 1, 1, 1))[:3]),
                       roughness=pbr.get("roughnessFactor", 1.0), metalness=pbr.get("metallicFactor", 1.0),
                       sheen=lum, emission=tuple(m.get("emissiveFactor", (0, 0, 0))))
        if "normalTexture" in m:
            mat.texture_normal = tex_image(m["normalTexture"])
            mat.normal_map_strength = m["normalTexture"].get("scale", 1.0)
        mat.texture_emission = tex_image(m.get("emissiveTexture"))
        mat.texture_albedo = tex_image(pbr.get("baseColorTexture"))
        mat.texture_metal_roughness = tex_image(pbr.get("metallicRoughnessTexture"))
        materials.append(mat)

    nodes = js.get("nodes", [])
    globals_ = [None] * len(nodes)

    def visit(i, parent):
        g = (parent @ _node_local(nodes[i])).astype(np.float32)
        globals_[i] = g
        for c in nodes[i].get("children", []):
            visit(c, g)

    scene_idx = js.get("scene", 0)
    roots = js["scenes"][scene_idx]["nodes"] if js.get("scenes") else range(len(nodes))
    for r in roots:
        visit(r, np.eye(4, dtype=np.float32))

    camera = None
    for i, nd in enumerate(nodes):                       # driver.c:599-612: first perspective camera node
        if "camera" in nd and globals_[i] is not None:
            cam = js["cameras"][nd["camera"]]
            if cam.get("type") != "perspective":
                continue
            camera = (globals_[i], float(cam["perspective"]["yfov"]))
            break

    P, N, UV, IDS = [], [], [], []
    default_id = None
    for i, nd in enumerate(nodes):
        if "mesh" not in nd or globals_[i] is None:
            continue
        g = globals_[i]
        m3 = g[:3, :3].astype(np.float64)
        nmat = np.linalg.inv(m3).T
        for prim in js["meshes"][nd["mesh"]]["primitives"]:
            if prim.get("mode", 4) != 4:
                continue
            att = prim["attributes"]
            pos = _accessor(js, buffers, att["POSITION"]).astype(np.float32)
            nrm = _accessor(js, buffers, att["NORMAL"]).astype(np.float32) if "NORMAL" in att else None
            uv = _accessor(js, buffers, att["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in att else None
            idx = _accessor(js, buffers, prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim \
                else np.arange(len(pos), dtype=np.int64)
            idx = idx[:len(idx) // 3 * 3].reshape(-1, 3)
            wp = (pos.astype(np.float64) @ m3.T + g[:3, 3].astype(np.float64)).astype(np.float32)
            tp = wp[idx]
            if nrm is not None:
                wn = nrm.astype(np.float64) @ nmat.T
                wn = wn / np.maximum(np.linalg.norm(wn, axis=1, keepdims=True), 1e-30)
                tn = wn.astype(np.float32)[idx]
            else:
                fn = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0])
                fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
                tn = np.repeat(fn[:, None, :], 3, axis=1).astype(np.float32)
            tuv = uv[idx] if uv is not None else np.zeros((len(idx), 3, 2), np.float32)
            if "material" in prim:
                mid = prim["material"]
            else:
                if default_id is None:
                    default_id = len(materials)
                    materials.append(Material(base_color=(1, 1, 1), roughness=1.0, metalness=1.0))
                mid = default_id
            P.append(tp)
            N.append(tn)
            UV.append(tuv)
            IDS.append(np.full(len(idx), mid, np.int64))
    if not P:
        raise ValueError(f"{path}: no triangles")
    return dict(positions=np.concatenate(P), normals=np.concatenate(N), uvs=np.concatenate(UV),
                material_ids=np.concatenate(IDS), materials=materials, images=images, camera=camera)


# ---------------------------------------------------------------------------------------

def load_model_data(path):
    ext = os.path.splitext(path)[1].lower()          # driver.c:685-728
    if ext == ".obj":
        return load_obj(path)
    if ext in (".glb", ".gltf"):
        return load_gltf(path)
    raise ValueError(f"Unrecognized file type: '{path}'")


def load_model(path, camera=None, background=None, shader="disney", builder="reference"):
    """File -> HostScene.  camera = (4x4 matrix, yfov) overrides the file's / the default camera."""
    d = load_model_data(path)
    cam = camera or d["camera"] or default_camera()
    bg = background if background is not None else procedural_background()
    return build_scene(d["positions"], d["normals"], d["uvs"], d["material_ids"], d["materials"], d["images"],
                       cam[0], cam[1], bg, shader=shader, builder=builder)
