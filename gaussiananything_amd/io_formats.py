"""On-disk hand-off formats of the cascaded sampling scripts, host side only.

  * stage 1 -> stage 2: the sampled point cloud as a PLY vertex list -- the reference writes it with
    ``pcu.save_mesh_v(path, xyz)`` (/root/reference/nsr/lsgm/flow_matching_trainer.py:1208, 1744-1753) and the stage-2 script
    reads it back with ``pcu.load_mesh_v(path)`` and clips to +-0.45 (:1079).  point_cloud_utils is not in the reference
    tree; this module writes / reads the standard PLY such a vertex-only mesh is (binary little-endian, float32 x y z) and
    also accepts ASCII files.
  * decoded surfels: ``[1, N, 13]`` float32 ``.npy`` (:1475).
  * what the engine leaves for viewers: coloured point clouds as binary glTF + PLY (:1452-1464, :1742-1753), meshes as GLB / OBJ
    (utils/mesh_util.py:113-136), the surfels as a 2DGS-style PLY (nsr/gs_surfel.py:206-265).

``gaussiananything_amd.cascade`` keeps the tensors on the device; these functions reproduce the file-based hand-off for
interoperability with the reference scripts.
"""
from __future__ import annotations

import json
import os
import struct

import numpy as np

XYZ_CLIP = 0.45


def save_points_ply(path, xyz):
    """xyz: [N, 3] array-like -> binary little-endian PLY with float32 x, y, z."""
    v = np.ascontiguousarray(np.asarray(xyz, dtype="<f4").reshape(-1, 3))
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def load_points_ply(path):
    """Vertex positions of a PLY file as float32 [N, 3] (binary little / big endian or ascii; x, y, z may be float or double
    and need not be the first properties; other elements are ignored as long as ``vertex`` comes first)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    if lines[0].strip() != "ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt = next(l.split()[1] for l in lines if l.startswith("format"))
    n, props, in_vertex = 0, [], False
    for l in lines:
        t = l.split()
        if t[:1] == ["element"]:
            if in_vertex:
                break
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n = int(t[2])
        elif t[:1] == ["property"] and in_vertex:
            if t[1] == "list":
                raise ValueError(f"{path}: list property on vertices is not supported")
            props.append((t[2], t[1]))
    np_types = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
                "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
                "uint": "u4", "uint32": "u4"}
    names = [p[0] for p in props]
    if not {"x", "y", "z"} <= set(names):
        raise ValueError(f"{path}: vertex element has no x / y / z")
    if fmt == "ascii":
        rows = np.loadtxt(data[end:].decode("ascii").splitlines()[:n], dtype=np.float64, ndmin=2)
        return np.stack([rows[:, names.index(k)] for k in "xyz"], 1).astype(np.float32)
    order = "<" if fmt == "binary_little_endian" else ">"
    dt = np.dtype([(nm, order + np_types[ty]) for nm, ty in props])
    rec = np.frombuffer(data, dtype=dt, count=n, offset=end)
    return np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float32)


def load_stage1_points(path):
    """What the stage-2 script does with the stage-1 file: load and clip to the scene box (flow_matching_trainer.py:1079)."""
    return np.clip(load_points_ply(path), -XYZ_CLIP, XYZ_CLIP)[None]


def save_gaussians_npy(path, gaussians):
    g = np.asarray(gaussians, dtype=np.float32)
    if g.ndim == 2:
        g = g[None]
    assert g.ndim == 3 and g.shape[-1] == 13, "expected [1, N, 13] surfel Gaussians"
    np.save(path, g)


def load_gaussians_npy(path):
    g = np.load(path)
    assert g.ndim == 3 and g.shape[-1] == 13
    return g.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------------
# Coloured point clouds and meshes as binary glTF / PLY / OBJ.
#
# The reference writes them with ``trimesh`` (``PointCloud(vtx, colors=...).export('*.glb' | '*.ply')`` at
# /root/reference/nsr/lsgm/flow_matching_trainer.py:1455-1464 and :1744-1748, ``Trimesh(...).export(fpath, 'glb' | 'obj')`` at
# /root/reference/utils/mesh_util.py:113-136).  trimesh is not in the reference tree and not in this image: the files below are
# the published formats themselves (glTF 2.0 binary container: 12-byte header, one JSON chunk, one BIN chunk; PLY 1.0), what any
# viewer -- gradio's model viewer, meshlab -- reads; "parity unpinned" against trimesh's own byte layout.
# ---------------------------------------------------------------------------------------------------------------------------
def rotation_matrix_x(theta_degrees):
    """flow_matching_trainer.py:67-75 (degrees)."""
    t = np.radians(theta_degrees)
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def rotation_matrix_y(theta):
    """flow_matching_trainer.py:93-107 (radians)."""
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def colors_to_rgba8(colors, n):
    """Float colours in [0, 1] (or uint8 0..255), 3 or 4 channels -> uint8 RGBA [n, 4]; floats are scaled by 255 and rounded, the
    way trimesh's ``to_rgba`` treats a float array, and clipped (surfel colours are not clamped by the decoder)."""
    c = np.asarray(colors)
    if c.ndim == 1:
        c = np.broadcast_to(c, (n, c.shape[0]))
    if c.shape[0] != n or c.shape[1] not in (3, 4):
        raise ValueError(f"colours must be [{n}, 3 or 4], got {tuple(c.shape)}")
    if c.dtype.kind == "f":
        c = np.clip(np.round(c.astype(np.float64) * 255.0), 0, 255).astype(np.uint8)
    else:
        c = c.astype(np.uint8)
    if c.shape[1] == 3:
        c = np.concatenate([c, np.full((n, 1), 255, np.uint8)], 1)
    return np.ascontiguousarray(c)


def _pad4(b, fill):
    return b + fill * ((-len(b)) % 4)


def write_glb(path, positions, colors=None, faces=None):
    """Binary glTF 2.0 with one mesh primitive: POINTS (mode 0) without ``faces``, TRIANGLES (mode 4, uint32 indices) with.
    POSITION float32 with the min / max the specification requires, COLOR_0 normalised uint8 RGBA."""
    v = np.ascontiguousarray(np.asarray(positions, dtype="<f4").reshape(-1, 3))
    n = v.shape[0]
    blobs, views, accessors, attributes = [], [], [], {}

    def add(data, target, accessor):
        off = sum(len(b) for b in blobs)
        raw = data.tobytes()
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(raw), "target": target})
        blobs.append(_pad4(raw, b"\x00"))
        accessor["bufferView"] = len(views) - 1
        accessors.append(accessor)
        return len(accessors) - 1

    pos = {"componentType": 5126, "count": n, "type": "VEC3"}
    if n:
        pos["min"], pos["max"] = [float(x) for x in v.min(0)], [float(x) for x in v.max(0)]
    attributes["POSITION"] = add(v, 34962, pos)
    if colors is not None:
        attributes["COLOR_0"] = add(colors_to_rgba8(colors, n), 34962,
                                    {"componentType": 5121, "count": n, "type": "VEC4", "normalized": True})
    prim = {"attributes": attributes, "mode": 0}
    if faces is not None:
        f = np.ascontiguousarray(np.asarray(faces, dtype="<u4").reshape(-1, 3))
        if f.size and int(f.max()) >= n:
            raise ValueError("face index beyond the vertex count")
        prim["indices"] = add(f.reshape(-1), 34963, {"componentType": 5125, "count": int(f.size), "type": "SCALAR"})
        prim["mode"] = 4
    binary = b"".join(blobs)
    doc = {"asset": {"version": "2.0", "generator": "gaussiananything_amd"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"mesh": 0}], "meshes": [{"primitives": [prim]}], "buffers": [{"byteLength": len(binary)}],
           "bufferViews": views, "accessors": accessors}
    js = _pad4(json.dumps(doc, separators=(",", ":")).encode("utf-8"), b" ")
    total = 12 + 8 + len(js) + 8 + len(binary)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<4sII", b"glTF", 2, total))
        fh.write(struct.pack("<I4s", len(js), b"JSON"))
        fh.write(js)
        fh.write(struct.pack("<I4s", len(binary), b"BIN\x00"))
        fh.write(binary)


def read_glb(path):
    """The first primitive of a .glb written by ``write_glb`` (or any glTF with embedded BIN chunk and tightly packed views):
    dict with ``positions`` [N,3] f32, ``colors`` [N,4] u8 or None, ``faces`` [F,3] u32 or None, ``mode``."""
    with open(path, "rb") as fh:
        data = fh.read()
    magic, version, total = struct.unpack_from("<4sII", data, 0)
    if magic != b"glTF" or version != 2 or total != len(data):
        raise ValueError(f"{path}: not a binary glTF 2.0 file")
    jl, jt = struct.unpack_from("<I4s", data, 12)
    if jt != b"JSON":
        raise ValueError(f"{path}: first chunk is not JSON")
    doc = json.loads(data[20:20 + jl].decode("utf-8"))
    bl, bt = struct.unpack_from("<I4s", data, 20 + jl)
    if bt != b"BIN\x00":
        raise ValueError(f"{path}: second chunk is not BIN")
    binary = data[28 + jl:28 + jl + bl]
    np_types = {5120: "i1", 5121: "u1", 5122: "<i2", 5123: "<u2", 5125: "<u4", 5126: "<f4"}
    width = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}

    def get(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        w = width[a["type"]]
        return np.frombuffer(binary, dtype=np_types[a["componentType"]], count=a["count"] * w, offset=off).reshape(a["count"], w)

    prim = doc["meshes"][0]["primitives"][0]
    att = prim["attributes"]
    return {"positions": get(att["POSITION"]).copy(),
            "colors": get(att["COLOR_0"]).copy() if "COLOR_0" in att else None,
            "faces": get(prim["indices"]).reshape(-1, 3).copy() if "indices" in prim else None,
            "mode": prim.get("mode", 4), "json": doc}


def save_colored_points_ply(path, xyz, colors):
    """Vertex-only PLY with float32 x y z and uchar red green blue alpha (what a coloured trimesh.PointCloud exports)."""
    v = np.asarray(xyz, dtype="<f4").reshape(-1, 3)
    rec = np.empty(v.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"),
                                      ("alpha", "u1")])
    rec["x"], rec["y"], rec["z"] = v[:, 0], v[:, 1], v[:, 2]
    c = colors_to_rgba8(colors, v.shape[0])
    rec["red"], rec["green"], rec["blue"], rec["alpha"] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


def load_colored_points_ply(path):
    """-> (xyz [N,3] f32, rgba [N,4] u8) of a file written by ``save_colored_points_ply``."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    n = next(int(l.split()[2]) for l in data[:end].decode("ascii").splitlines() if l.startswith("element vertex"))
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")])
    rec = np.frombuffer(data, dtype=dt, count=n, offset=end)
    return (np.stack([rec["x"], rec["y"], rec["z"]], 1).copy(),
            np.stack([rec["red"], rec["green"], rec["blue"], rec["alpha"]], 1).copy())


def export_gaussian_point_cloud(fine_gs, output_dir, name_prefix):
    """What the engine leaves beside a decoded sample (flow_matching_trainer.py:1452-1475): the surfel centres, turned for the
    viewer (``R_x(-90 deg)`` applied to the points, then ``@ R_y(pi).T``), coloured with the surfel rgb, as
    ``{prefix}-gaussian-pcd.glb`` (gradio) and ``{prefix}-gaussian-pcd.ply`` (meshlab), and the raw ``[1,N,13]`` array as
    ``{prefix}-gaussian.npy``.  ``fine_gs``: [1, N, 13] array or tensor.  Returns the three paths."""
    g = np.asarray(fine_gs.detach().cpu().float().numpy() if hasattr(fine_gs, "detach") else fine_gs, dtype=np.float32)
    if g.ndim != 3 or g.shape[-1] != 13:
        raise ValueError("expected [1, N, 13] surfel Gaussians")
    vtx = np.transpose(rotation_matrix_x(-90) @ np.transpose(g[0, :, :3]))
    vtx = vtx @ rotation_matrix_y(np.pi).T
    glb = f"{output_dir}/{name_prefix}-gaussian-pcd.glb"
    ply = f"{output_dir}/{name_prefix}-gaussian-pcd.ply"
    npy = f"{output_dir}/{name_prefix}-gaussian.npy"
    write_glb(glb, vtx, colors=g[0, :, 10:13])
    save_colored_points_ply(ply, vtx, g[0, :, 10:13])
    np.save(npy, g)
    return glb, ply, npy


def export_stage1_point_cloud(xyz, save_dir, name_prefix):
    """Stage-1 hand-off as the engine writes it (flow_matching_trainer.py:1742-1753): ``{prefix}.glb`` for display (points
    ``@ R_x(-90 deg).T``, colour 0.1 grey "since white background") and ``{prefix}.ply``, the un-rotated vertex list the stage-2
    script loads.  ``xyz``: [N, 3] un-normalised points.  Returns (glb path, ply path)."""
    v = np.asarray(xyz.detach().cpu().float().numpy() if hasattr(xyz, "detach") else xyz, dtype=np.float32).reshape(-1, 3)
    glb, ply = f"{save_dir}/{name_prefix}.glb", f"{save_dir}/{name_prefix}.ply"
    write_glb(glb, v @ rotation_matrix_x(-90).T, colors=np.ones_like(v) * 0.1)
    save_points_ply(ply, v)
    return glb, ply


def save_glb(pointnp_px3, facenp_fx3, colornp_px3, fpath):
    """utils/mesh_util.py:127-136: vertices mirrored in x and z, vertex colours, triangles as given."""
    p = np.asarray(pointnp_px3, dtype=np.float64) @ np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]])
    write_glb(fpath, p, colors=colornp_px3, faces=facenp_fx3)


def save_obj(pointnp_px3, facenp_fx3, colornp_px3, fpath):
    """utils/mesh_util.py:113-124: vertices mirrored in z, winding reversed, ``v x y z r g b`` lines (the vertex-colour extension
    trimesh writes), 1-based faces."""
    p = np.asarray(pointnp_px3, dtype=np.float64) @ np.array([[1, 0, 0], [0, 1, 0], [0, 0, -1]])
    f = np.asarray(facenp_fx3).reshape(-1, 3)[:, [2, 1, 0]].astype(np.int64) + 1
    c = colors_to_rgba8(colornp_px3, p.shape[0])[:, :3].astype(np.float64) / 255.0
    with open(fpath, "w") as fh:
        fh.write("".join("v %.8f %.8f %.8f %.5f %.5f %.5f\n" % (*a, *b) for a, b in zip(p.tolist(), c.tolist())))
        fh.write("".join("f %d %d %d\n" % tuple(t) for t in f.tolist()))


# ---- surfel Gaussians as a 2DGS-style PLY (nsr/gs_surfel.py:206-265) -------------------------------------------------------
SH_C0 = 0.28209479177387814
_2DGS_PLY_FIELDS = ("x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1",
                    "rot_0", "rot_1", "rot_2", "rot_3")


def save_2dgs_ply(path, gaussians, compatible=True):
    """``GaussianRenderer2DGS.save_2dgs_ply`` (/root/reference/nsr/gs_surfel.py:206-265).  The upstream body does not run (it
    concatenates ``xyz``, ``normals``, ``scale``, ``rotation``, none of which it defines, and lists ``f_dc_*`` twice); this is
    what it sets out to do: one float32 vertex per surfel with the 2D-Gaussian-splatting field order
    ``x y z nx ny nz f_dc_0..2 opacity scale_0 scale_1 rot_0..3`` (normals zero), and with ``compatible`` the activations
    inverted as upstream does (:221-225): opacity -> logit, scales -> log(s + 1e-8), colour -> (rgb - 0.5) / C0."""
    g = np.asarray(gaussians.detach().cpu().float().numpy() if hasattr(gaussians, "detach") else gaussians, dtype=np.float64)
    assert g.ndim == 3 and g.shape[-1] == 13, "expected [1, N, 13] surfel Gaussians"
    assert g.shape[0] == 1, "only support batch size 1"
    xyz, opacity, scales, rot, rgb = g[0, :, 0:3], g[0, :, 3:4], g[0, :, 4:6], g[0, :, 6:10], g[0, :, 10:13]
    if compatible:
        with np.errstate(divide="ignore"):
            opacity = np.log(opacity / (1.0 - opacity))  # kiui.op.inverse_sigmoid
        scales = np.log(scales + 1e-8)
        rgb = (rgb - 0.5) / SH_C0
    cols = np.concatenate([xyz, np.zeros_like(xyz), rgb, opacity, scales, rot], 1).astype("<f4")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {cols.shape[0]}\n" + \
        "".join(f"property float {k}\n" for k in _2DGS_PLY_FIELDS) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())


def load_2dgs_ply(path, compatible=True):
    """Inverse of ``save_2dgs_ply``: -> float32 ``[1, N, 13]`` (activations re-applied when ``compatible``)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    n = next(int(l.split()[2]) for l in lines if l.startswith("element vertex"))
    names = [l.split()[2] for l in lines if l.startswith("property")]
    rec = np.frombuffer(data, dtype="<f4", count=n * len(names), offset=end).reshape(n, len(names)).astype(np.float64)
    col = lambda *ks: np.stack([rec[:, names.index(k)] for k in ks], 1)
    xyz, rgb, opacity = col("x", "y", "z"), col("f_dc_0", "f_dc_1", "f_dc_2"), col("opacity")
    scales, rot = col("scale_0", "scale_1"), col("rot_0", "rot_1", "rot_2", "rot_3")
    if compatible:
        opacity = 1.0 / (1.0 + np.exp(-opacity))
        scales = np.exp(scales) - 1e-8
        rgb = rgb * SH_C0 + 0.5
    return np.concatenate([xyz, opacity, scales, rot, rgb], 1).astype(np.float32)[None]
