"""On-disk hand-off formats of the cascaded sampling scripts, host side only.

  * stage 1 -> stage 2: the sampled point cloud as a PLY vertex list -- the reference writes it with
    ``pcu.save_mesh_v(path, xyz)`` (/root/reference/nsr/lsgm/flow_matching_trainer.py:1208, 1744-1753) and the stage-2 script
    reads it back with ``pcu.load_mesh_v(path)`` and clips to +-0.45 (:1079).  point_cloud_utils is not in the reference
    tree; this module writes / reads the standard PLY such a vertex-only mesh is (binary little-endian, float32 x y z) and
    also accepts ASCII files.
  * decoded surfels: ``[1, N, 13]`` float32 ``.npy`` (:1475).

``gaussiananything_amd.cascade`` keeps the tensors on the device; these functions reproduce the file-based hand-off for
interoperability with the reference scripts.
"""
from __future__ import annotations

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
