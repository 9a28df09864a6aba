#!/usr/bin/env python3
"""Exports tests/golden/mesh_cam_ref.npz.  Run ONCE in the build container (needs /root/reference):

    python tests/golden/make_mesh_golden.py

The reference's own ``to_cam_open3d_compat`` and ``post_process_mesh`` (/root/reference/utils/mesh_util.py:80-110, 22-44) are
imported and executed here.  The module imports third-party packages that are absent from this image (open3d, xatlas,
trimesh, cv2); they are replaced by RECORDING stand-ins that implement only what the two functions touch:
  * ``o3d.camera.PinholeCameraIntrinsic(width, height, cx, cy, fx, fy)`` / ``PinholeCameraParameters`` -- plain records, so
    that the intrinsics and the extrinsic the reference computes (its float32 tensor arithmetic) are captured exactly;
  * ``mesh.cluster_connected_triangles()`` -- Open3D's documented semantics (clusters of triangles joined through shared
    edges; returns per-triangle cluster index, triangles per cluster, area per cluster) computed with scipy; the SELECTION
    logic around it (keep the ten largest, none below 50 triangles, remove unreferenced vertices / degenerate triangles) is
    then the reference's own code running on a stand-in mesh class.
What is pinned: gaussiananything_amd.mesh.to_cam_open3d_compat (both of its branches) against the reference's numbers, and
mesh.post_process_mesh's selection against the reference's code path.  Not pinned (Open3D absent): TSDF integration and the
triangle extraction -- oracle/tsdf.py."""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class _Intrinsic:
    def __init__(self, width, height, cx, cy, fx, fy):
        self.width, self.height, self.cx, self.cy, self.fx, self.fy = width, height, cx, cy, fx, fy


class _Params:
    extrinsic = None
    intrinsic = None


class _Vec(list):
    pass


class _Mesh:
    """what post_process_mesh touches of o3d.geometry.TriangleMesh"""

    def __init__(self, vertices, triangles):
        self.vertices = np.asarray(vertices, np.float64)
        self.triangles = np.asarray(triangles, np.int64)

    def __deepcopy__(self, memo):
        return _Mesh(self.vertices.copy(), self.triangles.copy())

    def cluster_connected_triangles(self):
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        t = self.triangles
        e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
        owner = np.tile(np.arange(len(t)), 3)
        key = e[:, 0] * (int(t.max()) + 1) + e[:, 1]
        o = np.argsort(key, kind="stable")
        ks, ow = key[o], owner[o]
        same = ks[1:] == ks[:-1]
        g = coo_matrix((np.ones(int(same.sum())), (ow[:-1][same], ow[1:][same])), shape=(len(t), len(t)))
        _, lab = connected_components(g, directed=False)
        p = self.vertices
        area = 0.5 * np.linalg.norm(np.cross(p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]]), axis=1)
        return lab, np.bincount(lab), np.bincount(lab, weights=area)

    def remove_triangles_by_mask(self, mask):
        self.triangles = self.triangles[~np.asarray(mask)]

    def remove_unreferenced_vertices(self):
        used = np.unique(self.triangles)
        remap = np.full(len(self.vertices), -1, np.int64)
        remap[used] = np.arange(len(used))
        self.vertices, self.triangles = self.vertices[used], remap[self.triangles]

    def remove_degenerate_triangles(self):
        t = self.triangles
        self.triangles = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]


class _Ctx:
    def __init__(self, *a):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _stand_ins():
    o3d = types.ModuleType("open3d")
    o3d.camera = types.SimpleNamespace(PinholeCameraIntrinsic=_Intrinsic, PinholeCameraParameters=_Params)
    o3d.utility = types.SimpleNamespace(VerbosityContextManager=_Ctx, VerbosityLevel=types.SimpleNamespace(Debug=0))
    mods = {"open3d": o3d}
    for name in ("xatlas", "trimesh", "cv2"):
        mods[name] = types.ModuleType(name)
    return mods


def load_ref_mesh_util():
    saved = {k: sys.modules.get(k) for k in ("open3d", "xatlas", "trimesh", "cv2")}
    sys.modules.update(_stand_ins())
    try:
        spec = importlib.util.spec_from_file_location("ref_mesh_util", os.path.join(REF, "utils/mesh_util.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def main():
    mu = load_ref_mesh_util()
    spec = importlib.util.spec_from_file_location("ref_graphics_utils", os.path.join(REF, "utils/gs_utils/graphics_utils.py"))
    gu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gu)
    cams = np.load(os.path.join(HERE, "cameras_eval8.npz"))
    poses = cams["poses"]
    intr, ext = [], []
    for i in range(len(poses)):
        fov = gu.focal2fov(poses[i][16], 1)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fov, fovY=fov).transpose(0, 1)
        c = {"projection_matrix": proj, "cam_view": torch.from_numpy(cams["cam_view"][i])}
        cam = mu.to_cam_open3d_compat(c)
        intr.append([cam.intrinsic.fx, cam.intrinsic.fy, cam.intrinsic.cx, cam.intrinsic.cy, cam.intrinsic.width, cam.intrinsic.height])
        ext.append(np.asarray(cam.extrinsic, np.float64))
    # post_process_mesh on a synthetic mesh of 14 clusters of known sizes (a strip of k triangles each)
    rng = np.random.default_rng(5)
    sizes = [400, 300, 260, 200, 150, 120, 90, 80, 70, 60, 55, 52, 40, 10]
    verts, tris = [], []
    for s_ in sizes:
        base = len(verts)
        off = rng.uniform(-1, 1, 3) * 5
        for k in range(s_ + 2):
            verts.append(off + np.array([k // 2 * 0.1, (k % 2) * 0.1, 0.0]))
        for k in range(s_):
            tris.append([base + k, base + k + 1, base + k + 2])
    verts.append(np.array([9.0, 9.0, 9.0]))           # an unreferenced vertex
    tris.append([0, 0, 1])                            # a degenerate triangle inside the largest cluster
    verts, tris = np.array(verts), np.array(tris)
    perm = rng.permutation(len(tris))
    tris = tris[perm]
    out = mu.post_process_mesh(_Mesh(verts, tris))
    np.savez(os.path.join(HERE, "mesh_cam_ref.npz"), intrinsics=np.array(intr, np.float64), extrinsics=np.stack(ext),
             pp_vertices=verts.astype(np.float32), pp_triangles=tris.astype(np.int32),
             pp_out_vertices=out.vertices.astype(np.float32), pp_out_triangles=out.triangles.astype(np.int32))
    print("intrinsics[0]", intr[0], "post-processed", out.triangles.shape, "of", tris.shape)


if __name__ == "__main__":
    main()
