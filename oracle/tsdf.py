"""CPU oracle of the TSDF fusion + mesh extraction.  TEST INFRASTRUCTURE ONLY: imported by tests/
and nothing else; the product path is gaussiananything_amd/csrc/tsdf.hip behind include/ga_tsdf.h.

PARITY UNPINNED.  The reference calls Open3D (third party, `open3d`, no version pinned in
/root/reference/requirements.txt:33, not installed in this image, no golden meshes in the reference tree):
    /root/reference/nsr/lsgm/flow_matching_trainer.py:1344-1348  ScalableTSDFVolume(voxel_length, sdf_trunc, RGB8)
    :1371-1388  depth[alpha < alpha_thres] = 0; RGBDImage.create_from_color_and_depth(uint8 colour, float depth,
                depth_trunc, depth_scale=1, convert_rgb_to_intensity=False)
    :1390       volume.integrate(rgbd, intrinsic, extrinsic)
    :1392       volume.extract_triangle_mesh()
What follows restates Open3D's published algorithm (cpp/open3d/pipelines/integration/ScalableTSDFVolume.cpp and
UniformTSDFVolume.cpp, geometry/PointCloudFactory.cpp, geometry/RGBDImageFactory.cpp) from the description of its loops:
  * Image::ConvertDepthToFloatImage: depth /= depth_scale; depth >= depth_trunc -> 0;
  * ScalableTSDFVolume::Integrate: every depth_sampling_stride-th pixel with depth > 0 is un-projected in double precision,
    the 16^3-voxel units of the lattice floor(p / unit_length) between p - sdf_trunc and p + sdf_trunc are opened, and each
    unit opened by THIS frame is integrated once;
  * UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier: float arithmetic, the camera-space point of a (x, y)
    column advanced along z by repeated addition; u = x fx / z + cx + 0.5 truncated; sdf = (d - z) * sqrt(xx^2 + yy^2 + 1);
    sdf > -trunc: tsdf <- (tsdf w + min(1, sdf / trunc)) / (w + 1), colour likewise, w <- w + 1;
  * ExtractTriangleMesh: cubes whose 8 corners all have w != 0, corner "inside" when tsdf < 0; a vertex per intersected edge
    at |f0| / (|f0| + |f1|), colours ((|f1| c0 + |f0| c1) / (|f0| + |f1|)) / 255, shared between cubes.
Deviations, common to this oracle and the HIP path (and stated in include/ga_tsdf.h): the volume is dense over a box of
units -- units outside it are not opened; colours are accumulated in float32 (Open3D: double); the marching-cubes case table
is derived by rule (tools/gen_mc_table.py), since Open3D's 256-row table cannot be reproduced from memory: vertices are the
same set, the choice of diagonals inside a cube may differ."""
import json
import os

import numpy as np

UNIT = 16
_TABLE = None


def mc_table():
    global _TABLE
    if _TABLE is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_table.json")) as f:
            _TABLE = json.load(f)
    return _TABLE


class Volume:
    def __init__(self, units, unit0, voxel_length, sdf_trunc):
        self.units = tuple(int(u) for u in units)
        self.unit0 = tuple(int(u) for u in unit0)
        self.voxel_length = float(voxel_length)
        self.sdf_trunc = float(sdf_trunc)
        self.unit_length = self.voxel_length * UNIT
        r = tuple(UNIT * u for u in self.units)
        self.tsdf = np.zeros(r, np.float32)
        self.weight = np.zeros(r, np.float32)
        self.color = np.zeros((3,) + r, np.float32)
        self.allocated = np.zeros(self.units, bool)


def frame_depth(depth, alpha, alpha_thres, depth_trunc):
    d = np.array(depth, np.float32, copy=True)
    if alpha is not None:
        d[np.asarray(alpha, np.float32) < np.float32(alpha_thres)] = 0
    d[d >= np.float32(depth_trunc)] = 0
    return d


def integrate(vol: Volume, rgb, depth, alpha, alpha_thres, depth_trunc, intr, extrinsic, stride=4):
    """rgb [3,H,W] float, depth [H,W] float, intr = (fx, fy, cx, cy), extrinsic 4x4 world->camera (float64)."""
    f32 = np.float32
    H, W = depth.shape
    d = frame_depth(depth, alpha, alpha_thres, depth_trunc)
    rgb8 = (np.clip(np.asarray(rgb, np.float32), 0.0, 1.0) * f32(255.0)).astype(np.uint8).astype(np.float32)
    fx, fy, cx, cy = (float(v) for v in intr)
    ext = np.asarray(extrinsic, np.float64)
    pose = np.linalg.inv(ext)
    # --- open units
    touched = np.zeros(vol.units, bool)
    ii, jj = np.meshgrid(np.arange(0, H, stride), np.arange(0, W, stride), indexing="ij")
    ds = d[ii, jj]
    m = ds > 0
    z = ds[m].astype(np.float64)
    x = (jj[m].astype(np.float64) - cx) * z / fx
    y = (ii[m].astype(np.float64) - cy) * z / fy
    p = np.stack([((pose[r, 0] * x + pose[r, 1] * y) + pose[r, 2] * z) + pose[r, 3] for r in range(3)], 1)
    lo = np.floor((p - vol.sdf_trunc) / vol.unit_length).astype(np.int64) - np.array(vol.unit0)
    hi = np.floor((p + vol.sdf_trunc) / vol.unit_length).astype(np.int64) - np.array(vol.unit0)
    lo = np.maximum(lo, 0)
    hi = np.minimum(hi, np.array(vol.units) - 1)
    for a, b in {(tuple(l), tuple(h)) for l, h in zip(lo.tolist(), hi.tolist())}:
        if all(a[k] <= b[k] for k in range(3)):
            touched[a[0]:b[0] + 1, a[1]:b[1] + 1, a[2]:b[2] + 1] = True
    vol.allocated |= touched
    # --- integrate the opened units
    vl = f32(vol.voxel_length)
    half = f32(vl * f32(0.5))
    trunc = f32(vol.sdf_trunc)
    trunc_inv = f32(1.0) / trunc
    fxf, fyf, cxf, cyf = f32(fx), f32(fy), f32(cx), f32(cy)
    safe_w, safe_h = f32(W) - f32(0.0001), f32(H) - f32(0.0001)
    inv0, inv1 = f32(1.0) / fxf, f32(1.0) / fyf
    extf = ext.astype(np.float32)
    lx, ly = np.meshgrid(np.arange(UNIT), np.arange(UNIT), indexing="ij")
    for ux, uy, uz in zip(*np.nonzero(touched)):
        org = [(vol.unit0[k] + u) * vol.unit_length for k, u in enumerate((ux, uy, uz))]
        px = ((half + vl * lx.astype(np.float32)).astype(np.float64) + org[0]).astype(np.float32)
        py = ((half + vl * ly.astype(np.float32)).astype(np.float64) + org[1]).astype(np.float32)
        pz = f32(np.float64(half) + org[2])
        pc = [((extf[r, 0] * px + extf[r, 1] * py) + extf[r, 2] * pz) + extf[r, 3] for r in range(3)]
        step = [extf[r, 2] * vl for r in range(3)]
        sx, sy, sz = slice(ux * UNIT, (ux + 1) * UNIT), slice(uy * UNIT, (uy + 1) * UNIT), uz * UNIT
        for lz in range(UNIT):
            with np.errstate(divide="ignore", invalid="ignore"):
                u_f = pc[0] * fxf / pc[2] + cxf + f32(0.5)
                v_f = pc[1] * fyf / pc[2] + cyf + f32(0.5)
            ok = (pc[2] > 0) & (u_f >= f32(0.0001)) & (u_f < safe_w) & (v_f >= f32(0.0001)) & (v_f < safe_h)
            iu = np.where(ok, u_f, 0).astype(np.int32)
            iv = np.where(ok, v_f, 0).astype(np.int32)
            dd = d[iv, iu]
            ok &= dd > 0
            xx = (iu.astype(np.float32) - cxf) * inv0
            yy = (iv.astype(np.float32) - cyf) * inv1
            mult = np.sqrt(xx * xx + yy * yy + f32(1.0))
            sdf = (dd - pc[2]) * mult
            ok &= sdf > -trunc
            t = np.minimum(f32(1.0), sdf * trunc_inv)
            w = vol.weight[sx, sy, sz + lz]
            w1 = w + f32(1.0)
            cur = vol.tsdf[sx, sy, sz + lz]
            vol.tsdf[sx, sy, sz + lz] = np.where(ok, (cur * w + t) / w1, cur)
            for c in range(3):
                cc = vol.color[c, sx, sy, sz + lz]
                vol.color[c, sx, sy, sz + lz] = np.where(ok, (cc * w + rgb8[c][iv, iu]) / w1, cc)
            vol.weight[sx, sy, sz + lz] = np.where(ok, w1, w)
            pc = [pc[r] + step[r] for r in range(3)]
    return touched


def extract_mesh(vol: Volume):
    """-> vertices [nv,3] float32, colors [nv,3] float32, triangles [nt,3] int32; order: units lexicographic, within a unit
    voxels in the HIP path's storage order (z, x, y with y fastest), edges x, y, z."""
    tab = mc_table()
    tri = tab["triangles"]
    R = vol.tsdf.shape
    w, f = vol.weight, vol.tsdf
    inside = f < 0
    valid = np.ones(tuple(r - 1 for r in R), bool)
    case = np.zeros(tuple(r - 1 for r in R), np.int32)
    for i in range(8):
        o = (i & 1, (i >> 1) & 1, (i >> 2) & 1)
        sl = tuple(slice(o[k], R[k] - 1 + o[k]) for k in range(3))
        valid &= w[sl] != 0
        case |= inside[sl].astype(np.int32) << i
    cube = np.zeros(R, np.int32)
    cube[:-1, :-1, :-1] = np.where(valid & (case != 255), case, 0)

    def at(dx, dy, dz):   # cube[p - (dx, dy, dz)], 0 outside
        out = np.zeros(R, np.int32)
        out[dx:, dy:, dz:] = cube[:R[0] - dx, :R[1] - dy, :R[2] - dz]
        return out

    def differ(c, a, b):
        return ((c >> a) ^ (c >> b)) & 1

    c, cx, cy, cz = cube, at(1, 0, 0), at(0, 1, 0), at(0, 0, 1)
    cxy, cxz, cyz = at(1, 1, 0), at(1, 0, 1), at(0, 1, 1)
    flags = (differ(c, 0, 1) | differ(cy, 2, 3) | differ(cz, 4, 5) | differ(cyz, 6, 7)) \
        | ((differ(c, 0, 2) | differ(cx, 1, 3) | differ(cz, 4, 6) | differ(cxz, 5, 7)) << 1) \
        | ((differ(c, 0, 4) | differ(cx, 1, 5) | differ(cy, 2, 6) | differ(cxy, 3, 7)) << 2)
    vid = np.full(R, -1, np.int64)
    verts, cols, tris = [], [], []
    vl, half = vol.voxel_length, vol.voxel_length * 0.5
    order = [(lz, lx, ly) for lz in range(UNIT) for lx in range(UNIT) for ly in range(UNIT)]
    units = [(a, b, cc) for a in range(vol.units[0]) for b in range(vol.units[1]) for cc in range(vol.units[2])]
    nv = 0
    for ux, uy, uz in units:
        blk = flags[ux * UNIT:(ux + 1) * UNIT, uy * UNIT:(uy + 1) * UNIT, uz * UNIT:(uz + 1) * UNIT]
        if not blk.any():
            continue
        org = [(vol.unit0[k] + u) * vol.unit_length for k, u in enumerate((ux, uy, uz))]
        for lz, lx, ly in order:
            fl = int(blk[lx, ly, lz])
            if not fl:
                continue
            g = (ux * UNIT + lx, uy * UNIT + ly, uz * UNIT + lz)
            vid[g] = nv
            f0 = abs(float(f[g]))
            for axis in range(3):
                if not (fl >> axis) & 1:
                    continue
                g1 = tuple(g[k] + (k == axis) for k in range(3))
                f1 = abs(float(f[g1]))
                pt = [half + vl * lx, half + vl * ly, half + vl * lz]
                pt[axis] += f0 * vl / (f0 + f1)
                verts.append([np.float32(pt[k] + org[k]) for k in range(3)])
                cols.append([np.float32(((f1 * float(vol.color[k][g]) + f0 * float(vol.color[k][g1])) / (f0 + f1)) / 255.0) for k in range(3)])
                nv += 1
    for ux, uy, uz in units:
        blk = cube[ux * UNIT:(ux + 1) * UNIT, uy * UNIT:(uy + 1) * UNIT, uz * UNIT:(uz + 1) * UNIT]
        if not blk.any():
            continue
        for lz, lx, ly in order:
            cs = int(blk[lx, ly, lz])
            if cs == 0:
                continue
            g = (ux * UNIT + lx, uy * UNIT + ly, uz * UNIT + lz)
            row = [e for e in tri[cs] if e >= 0]
            ids = []
            for e in row:
                axis, a, b = e >> 2, e & 1, (e >> 1) & 1
                q = (g[0] + (0 if axis == 0 else a), g[1] + (a if axis == 0 else (0 if axis == 1 else b)), g[2] + (0 if axis == 2 else b))
                ids.append(int(vid[q]) + bin(int(flags[q]) & ((1 << axis) - 1)).count("1"))
            tris.extend(ids[k:k + 3] for k in range(0, len(ids), 3))
    return (np.array(verts, np.float32).reshape(-1, 3), np.array(cols, np.float32).reshape(-1, 3),
            np.array(tris, np.int32).reshape(-1, 3))


def post_process_mesh(vertices, colors, triangles):
    """utils/mesh_util.py:22-44: keep the (at most) 10 largest connected triangle clusters, none smaller than 50 triangles;
    drop unreferenced vertices and degenerate triangles.  (Open3D's cluster_connected_triangles joins triangles that share
    an edge.)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    t = np.asarray(triangles, np.int64)
    if len(t) == 0:
        return vertices[:0], colors[:0], t.astype(np.int32)
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
    owner = np.tile(np.arange(len(t)), 3)
    key = e[:, 0] * (int(t.max()) + 1) + e[:, 1]
    o = np.argsort(key, kind="stable")
    ks, os_ = key[o], owner[o]
    same = ks[1:] == ks[:-1]
    g = coo_matrix((np.ones(int(same.sum())), (os_[:-1][same], os_[1:][same])), shape=(len(t), len(t)))
    _, lab = connected_components(g, directed=False)
    n_tri = np.bincount(lab)
    keep_n = min(len(n_tri), 10)
    thr = max(int(np.sort(n_tri)[-keep_n]), 50)
    t = t[n_tri[lab] >= thr]
    used = np.unique(t)
    remap = np.full(len(vertices), -1, np.int64)
    remap[used] = np.arange(len(used))
    t = remap[t]
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]
    return vertices[used], colors[used], t.astype(np.int32)
