"""CPU tests of the TSDF / mesh-export oracle and host code: the derived marching-cubes table,
the restatement of Open3D's integration on analytic frames, the post-processing and the OBJ writer.  Open3D is absent:
parity with the reference's mesh is UNPINNED (oracle/tsdf.py header); what is checked here are the properties any correct
fusion + extraction has."""
import os

import numpy as np

from oracle import tsdf as otsdf
from tests._tsdf_util import sphere_frames


def _edge_counts(tris):
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    und = np.sort(e, axis=1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    _, dcnt = np.unique(e, axis=0, return_counts=True)
    return cnt, dcnt


def _sphere_volume(res_units=4, radius=0.31, voxel=0.02, seed=0):
    vol = otsdf.Volume((res_units,) * 3, (-(res_units // 2),) * 3, voxel, 0.1)
    R = res_units * 16
    idx = (np.arange(R) + 0.5) * voxel + vol.unit0[0] * vol.unit_length
    x, y, z = np.meshgrid(idx, idx, idx, indexing="ij")
    rng = np.random.default_rng(seed)
    sdf = np.sqrt((x - 0.013) ** 2 + (y + 0.021) ** 2 + (z - 0.007) ** 2) - radius
    vol.tsdf[:] = np.clip(sdf / 0.1, -1, 1).astype(np.float32)
    vol.weight[:] = 1
    vol.color[:] = rng.uniform(0, 255, vol.color.shape).astype(np.float32)
    vol.allocated[:] = True
    return vol


def test_every_case_of_the_table_is_a_closed_cut_of_the_cube():
    t = otsdf.mc_table()
    assert t["max_triangles"] == 5 and len(t["triangles"]) == 256
    corners = t["edge_corners"]
    for case, row in enumerate(t["triangles"]):
        edges = [e for e in row if e >= 0]
        cut = {e for e in range(12) if ((case >> corners[e][0]) ^ (case >> corners[e][1])) & 1}
        assert set(edges) == cut, case            # every intersected edge is used, no other
        assert len(edges) % 3 == 0


def test_mesh_of_a_sphere_is_closed_oriented_and_has_the_sphere_s_volume():
    vol = _sphere_volume()
    v, c, t = otsdf.extract_mesh(vol)
    assert len(v) > 500 and len(t) > 1000 and t.min() == 0 and t.max() == len(v) - 1
    cnt, dcnt = _edge_counts(t)
    assert (cnt == 2).all() and (dcnt == 1).all()          # closed, consistently oriented
    E = len(cnt)
    assert len(v) - E + len(t) == 2                          # one sphere
    p = v.astype(np.float64)
    vol6 = np.einsum("ij,ij->i", p[t[:, 0]], np.cross(p[t[:, 1]], p[t[:, 2]])).sum() / 6.0
    assert abs(vol6 - 4 / 3 * np.pi * 0.31 ** 3) / (4 / 3 * np.pi * 0.31 ** 3) < 0.01     # outward normals: positive volume
    r = np.linalg.norm(p - np.array([0.013, -0.021, 0.007]), axis=1)
    assert np.abs(r - 0.31).max() < 0.02 * 0.25               # vertices on the zero level set (a quarter voxel)
    assert c.min() >= 0 and c.max() <= 1


def test_cubes_with_an_unobserved_corner_give_no_triangles():
    vol = _sphere_volume()
    vol.weight[:32] = 0                                        # half the volume never observed
    v, c, t = otsdf.extract_mesh(vol)
    assert len(t) > 0 and v[:, 0].min() > (32 + 0.5) * 0.02 + vol.unit0[0] * vol.unit_length - 1e-6
    cnt, _ = _edge_counts(t)
    assert (cnt <= 2).all() and (cnt == 1).any()               # an open boundary where the observed part ends


def test_integration_of_analytic_sphere_frames():
    fr = sphere_frames(3, 64)
    vol = otsdf.Volume((4, 4, 4), (-2, -2, -2), 0.0125, 0.075)
    for f in fr:
        touched = otsdf.integrate(vol, f["rgb"], f["depth"], f["alpha"], 0.08, f["depth_trunc"], f["intr"], f["ext"])
        assert touched.any()
    assert vol.weight.max() == 3 and vol.allocated.any()
    R = 64
    idx = (np.arange(R) + 0.5) * 0.0125 - 2 * vol.unit_length
    x, y, z = np.meshgrid(idx, idx, idx, indexing="ij")
    true = np.sqrt((x - 0.02) ** 2 + (y + 0.01) ** 2 + (z - 0.03) ** 2) - 0.3
    seen3 = (vol.weight == 3) & (np.abs(true) < 0.03)
    assert seen3.sum() > 1000
    # (the projective distance along the ray exceeds the true distance by the cosine of the incidence angle, and the three
    # cameras see the band around the silhouettes at grazing angles: loose bars on the volume, tight ones on the mesh)
    err = vol.tsdf[seen3] * 0.075 - true[seen3]
    assert np.abs(err).mean() < 0.015 and (np.sign(vol.tsdf[seen3]) == np.sign(true[seen3])).mean() > 0.93
    v, c, t = otsdf.extract_mesh(vol)
    r = np.linalg.norm(v.astype(np.float64) - np.array([0.02, -0.01, 0.03]), axis=1)
    assert len(t) > 1000 and np.abs(r - 0.3).mean() < 0.004 and np.abs(r - 0.3).max() < 0.02


def test_post_process_keeps_large_clusters_only_and_obj_writer(tmp_path):
    from gaussiananything_amd import mesh
    vol = _sphere_volume()
    v, c, t = otsdf.extract_mesh(vol)
    # a floater of four triangles far away
    fv = np.array([[2, 2, 2], [2.1, 2, 2], [2, 2.1, 2], [2, 2, 2.1]], np.float32)
    ft = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], np.int32) + len(v)
    v2, c2, t2 = np.concatenate([v, fv]), np.concatenate([c, np.zeros((4, 3), np.float32)]), np.concatenate([t, ft])
    for fn in (mesh.post_process_mesh, otsdf.post_process_mesh):
        pv, pc, pt = fn(v2, c2, t2)
        assert len(pv) == len(v) and len(pt) == len(t) and np.array_equal(pt, t) and np.array_equal(pv, v)
    path = os.path.join(tmp_path, "a", "0-mesh_raw.obj")
    mesh.write_obj(path, v, c, t)
    lines = open(path).read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == len(v) and sum(l.startswith("f ") for l in lines) == len(t)
    assert np.allclose(mesh.rotation_matrix_x(-90) @ np.array([0, 1, 0]), [0, 0, -1]) and np.allclose(
        mesh.rotation_matrix_y(np.pi) @ np.array([1, 0, 0]), [-1, 0, 0], atol=1e-12)


def test_camera_conversion_and_post_processing_against_the_reference_s_own_functions():
    """tests/golden/mesh_cam_ref.npz: produced by running /root/reference/utils/mesh_util.py (to_cam_open3d_compat,
    post_process_mesh) itself, see tests/golden/make_mesh_golden.py."""
    import torch
    from gaussiananything_amd import cameras, mesh, synthetic
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_cam_ref.npz"))
    cams = synthetic.eval_cameras(8)
    for i in range(8):
        c = {"cam_view": cams["cam_view"][i], "cam_pos": cams["cam_pos"][i], "tanfov": cams["tanfov"]}
        intr, ext = mesh.to_cam_open3d_compat(c, 512)
        assert intr == tuple(z["intrinsics"][i][:4]), (intr, z["intrinsics"][i])
        assert np.array_equal(ext, z["extrinsics"][i])
    # ... and the branch that takes the reference's projection matrix
    fov = cameras.focal2fov(float(cams["poses"][0][16]), 1)
    pm = cameras.getProjectionMatrix(0.01, 100.0, fov, fov).transpose(0, 1)
    intr2, ext2 = mesh.to_cam_open3d_compat({"cam_view": cams["cam_view"][0], "projection_matrix": pm}, 512)
    assert intr2 == tuple(z["intrinsics"][0][:4]) and np.array_equal(ext2, z["extrinsics"][0])
    pv, pc, pt = mesh.post_process_mesh(torch.from_numpy(z["pp_vertices"]), torch.zeros(len(z["pp_vertices"]), 3),
                                        torch.from_numpy(z["pp_triangles"]))
    assert np.array_equal(pt.numpy(), z["pp_out_triangles"]) and np.array_equal(pv.numpy(), z["pp_out_vertices"])
    ov, oc, ot = otsdf.post_process_mesh(z["pp_vertices"], np.zeros((len(z["pp_vertices"]), 3), np.float32), z["pp_triangles"])
    assert np.array_equal(ot, z["pp_out_triangles"]) and np.array_equal(ov, z["pp_out_vertices"])


def test_random_fields_give_closed_meshes_through_every_ambiguous_case():
    """White-noise signs put every one of the 256 cube cases, including all ambiguous faces, next to every other: the derived
    table must still give a surface without cracks -- every mesh edge that is not on the volume's outer faces is shared by
    exactly two triangles, with opposite directions."""
    rng = np.random.default_rng(7)
    vol = otsdf.Volume((2, 2, 2), (0, 0, 0), 1.0, 1.0)
    vol.tsdf[:] = rng.uniform(-1, 1, vol.tsdf.shape).astype(np.float32)
    vol.weight[:] = 1
    vol.allocated[:] = True
    v, c, t = otsdf.extract_mesh(vol)
    cases = set()
    ins = vol.tsdf < 0
    R = 32
    cs = np.zeros((R - 1,) * 3, np.int32)
    for i in range(8):
        o = (i & 1, (i >> 1) & 1, (i >> 2) & 1)
        cs |= ins[o[0]:R - 1 + o[0], o[1]:R - 1 + o[1], o[2]:R - 1 + o[2]].astype(np.int32) << i
    assert len(np.unique(cs)) >= 250                                  # (practically all of the 256 cases occur)
    assert len(t) > 50000
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    und = np.sort(e, axis=1)
    uniq, inv, cnt = np.unique(und, axis=0, return_inverse=True, return_counts=True)
    lo, hi = 0.5, R - 0.5                                             # voxel centres span [0.5, 31.5]
    on_face = lambda p: (np.isclose(p, lo) | np.isclose(p, hi)).any(axis=1)
    boundary_edge = on_face(v[uniq[:, 0]]) & on_face(v[uniq[:, 1]])
    assert (cnt[~boundary_edge] == 2).all() and (cnt <= 2).all()
    # orientation: an interior edge is traversed once in each direction
    direction = np.where(e[:, 0] < e[:, 1], 1, -1)
    balance = np.bincount(inv.reshape(-1), weights=direction, minlength=len(uniq))
    assert (balance[~boundary_edge] == 0).all()
