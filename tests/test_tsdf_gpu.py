"""GPU parity of the HIP TSDF fusion + mesh extraction (through the C-ABI, include/ga_tsdf.h) against oracle/tsdf.py, and the
reference's mesh-export sequence end to end on rendered views.

Bars: the opened units and the triangle index lists are integer artefacts -- identical; tsdf / weight / colour volumes and
vertex positions / colours within 1e-6 (the float expressions are the same operation by operation; the oracle's numpy and
the device may differ in the last bit of a division or square root)."""
import os

import numpy as np
import pytest
import torch

from gaussiananything_amd import mesh, synthetic
from oracle import tsdf as otsdf
from tests._tsdf_util import sphere_frames

pytestmark = pytest.mark.gpu


def _fuse_both(dev, frames, units, unit0, voxel, trunc, alpha_thres=0.08):
    ovol = otsdf.Volume(units, unit0, voxel, trunc)
    ul = voxel * 16
    lo = [(u + 0.5) * ul for u in unit0]
    hi = [(u0 + n - 0.5) * ul for u0, n in zip(unit0, units)]
    hvol = mesh.TSDFVolume(voxel, trunc, lo, hi, device=dev)
    assert hvol.units == list(units) and hvol.unit0 == list(unit0)
    for f in frames:
        touched = otsdf.integrate(ovol, f["rgb"], f["depth"], f["alpha"], alpha_thres, f["depth_trunc"], f["intr"], f["ext"])
        hvol.integrate(torch.from_numpy(f["rgb"]), torch.from_numpy(f["depth"]), f["intr"], f["ext"], f["depth_trunc"],
                       alpha=torch.from_numpy(f["alpha"]), alpha_thres=alpha_thres)
        assert np.array_equal(hvol.touched.cpu().numpy().reshape(units).astype(bool), touched), "opened units differ"
    return ovol, hvol


def test_fusion_matches_the_oracle_voxel_by_voxel(gpu_device):
    frames = sphere_frames(3, 64)
    ovol, hvol = _fuse_both(gpu_device, frames, (4, 4, 4), (-2, -2, -2), 0.0125, 0.075)
    t, w, c = hvol.dense()
    assert np.array_equal(hvol.allocated.cpu().numpy().reshape(4, 4, 4).astype(bool), ovol.allocated)
    assert np.array_equal(w, ovol.weight), "observation counts differ"
    assert w.max() == 3
    assert float(np.abs(t - ovol.tsdf).max()) <= 1e-6
    assert float(np.abs(c - ovol.color).max()) <= 1e-4      # (8-bit colours: values up to 255)


def test_mesh_matches_the_oracle_vertex_by_vertex_and_triangle_by_triangle(gpu_device):
    frames = sphere_frames(4, 96, radius=0.28)
    ovol, hvol = _fuse_both(gpu_device, frames, (4, 4, 4), (-2, -2, -2), 0.0125, 0.075)
    ov, oc, ot = otsdf.extract_mesh(ovol)
    # same volume on both sides, so that a last-bit difference of the fusion cannot flip a sign: the HIP volume is the input
    t, w, c = hvol.dense()
    ovol.tsdf[:], ovol.weight[:], ovol.color[:] = t, w, c
    ov, oc, ot = otsdf.extract_mesh(ovol)
    dv, dc, dt = hvol.extract_triangle_mesh()
    hv, hc, ht = (x.cpu().numpy() for x in (dv, dc, dt))
    assert len(ot) > 2000
    assert hv.shape == ov.shape and ht.shape == ot.shape
    assert np.array_equal(ht, ot), "triangle index lists differ"
    assert float(np.abs(hv - ov).max()) <= 1e-6 and float(np.abs(hc - oc).max()) <= 1e-6
    r = np.linalg.norm(hv.astype(np.float64) - np.array([0.02, -0.01, 0.03]), axis=1)
    assert np.abs(r - 0.28).mean() < 0.004
    # the connected-component filter on the device against the host restatement (scipy), with floaters added
    fv = torch.tensor([[2, 2, 2], [2.1, 2, 2], [2, 2.1, 2], [2, 2, 2.1]], dtype=torch.float32, device=gpu_device)
    ft = torch.tensor([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], dtype=torch.int32, device=gpu_device) + dv.shape[0]
    v2, c2, t2 = torch.cat([dv, fv]), torch.cat([dc, torch.zeros_like(fv)]), torch.cat([dt, ft])
    pv, pc, pt = mesh.post_process_mesh(v2, c2, t2)
    ov2, oc2, ot2 = otsdf.post_process_mesh(v2.cpu().numpy(), c2.cpu().numpy(), t2.cpu().numpy())
    assert np.array_equal(pt.cpu().numpy(), ot2) and np.array_equal(pv.cpu().numpy(), ov2) and np.array_equal(pc.cpu().numpy(), oc2)
    assert len(ot2) < len(t2)


def test_cluster_labels_union_find_against_scipy(gpu_device):
    """ga_mesh_cluster_labels (round 5: one lock-free union-find pass instead of seven rounds of device-wide torch operations): random pair
    lists -- long chains, many small clusters, isolated triangles, duplicate and self pairs -- give exactly the labels scipy's connected
    components give when every cluster is named by its smallest member, run to run (the labels do not depend on the service order)."""
    import ctypes
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    from gaussiananything_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n, m, chain in ((1000, 700, False), (200_000, 150_000, False), (50_000, 49_999, True), (10, 0, False)):
        if chain:
            perm = rng.permutation(n)
            a, b = perm[:-1].copy(), perm[1:].copy()          # one cluster, a chain in random index order
        else:
            a, b = rng.integers(0, n, m), rng.integers(0, n, m)
        g = sp.coo_matrix((np.ones(len(a)), (a, b)), shape=(n, n))
        _, comp = connected_components(g, directed=False)
        smallest = np.full(comp.max() + 1, n, np.int64)
        np.minimum.at(smallest, comp, np.arange(n))
        want = smallest[comp]
        ta, tb = torch.from_numpy(a.astype(np.int64)).to(gpu_device), torch.from_numpy(b.astype(np.int64)).to(gpu_device)
        for _ in range(2):
            lab = torch.empty(n, dtype=torch.int32, device=gpu_device)
            rc = L.ga_mesh_cluster_labels(ta.data_ptr() if len(a) else None, tb.data_ptr() if len(a) else None, len(a), lab.data_ptr(), n,
                                          ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream))
            assert rc == 0
            assert np.array_equal(lab.cpu().numpy().astype(np.int64), want)


def test_partially_observed_volume_and_empty_volume(gpu_device):
    frames = sphere_frames(1, 64)
    ovol, hvol = _fuse_both(gpu_device, frames, (3, 4, 5), (-1, -2, -3), 0.0125, 0.075)
    t, w, c = hvol.dense()
    ovol.tsdf[:], ovol.weight[:], ovol.color[:] = t, w, c
    ov, oc, ot = otsdf.extract_mesh(ovol)
    hv, hc, ht = (x.cpu().numpy() for x in hvol.extract_triangle_mesh())
    assert len(ot) > 100 and np.array_equal(ht, ot) and float(np.abs(hv - ov).max()) <= 1e-6
    empty = mesh.TSDFVolume(0.0125, 0.075, [-0.3] * 3, [0.3] * 3, device=gpu_device)
    v, c2, t2 = empty.extract_triangle_mesh()
    assert v.shape == (0, 3) and t2.shape == (0, 3)
    # a frame with no valid depth opens nothing
    f = frames[0]
    empty.integrate(torch.from_numpy(f["rgb"]), torch.zeros(64, 64), f["intr"], f["ext"], f["depth_trunc"])
    assert int(empty.allocated.sum()) == 0 and float(empty.weight.max()) == 0


def test_export_mesh_from_rendered_views_end_to_end(gpu_device, tmp_path):
    """The reference's sequence (flow_matching_trainer.py:1244-1395) on views rendered by the HIP rasterizer: 8 cameras,
    512 x 512, aabb +-0.495, voxel = radius / 160.  The fused mesh of a shell of opaque surfels is that shell."""
    from gaussiananything_amd.gs_surfel import GaussianRenderer2DGS
    n = 60000
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    xyz = 0.3 * d
    # surfels tangent to the sphere: the quaternion that turns +z into the normal d
    z = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3)
    axis = torch.cross(z, d, dim=-1)
    quat = torch.nn.functional.normalize(torch.cat([1.0 + (z * d).sum(-1, keepdim=True), axis], -1), dim=-1)
    gauss = torch.cat([xyz, torch.full((n, 1), 0.99), torch.full((n, 2), 0.006), quat, 0.5 + 0.5 * d], 1)[None].to(gpu_device)
    cams = synthetic.eval_cameras(8)
    r = GaussianRenderer2DGS(512, 8, {})
    cv, cvp, cp = (cams[k][None].to(gpu_device) for k in ("cam_view", "cam_view_proj", "cam_pos"))
    out = r.render(gauss, cv, cvp, cp, cams["tanfov"])
    rgbs = [out["image"][0, i][None] for i in range(8)]
    depths = [out["depth"][0, i][None] for i in range(8)]
    alphas = [out["alpha"][0, i][None] for i in range(8)]
    cam_pathes = [{"cam_view": cams["cam_view"][i], "cam_pos": cams["cam_pos"][i], "tanfov": cams["tanfov"]} for i in range(8)]
    raw = os.path.join(tmp_path, "0", "0-mesh_raw.obj")
    post = mesh.export_mesh_from_2dgs(rgbs, depths, alphas, cam_pathes, raw)
    assert post.endswith("0-mesh.obj") and os.path.exists(raw) and os.path.exists(post)
    v = np.array([[float(x) for x in l.split()[1:4]] for l in open(raw) if l.startswith("v ")])
    nt = sum(1 for l in open(raw) if l.startswith("f "))
    assert len(v) > 20000 and nt > 40000
    rr = np.linalg.norm(v, axis=1)
    assert np.median(np.abs(rr - 0.3)) < 0.004 and np.percentile(np.abs(rr - 0.3), 99) < 0.03
    pv = np.array([[float(x) for x in l.split()[1:4]] for l in open(post) if l.startswith("v ")])
    assert 0 < len(pv) <= len(v) and abs(np.median(np.linalg.norm(pv, axis=1)) - 0.3) < 0.004   # rotated about the origin
