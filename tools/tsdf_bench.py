#!/usr/bin/env python3
"""Time the mesh export at the reference's size on the GPU box: 8 rendered 512^2 views of a 100k-surfel scene fused into the
volume of extract_mesh_bounded (voxel = radius / 160: 352^3 voxels), then marching cubes.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiananything_amd import mesh, synthetic  # noqa: E402
from gaussiananything_amd.gs_surfel import GaussianRenderer2DGS  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cams = synthetic.eval_cameras(8)
    g = synthetic.surface_surfels(100_000, seed=1).to(dev)
    g[..., 3] = 0.95
    r = GaussianRenderer2DGS(512, 8, {})
    cv, cvp, cp = (cams[k][None].to(dev) for k in ("cam_view", "cam_view_proj", "cam_pos"))
    out = r.render(g, cv, cvp, cp, cams["tanfov"])
    rgbs = [out["image"][0, i][None] for i in range(8)]
    depths = [out["depth"][0, i][None] for i in range(8)]
    alphas = [out["alpha"][0, i][None] for i in range(8)]
    cam_pathes = [{"cam_view": cams["cam_view"][i], "cam_pos": cams["cam_pos"][i], "tanfov": cams["tanfov"]} for i in range(8)]
    aabb = np.array([-0.45, -0.45, -0.45, 0.45, 0.45, 0.45]).reshape(2, 3) * 1.1
    center, radius = aabb.mean(0), float(np.linalg.norm(aabb[1] - aabb[0]) * 0.5)
    voxel, trunc = radius / 160, radius / 160 * 12
    res = {}
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vol = mesh.TSDFVolume(voxel, trunc, center - radius - trunc, center + radius + trunc, device=dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for i, cam in enumerate(cam_pathes):
            intr, ext = mesh.to_cam_open3d_compat(cam, 512)
            campos = cam["cam_pos"].numpy().astype(np.float64)
            vol.integrate(rgbs[i][0], depths[i][0], intr, ext, float(np.linalg.norm(campos - center) + radius), alpha=alphas[i][0], alpha_thres=0.08)
        ev[1].record()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        v, c, t = vol.extract_triangle_mesh()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        opened = int(vol.allocated.sum())
        res = {"units": vol.units, "voxels": vol.nvox, "units_opened": opened, "alloc_ms": round((t1 - t0) * 1e3, 3),
               "fuse8_ms_wall": round((t2 - t1) * 1e3, 3), "fuse8_ms_device": round(ev[0].elapsed_time(ev[1]), 3),
               "mesh_ms": round((t3 - t2) * 1e3, 3), "vertices": int(v.shape[0]), "triangles": int(t.shape[0]),
               "integrate_bytes_per_frame_upper": opened * 4096 * 20 * 2}
    for rep in range(3):
        torch.cuda.synchronize(); tt = time.perf_counter()
        pv, pc, pt = mesh.post_process_mesh(v, c, t)
        torch.cuda.synchronize(); res["post_process_ms"] = round((time.perf_counter() - tt) * 1e3, 2)
    res["post_triangles"] = int(pt.shape[0]); res["post_process_rounds"] = mesh.post_process_mesh.rounds
    tt = time.perf_counter()
    mesh.write_obj("/tmp/tsdf_bench_raw.obj", v, c, t)
    res["write_obj_raw_ms"] = round((time.perf_counter() - tt) * 1e3, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
