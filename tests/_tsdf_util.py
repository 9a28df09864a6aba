"""Synthetic RGB-D frames for the TSDF tests: analytic z-depth of a sphere seen from the in-tree evaluation cameras."""
import numpy as np

from gaussiananything_amd import synthetic
from gaussiananything_amd.mesh import to_cam_open3d_compat


def sphere_frames(n_views=3, size=64, radius=0.3, centre=(0.02, -0.01, 0.03)):
    cams = synthetic.eval_cameras(max(n_views, 1))
    frames = []
    for i in range(n_views):
        c = {"cam_view": cams["cam_view"][i], "cam_pos": cams["cam_pos"][i], "tanfov": cams["tanfov"]}
        intr, ext = to_cam_open3d_compat(c, size)
        fx, fy, cx, cy = intr
        pose = np.linalg.inv(ext)
        v, u = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
        d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u, dtype=np.float64)], -1)   # per unit of z-depth
        d_w = d_cam @ pose[:3, :3].T
        o = pose[:3, 3] - np.asarray(centre)
        a = (d_w * d_w).sum(-1)
        b = 2.0 * (d_w @ o)
        cc = float(o @ o) - radius * radius
        disc = b * b - 4 * a * cc
        hit = disc > 0
        z = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), 0.0)
        p = pose[:3, 3] + d_w * z[..., None]
        n = (p - np.asarray(centre)) / radius
        rgb = np.where(hit[None], np.moveaxis(0.5 + 0.5 * n, -1, 0), 1.0).astype(np.float32)
        alpha = hit.astype(np.float32)
        frames.append(dict(cam=c, intr=intr, ext=ext, rgb=rgb, depth=z.astype(np.float32), alpha=alpha,
                           depth_trunc=float(np.linalg.norm(pose[:3, 3]) + 0.8)))
    return frames
