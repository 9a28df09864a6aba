"""Camera conventions of the surfel render path (host side, numpy/torch; negligible cost).

Mirrors, with the same names and argument meaning:
  * ``focal2fov`` / ``getWorld2View2`` / ``getProjectionMatrix``
        -- /root/reference/utils/gs_utils/graphics_utils.py:38-79,88-89
  * ``c_to_3dgs_format`` (pose25 -> cam_view, cam_view_proj, cam_pos, tanfov)
        -- /root/reference/nsr/lsgm/flow_matching_trainer.py:2174-2228
All returned matrices are in the reference's ROW-VECTOR convention (``p_view = [p,1] @ cam_view``), i.e. the
transposes of the usual column-vector matrices; that is what ``GaussianRasterizationSettings`` expects.
``c_to_3dgs_format_batched`` is the device-friendly batched form.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World->view 4x4 (column-vector form) from the transposed rotation ``R`` and translation ``t``; the optional
    re-centring moves the camera centre by ``translate`` then scales it (identity in every reference call)."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = t
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Symmetric-frustum perspective matrix with z mapped to [0,1] and w = +z (column-vector form)."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2.0 * tx * znear)
    P[1, 1] = 2.0 * znear / (2.0 * ty * znear)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    P[3, 2] = 1.0
    return P


def c_to_3dgs_format(pose, znear=0.01, zfar=100.0):
    """pose: numpy float[25] = c2w(16, row-major) + intrinsics(9, normalised; pose[16] = fx)."""
    pose = np.asarray(pose)
    c2w = pose[:16].reshape(4, 4)
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    fx = pose[16]
    FovX = focal2fov(fx, 1)
    FovY = focal2fov(fx, 1)
    tanfovx = math.tan(FovX * 0.5)
    world_view_transform = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
    projection_matrix = getProjectionMatrix(znear=znear, zfar=zfar, fovX=FovX, fovY=FovY).transpose(0, 1)
    full_proj_transform = (world_view_transform.unsqueeze(0).bmm(projection_matrix.unsqueeze(0))).squeeze(0)
    camera_center = world_view_transform.inverse()[3, :3]
    return dict(projection_matrix=projection_matrix, cam_view=world_view_transform,
                cam_view_proj=full_proj_transform, cam_pos=camera_center, tanfov=tanfovx,
                orig_pose=torch.from_numpy(np.asarray(pose)))


def c_to_3dgs_format_batched(poses, znear=0.01, zfar=100.0):
    """poses: [V,25] -> dict of stacked ``cam_view [V,4,4]``, ``cam_view_proj [V,4,4]``, ``cam_pos [V,3]``, ``tanfov``."""
    items = [c_to_3dgs_format(np.asarray(p), znear, zfar) for p in np.asarray(poses)]
    return dict(cam_view=torch.stack([c["cam_view"] for c in items]),
                cam_view_proj=torch.stack([c["cam_view_proj"] for c in items]),
                cam_pos=torch.stack([c["cam_pos"] for c in items]),
                tanfov=items[0]["tanfov"])


def c_to_3dgs_format_device(poses: torch.Tensor, znear=0.01, zfar=100.0):
    """The same conversion as ONE batched tensor program on whatever device ``poses`` lives on (SURVEY.md section 8(f)-2:
    the video path renders 50 cameras x 4 levels per sample, flow_matching_trainer.py:1545-1616, and the reference converts
    them one by one in numpy).  poses: [..., 25] -> ``cam_view`` / ``cam_view_proj`` [..., 4, 4] (row-vector convention),
    ``cam_pos`` [..., 3], ``tanfov`` (python float of the first camera, as the reference passes a scalar)."""
    poses = poses.float()
    c2w = poses[..., :16].reshape(*poses.shape[:-1], 4, 4)
    w2c = torch.linalg.inv(c2w)                       # column-vector world->view (getWorld2View2 with no re-centring)
    cam_view = w2c.transpose(-1, -2)
    fx = poses[..., 16]
    tan = 1.0 / (2.0 * fx)                            # tan(focal2fov(fx, 1) / 2)
    P = torch.zeros(*poses.shape[:-1], 4, 4, device=poses.device)
    P[..., 0, 0] = 1.0 / tan
    P[..., 1, 1] = 1.0 / tan
    P[..., 2, 2] = zfar / (zfar - znear)
    P[..., 2, 3] = -(zfar * znear) / (zfar - znear)
    P[..., 3, 2] = 1.0
    cam_view_proj = cam_view @ P.transpose(-1, -2)
    cam_pos = c2w[..., :3, 3]                         # = inverse(cam_view)[3, :3]
    return dict(cam_view=cam_view, cam_view_proj=cam_view_proj, cam_pos=cam_pos,
                tanfov=float(tan.reshape(-1)[0]))


def orbit_poses(num_views, radius=1.77, fx=1.3889, elevation_deg=15.0, seed=None):
    """Synthetic look-at-origin orbit in the pose25 format (c2w + normalised K), for inputs that must not depend
    on reference assets.  Camera looks down its +z axis at the origin (the convention of ``eval_pose.pt``)."""
    poses = np.zeros((num_views, 25), np.float32)
    for v in range(num_views):
        az = 2.0 * math.pi * v / num_views
        el = math.radians(elevation_deg)
        pos = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
        fwd = -pos / np.linalg.norm(pos)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(up, fwd); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        poses[v, :16] = c2w.reshape(-1)
        poses[v, 16:] = [fx, 0, 0.5, 0, fx, 0.5, 0, 0, 1]
    return poses
