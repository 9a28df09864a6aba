"""``GaussianRenderer2DGS`` -- same class, method, arguments and returned dict as the reference's
``/root/reference/nsr/gs_surfel.py:21-202`` -- on top of the MI355X surfel rasterizer.

The reference loops over (batch, view) in Python and issues one extension call plus ~10 small torch ops per view.
Here one C-ABI call rasterizes all V views of a batch item (``rasterize_views``) and the post-processing of
``:121-163`` (alpha slice, view->world normal rotation, NaN scrubbing of the median depth, clamp of the image) is done
once over the stacked ``[V,...]`` tensors.
"""
from __future__ import annotations

import torch

from .diff_surfel_rasterization import rasterize_views


class GaussianRenderer2DGS:
    def __init__(self, output_size, out_chans, rendering_kwargs, **kwargs):
        self.bg_color = torch.tensor([1, 1, 1], dtype=torch.float32, device="cuda")
        self.output_size = output_size
        self.out_chans = out_chans
        self.rendering_kwargs = rendering_kwargs

    def render(self, gaussians, cam_view, cam_view_proj, cam_pos, tanfov, bg_color=None, scale_modifier=1,
               output_size=None):
        # gaussians: [B, N, 13] = xyz(3) opacity(1) scale(2) rotation wxyz(4) rgb(3); cam_*: [B, V, 4, 4] / [B, V, 3]
        if output_size is None:
            output_size = self.output_size
        B, V = cam_view.shape[:2]
        assert gaussians.shape[2] == 13  # scale with 2dof
        gaussians = gaussians.contiguous().float()
        if bg_color is None:
            bg_color = self.bg_color

        images, alphas, depths, rend_normals, dists = [], [], [], [], []
        for b in range(B):
            g = gaussians[b]
            view = cam_view[b].float()
            color, _radii, allmap, _ = rasterize_views(
                g[:, 0:3], g[:, 3:4], g[:, 10:13], g[:, 4:6], g[:, 6:10], view, cam_view_proj[b].float(),
                bg_color.to(g.device), output_size, output_size, scale_modifier)
            # normals: view space -> world space, n_world = n_view @ view[:3,:3].T   (nsr/gs_surfel.py:126-128)
            normal = torch.einsum("vchw,vdc->vdhw", allmap[:, 2:5], view[:, :3, :3])
            images.append(color.clamp(0, 1))
            alphas.append(allmap[:, 1:2])
            depths.append(torch.nan_to_num(allmap[:, 5:6], 0, 0))  # depth_ratio = 1: median depth (:133-134,150)
            rend_normals.append(normal)
            dists.append(allmap[:, 6:7])

        return {
            "image": torch.stack(images, dim=0).view(B, V, 3, output_size, output_size),
            "alpha": torch.stack(alphas, dim=0).view(B, V, 1, output_size, output_size),
            "depth": torch.stack(depths, dim=0).view(B, V, 1, output_size, output_size),
            "rend_normal": torch.stack(rend_normals, dim=0).view(B, V, 3, output_size, output_size),
            "dist": torch.stack(dists, dim=0).view(B, V, 1, output_size, output_size),
        }
