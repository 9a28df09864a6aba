"""``GaussianRenderer2DGS`` -- same class, method, arguments and returned dict as the reference's
``/root/reference/nsr/gs_surfel.py:21-202`` -- on top of the MI355X surfel rasterizer.

The reference loops over (batch, view) in Python and issues one extension call plus ~10 small torch ops per view.
Here one C-ABI call rasterizes all V views of a batch item (``rasterize_views``) and the post-processing of
``:121-163`` (alpha slice, view->world normal rotation, NaN scrubbing of the median depth, clamp of the image) is one
fused HIP pass over the stacked ``[V,...]`` tensors (``ga_surfel_postprocess``).
"""
from __future__ import annotations

import torch

from . import io_formats
from .diff_surfel_rasterization import postprocess_views, rasterize_views


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

        S = output_size
        dev = gaussians.device
        if torch.is_grad_enabled() and gaussians.requires_grad:
            # training call sites: the rasterizer is differentiable (ga_surfel_backward); the renderer-level post-processing
            # then runs as the reference writes it (nsr/gs_surfel.py:121-163), in PyTorch, so that autograd sees it
            outs = {k: [] for k in ("image", "alpha", "depth", "rend_normal", "dist")}
            for b in range(B):
                g = gaussians[b]
                view = cam_view[b].float()
                color, _radii, allmap, _ = rasterize_views(
                    g[:, 0:3], g[:, 3:4], g[:, 10:13], g[:, 4:6], g[:, 6:10], view, cam_view_proj[b].float(),
                    bg_color.to(g.device), S, S, scale_modifier)
                outs["image"].append(color.clamp(0, 1))
                outs["alpha"].append(allmap[:, 1:2])
                outs["rend_normal"].append(torch.einsum("vchw,vdc->vdhw", allmap[:, 2:5], view[:, :3, :3]))
                outs["depth"].append(torch.nan_to_num(allmap[:, 5:6], 0, 0))
                outs["dist"].append(allmap[:, 6:7])
            return {k: torch.stack(v, dim=0) for k, v in outs.items()}
        image = torch.empty((B, V, 3, S, S), dtype=torch.float32, device=dev)
        rend_normal = torch.empty((B, V, 3, S, S), dtype=torch.float32, device=dev)
        depth = torch.empty((B, V, 1, S, S), dtype=torch.float32, device=dev)
        allmaps = []
        for b in range(B):
            g = gaussians[b]
            view = cam_view[b].float()
            color, _radii, allmap, _ = rasterize_views(
                g[:, 0:3], g[:, 3:4], g[:, 10:13], g[:, 4:6], g[:, 6:10], view, cam_view_proj[b].float(),
                bg_color.to(g.device), S, S, scale_modifier)
            # clamp of the image (:163), view -> world rotation of the normals (:126-128), NaN scrubbing of the median depth
            # (:133-134, depth_ratio = 1) -- one fused pass written straight into the stacked outputs
            postprocess_views(color, allmap, view, image[b], rend_normal[b], depth[b])
            allmaps.append(allmap)
        # alpha / dist are channels 1 / 6 of allmap (:121,142): views for one batch item, one stack otherwise
        am = allmaps[0].unsqueeze(0) if B == 1 else torch.stack(allmaps, dim=0)
        return {"image": image, "alpha": am[:, :, 1:2], "depth": depth, "rend_normal": rend_normal, "dist": am[:, :, 6:7]}

    def save_2dgs_ply(self, path, gaussians, compatible=True):
        """nsr/gs_surfel.py:206-265; the upstream body references undefined names -- ``io_formats.save_2dgs_ply`` writes the
        file it describes (host I/O)."""
        io_formats.save_2dgs_ply(path, gaussians, compatible=compatible)

    def load_2dgs_ply(self, path, compatible=True):
        return torch.from_numpy(io_formats.load_2dgs_ply(path, compatible=compatible))
