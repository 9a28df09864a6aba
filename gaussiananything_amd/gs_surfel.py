"""``GaussianRenderer2DGS`` -- same class, method, arguments and returned dict as the reference's
``/root/reference/nsr/gs_surfel.py:21-202`` -- on top of the MI355X surfel rasterizer.

The reference loops over (batch, view) in Python and issues one extension call plus ~10 small torch ops per view.
Here one C-ABI call rasterizes all V views of a batch item (``rasterize_views``) and the post-processing of
``:121-163`` (alpha slice, view->world normal rotation, NaN scrubbing of the median depth, clamp of the image) is one
fused HIP pass over the stacked ``[V,...]`` tensors (``ga_surfel_postprocess``).
"""
from __future__ import annotations

import os

import torch

from . import io_formats
from .diff_surfel_rasterization import _get_workspace, postprocess_views, rasterize_views
from . import _lib


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
        return self._render_nograd(gaussians, cam_view, cam_view_proj, bg_color, scale_modifier, S, True)[0]

    def _render_nograd(self, gaussians, cam_view, cam_view_proj, bg_color, scale_modifier, S, check_overflow):
        """The inference branch of ``render``: per batch item one rasterizer call for all V views and one fused post-processing
        pass.  Returns (dict, workspaces used): with ``check_overflow=False`` nothing synchronises and the caller reads the
        workspaces' status words itself (``render_levels``)."""
        B, V = cam_view.shape[:2]
        dev = gaussians.device
        image = torch.empty((B, V, 3, S, S), dtype=torch.float32, device=dev)
        rend_normal = torch.empty((B, V, 3, S, S), dtype=torch.float32, device=dev)
        depth = torch.empty((B, V, 1, S, S), dtype=torch.float32, device=dev)
        allmaps, used = [], []
        for b in range(B):
            g = gaussians[b]
            view = cam_view[b].float()
            color, _radii, allmap, ws = rasterize_views(
                g[:, 0:3], g[:, 3:4], g[:, 10:13], g[:, 4:6], g[:, 6:10], view, cam_view_proj[b].float(),
                bg_color.to(g.device), S, S, scale_modifier, check_overflow=check_overflow)
            # clamp of the image (:163), view -> world rotation of the normals (:126-128), NaN scrubbing of the median depth
            # (:133-134, depth_ratio = 1) -- one fused pass written straight into the stacked outputs
            postprocess_views(color, allmap, view, image[b], rend_normal[b], depth[b])
            allmaps.append(allmap)
            used.append(ws)
        # alpha / dist are channels 1 / 6 of allmap (:121,142): views for one batch item, one stack otherwise
        am = allmaps[0].unsqueeze(0) if B == 1 else torch.stack(allmaps, dim=0)
        return {"image": image, "alpha": am[:, :, 1:2], "depth": depth, "rend_normal": rend_normal, "dist": am[:, :, 6:7]}, used

    @torch.no_grad()
    def render_levels(self, gaussian_sets, output_sizes, cam_view, cam_view_proj, cam_pos, tanfov, bg_color=None, scale_modifier=1):
        """Several INDEPENDENT surfel sets for the same cameras -- the four levels of ``triplane_decode``
        (vit/vit_triplane.py:1550-1591; 50 views x 4 levels in the video path, flow_matching_trainer.py:1545-1616) -- rendered as
        ``[render(g, ..., output_size=s) for g, s in zip(gaussian_sets, output_sizes)]`` would, but overlapped: set k runs on side
        stream k % 2, so the latency-bound front-end (preprocess / binning / sort) of one set hides under the issue-bound blend of
        its neighbour (two independent forwards on two streams: +11 %, profiles/r5_overlap_probe.txt), and the overflow words of
        all sets are read back ONCE after the last launch instead of one host synchronisation per set.  A set whose workspace
        overflowed (first use of a shape with unusually long lists) is rendered again the ordinary way."""
        if bg_color is None:
            bg_color = self.bg_color
        sets = [g.contiguous().float() for g in gaussian_sets]
        dev = sets[0].device
        if dev.type != "cuda":
            raise RuntimeError("gaussiananything_amd surfel rasterizer only runs on an MI355X (HIP) device; there is no CPU path")
        if cam_view.shape[0] != 1 or len(sets) < 2 or os.environ.get("GA_RENDER_LEVELS", "1") == "0":   # (GA_RENDER_LEVELS=0: A/B aid; batch items of one set share a workspace, hence its status words: one at a time)
            return [self.render(g, cam_view, cam_view_proj, cam_pos, tanfov, bg_color, scale_modifier, S) for g, S in zip(sets, output_sizes)]
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None or self._side[0].device != dev:
            self._side = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        B, V = cam_view.shape[:2]
        lane_of = {}
        for k, (g, S) in enumerate(zip(sets, output_sizes)):        # workspaces are long-lived: created on the caller's stream; two sets
            _get_workspace(dev, g.shape[1], V, S, S)                 # of one shape share a workspace and therefore a stream
            lane_of.setdefault((g.shape[1], S), k % 2)
        fork = torch.cuda.Event()
        fork.record(main)
        results = []
        for g, S in zip(sets, output_sizes):
            st = self._side[lane_of[(g.shape[1], S)]]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                res, used = self._render_nograd(g, cam_view, cam_view_proj, bg_color, scale_modifier, S, False)
            for t in (res["image"], res["rend_normal"], res["depth"], res["alpha"]):
                t.record_stream(main)                                # (allocated on the side stream, consumed on the caller's)
            results.append((res, used))
        for st in self._side:
            main.wait_stream(st)
        words = torch.stack([ws.status()[:4] for _, used in results for ws in used]).cpu()       # the one synchronisation
        at = 0
        out = []
        for (res, used), g, S in zip(results, sets, output_sizes):
            bad = bool(words[at:at + len(used), _lib.GA_STATUS_OVERFLOW].any())
            at += len(used)
            out.append(self.render(g, cam_view, cam_view_proj, cam_pos, tanfov, bg_color, scale_modifier, S) if bad else res)
        return out

    def save_2dgs_ply(self, path, gaussians, compatible=True):
        """nsr/gs_surfel.py:206-265; the upstream body references undefined names -- ``io_formats.save_2dgs_ply`` writes the
        file it describes (host I/O)."""
        io_formats.save_2dgs_ply(path, gaussians, compatible=compatible)

    def load_2dgs_ply(self, path, compatible=True):
        return torch.from_numpy(io_formats.load_2dgs_ply(path, compatible=compatible))
