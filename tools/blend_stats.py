#!/usr/bin/env python3
"""Blend walk statistics at BASELINE configs[1] (GA_SURFEL_FLAG_STATS): trip slots executed, survivors, and the slots a walk
in which the 64 pixels of a wave do not wait for each other would need (status word BLEND_LANE_MAX)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import synthetic, _lib
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan

dev = torch.device("cuda:0")
points, views, size = int(os.environ.get("POINTS", 100000)), int(os.environ.get("VIEWS", 8)), int(os.environ.get("SIZE", 512))
cams = synthetic.eval_cameras(views)
g = synthetic.surface_surfels(points, seed=1)[0]
m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), size, size,
                         flags=_lib.GA_SURFEL_FLAG_STATS)
plan.run(); plan.ensure_capacity(); plan.run()
torch.cuda.synchronize()
st = plan.ws.status().cpu()
it, sl, ch, lm = int(st[4]), int(st[8]), int(st[6]), int(st[11])
print(f"rendered {int(st[0])}  iters {it}  survivors {sl}  chunks {ch}  lane_max_sum {lm}")
print(f"lane use {sl / (64.0 * it):.3f}   decoupled bound: {lm} slots = {lm / it:.3f} of today's; survivors / (64 lane_max) = {sl / (64.0 * lm):.3f}")
