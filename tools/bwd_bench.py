#!/usr/bin/env python3
"""Time the rasterizer's forward + backward at BASELINE configs[1] (100k surfels x 8 views x 512^2) on the GPU box.
Usage: python tools/bwd_bench.py [--scene surface|stress] [--reps 20]
Prints one JSON line: forward ms (autograd Function, own workspace), backward ms (ga_surfel_backward: three kernels)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiananything_amd import synthetic  # noqa: E402
from gaussiananything_amd.diff_surfel_rasterization import rasterize_views  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="surface")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cams = synthetic.eval_cameras(a.views)
    g = (synthetic.surface_surfels(a.points, seed=1)[0] if a.scene == "surface" else synthetic.random_surfels(a.points, seed=0)[0])
    m, o, s, r, c = [t.to(dev).requires_grad_(True) for t in synthetic.split_gaussians(g)]
    vm, pm = cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev)
    bg = torch.ones(3, device=dev)
    wc = torch.rand(a.views, 3, a.size, a.size, device=dev)
    wo = torch.rand(a.views, 7, a.size, a.size, device=dev) * 0.1
    fwd, bwd = [], []
    for k in range(a.reps + 3):
        for t in (m, o, s, r, c):
            t.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        color, radii, allmap, _ = rasterize_views(m, o, c, s, r, vm, pm, bg, a.size, a.size)
        e[1].record()
        loss = (color * wc).sum() + (allmap * wo).sum()
        loss.backward()
        e[2].record()
        torch.cuda.synchronize()
        if k >= 3:
            fwd.append(e[0].elapsed_time(e[1]))
            bwd.append(e[1].elapsed_time(e[2]))
    fwd.sort(); bwd.sort()
    print(json.dumps({"scene": a.scene, "points": a.points, "views": a.views, "size": a.size,
                      "forward_ms_median": round(fwd[len(fwd) // 2], 4), "loss_plus_backward_ms_median": round(bwd[len(bwd) // 2], 4),
                      "backward_ms_min": round(bwd[0], 4), "grad_norms": [float(t.grad.norm()) for t in (m, o, s, r, c)]}))


if __name__ == "__main__":
    main()
