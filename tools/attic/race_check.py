"""Repeat the rasterizer forward of the stress / surface scenes and compare every run with the first (bit-exact): a data race
shows up as run-to-run differences.  usage: python tools/race_check.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiananything_amd import synthetic
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for scene in ("surface", "stress"):
    g = synthetic.surface_surfels(100000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100000, seed=0)[0]
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512)
    plan.run(); plan.ensure_capacity(); plan.run(); torch.cuda.synchronize()
    ref_c, ref_a = plan.color.clone(), plan.allmap.clone()
    bad = 0
    for k in range(reps):
        plan.color.zero_(); plan.allmap.zero_()
        plan.run(); torch.cuda.synchronize()
        dc = (plan.color != ref_c) | (plan.color.isnan() != ref_c.isnan())
        da = (plan.allmap != ref_a)
        n = int(dc.sum()) + int(da.sum())
        if n:
            bad += 1
            idx = torch.nonzero(da.any(1) | dc.any(1))
            print(scene, "run", k, "differs in", n, "values; first pixel (view, y, x):", idx[0].tolist(), "tile", (idx[0][1] // 16).item(), (idx[0][2] // 16).item(), flush=True)
    print(scene, "runs differing from the first:", bad, "of", reps, flush=True)
