import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import synthetic
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)
for name, g in (("surface", synthetic.surface_surfels(100000)[0]), ("stress", synthetic.random_surfels(100000, seed=0)[0])):
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512, flags=1)
    plan.run(); plan.ensure_capacity(); plan.run(); torch.cuda.synchronize()
    st = plan.ws.status().cpu().tolist()
    print(name, "D", st[0], "max_tile", st[2], "blend iters total", st[4], "max per wave", st[5], "chunks", st[6], "iters/chunk", st[4] / max(st[6], 1), "lane utilisation", st[8] / max(64 * st[4], 1))
