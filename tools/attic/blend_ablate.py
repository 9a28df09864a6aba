import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gaussiananything_amd import synthetic
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)
scene = sys.argv[1] if len(sys.argv) > 1 else "surface"
g = (synthetic.surface_surfels(100000)[0] if scene == "surface" else synthetic.random_surfels(100000)[0])
m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
for flags in (0, 8, 2, 10):  # 8 = tile-per-workgroup kernel, 2 = staging only
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512, flags=flags)
    for _ in range(5): plan.run()
    ev = [bench.HipEvents(5) for _ in range(20)]
    for e in ev:
        plan.set_stage_events(e.arr); plan.run()
    torch.cuda.synchronize()
    import numpy as np
    print("flags", flags, "blend ms", np.mean([e.elapsed(3, 4) for e in ev]))
