"""Timeline of the fill + scan launch (GA_FILL_STAMPS build): (start, end) per workgroup, row 0 = the schedule workgroup.
usage (GPU box): python tools/fill_stamps.py <variant name>"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")
name = sys.argv[1]
backup = MAIN + ".stamp_backup"
shutil.copy(MAIN, backup)
try:
    shutil.copy(os.path.join(ROOT, "tools", "_build", f"libga_{name}.so"), MAIN)
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from gaussiananything_amd import synthetic
    from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
    dev = torch.device("cuda:0")
    cams = synthetic.eval_cameras(8)
    g = synthetic.surface_surfels(100000, seed=1)[0]
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512)
    plan.run(); plan.ensure_capacity()
    for _ in range(4):
        plan.run()
    torch.cuda.synchronize()
    rows = plan.ws.section("seg_scratch", torch.int64, 3300000 + 4 * 49 * 9)[3300000:].view(-1, 4).cpu().numpy()
    rows = rows[(rows[:, 2] >> 48) == 0x5A5A]
    t0, t1, row = rows[:, 0], rows[:, 1], (rows[:, 2] & 0xFFFF)
    base = t0.min()
    print(f"{name}: {rows.shape[0]} workgroups stamped; launch spans {(t1.max() - base) / 100:.2f} us")
    sched = (row == 0) & ((t1 - t0) > 100)
    print("  schedule workgroups: start", (t0[sched] - base) / 100, "end", (t1[sched] - base) / 100)
    f = row > 0
    print("  fill workgroups: start", np.percentile(t0[f] - base, [0, 50, 100]) / 100, "end", np.percentile(t1[f] - base, [0, 50, 100]) / 100)
finally:
    shutil.copy(backup, MAIN)
    os.remove(backup)
