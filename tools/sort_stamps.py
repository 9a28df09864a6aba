"""Timeline of the per-tile sort launch (GA_SORT_STAMPS build, tools/blend_variants.sh): per-wave (start, end, list length) stamps left in
the binning's depth array.  usage (GPU box): python tools/sort_stamps.py <variant name> [scene]"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")
name, scene = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "surface")
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
    g = synthetic.surface_surfels(100000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100000, seed=0)[0]
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512)
    plan.run(); plan.ensure_capacity()
    for _ in range(4):
        plan.run()
    torch.cuda.synchronize()
    rows = plan.ws.section("depth", torch.int64, 100000).view(-1, 4).cpu().numpy()
    rows = rows[(rows[:, 2] >> 48) == 0x5A5A]
    t0, t1, n = rows[:, 0], rows[:, 1], (rows[:, 2] & 0xFFFFFFFF).astype(np.int32)
    base = t0.min()
    print(f"{name} {scene}: {rows.shape[0]} waves stamped; launch spans {(t1.max() - base) / 100:.2f} us (100 MHz clock)")
    for lo, hi in ((-2, -2), (-1, -1), (0, 0), (1, 63), (64, 255), (256, 511), (512, 1023), (1024, 2047), (2048, 1 << 30)):
        sel = (n >= lo) & (n <= hi)
        if sel.any():
            print(f"  n in [{lo},{hi}]: {sel.sum():6d} waves  start {np.percentile(t0[sel] - base, [0, 50, 100]) / 100}  end {np.percentile(t1[sel] - base, [0, 50, 100]) / 100}"
                  f"  life {np.percentile((t1 - t0)[sel], [50, 100]) / 100} us")
finally:
    shutil.copy(backup, MAIN)
    os.remove(backup)
