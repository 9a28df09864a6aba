"""Where the waves of the split blend spend their cycles (GA_SPLIT_PROFILE builds, tools/blend_variants.sh): per-wave section sums left in
the binning's depth array.  usage (GPU box): python tools/split_profile.py <variant name> [scene]"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")
name, scene = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "surface")
backup = MAIN + ".prof_backup"
shutil.copy(MAIN, backup)
try:
    shutil.copy(os.path.join(ROOT, "tools", "_build", f"libga_{name}.so"), MAIN)
    sys.path.insert(0, ROOT)
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
    rows = plan.ws.section("depth", torch.int64, (512 + 8 * 1024) * 4 * 8).view(-1, 8).cpu()
    rows = rows[(rows[:, 7] >> 48) == 0x5A5A]
    rows[:, 7] &= (1 << 48) - 1
    rows = rows.double()
    names = ["prologue", "staging duty", "stamp wait", "chunk table", "pair instructions", "composite passes", "trailing duties", "total"]
    tot = rows[:, 7].sum()
    print(f"{name} {scene}: {rows.shape[0]} waves, {tot / rows.shape[0]:.0f} cycles per wave")
    for q, nm in enumerate(names[:7]):
        print(f"  {nm:18s} {rows[:, q].sum() / tot * 100:5.1f} %   mean {rows[:, q].mean():8.0f}   max {rows[:, q].max():8.0f}")
finally:
    shutil.copy(backup, MAIN)
    os.remove(backup)
