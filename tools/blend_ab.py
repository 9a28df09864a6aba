"""Time the rasterizer forward of both bench scenes with alternative builds of the library (tools/blend_variants.sh).
usage (GPU box): python tools/blend_ab.py name1 name2 ...   ('main' = the in-tree library)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")
CODE = r'''
import sys, json, torch, time
sys.path.insert(0, %r)
from gaussiananything_amd import synthetic
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan
dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)
out = {}
for scene in ("surface", "stress"):
    g = synthetic.surface_surfels(100000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100000, seed=0)[0]
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    plan = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"].to(dev), cams["cam_view_proj"].to(dev), torch.ones(3, device=dev), 512, 512, flags=FLAGS)
    plan.run(); plan.ensure_capacity()
    for _ in range(10): plan.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): plan.run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    st = plan.ws.status().cpu().tolist()
    out[scene] = {"ms": round(ms, 4), "status": st[3:16]}
print(json.dumps(out))
'''


def main():
    names = sys.argv[1:] or ["main"]
    backup = MAIN + ".ab_backup"
    shutil.copy(MAIN, backup)
    try:
        for name in names:
            flags = 0
            if name.endswith("+stats"):
                name, flags = name[:-6], 1
            if name != "main":
                shutil.copy(os.path.join(ROOT, "tools", "_build", f"libga_{name}.so"), MAIN)
            else:
                shutil.copy(backup, MAIN)
            r = subprocess.run([sys.executable, "-c", (CODE % ROOT).replace("FLAGS", str(flags))], capture_output=True, text=True, timeout=60)
            print(name, "flags", flags, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + r.stderr[-800:]), flush=True)
    finally:
        shutil.copy(backup, MAIN)
        os.remove(backup)


if __name__ == "__main__":
    main()
