"""Does the rasterizer's latency-bound front-end hide under another forward's issue-bound blend?  Two independent forwards (own
Gaussians, workspaces, outputs) back to back on ONE stream against the same two on TWO streams; BASELINE configs[1], both scenes; also
4 views + 4 views of ONE Gaussian set (what a view-split inside ga_surfel_forward would run).  usage (GPU box): python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import synthetic
from gaussiananything_amd.diff_surfel_rasterization import SurfelForwardPlan

dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)

def plan(g, views):
    m, o, s, r, c = [t.to(dev) for t in synthetic.split_gaussians(g)]
    p = SurfelForwardPlan(m, o, c, s, r, cams["cam_view"][views].to(dev), cams["cam_view_proj"][views].to(dev), torch.ones(3, device=dev), 512, 512)
    p.run(); p.ensure_capacity()
    for _ in range(5): p.run()
    return p

def timeit(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for scene in ("surface", "stress"):
    mk = (lambda sd: synthetic.surface_surfels(100_000, seed=sd)[0]) if scene == "surface" else (lambda sd: synthetic.random_surfels(100_000, seed=sd)[0])
    all8 = list(range(8))
    a, b = plan(mk(1), all8), plan(mk(2), all8)
    one = timeit(lambda: a.run())
    def seq():
        a.run(); b.run()
    def par():
        with torch.cuda.stream(s1): a.run()
        with torch.cuda.stream(s2): b.run()
    g = mk(1)
    h1, h2 = plan(g, [0, 1, 2, 3]), plan(g, [4, 5, 6, 7])
    def halves():
        with torch.cuda.stream(s1): h1.run()
        with torch.cuda.stream(s2): h2.run()
    def halves_seq():
        h1.run(); h2.run()
    print(f"{scene}: one forward (8 views) {one:.4f} ms | two forwards, one stream {timeit(seq):.4f} | two streams {timeit(par):.4f} | "
          f"4 + 4 views of one set: one stream {timeit(halves_seq):.4f}, two streams {timeit(halves):.4f}", flush=True)
