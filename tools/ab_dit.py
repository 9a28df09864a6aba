#!/usr/bin/env python3
"""Same-box A/B of the DiT evaluation time between two checkouts (each with its own built library): alternates them.
usage: tools/ab_dit.py <treeA> <treeB> [rounds]"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import torch, bench
    os.environ["GA_SKIP_SAMPLER"] = "1"
    for a in ("DiT-PixArt-PCD-CLAY-L", "DiT-PixArt-PCD-CLAY-B"):
        print(a[-6:], bench.bench_dit(torch.device("cuda:0"), a, 40, 5)["ms_per_nfe"], flush=True)
    sys.exit(0)
trees = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
for r in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for name, t in zip("AB", trees):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], cwd=t, capture_output=True, text=True)
        print(name, " ".join(out.stdout.split()) or out.stderr[-300:], flush=True)
