#!/usr/bin/env python3
"""ms per DiT evaluation (L, B, L with four samples batched), fresh process: `child` prints one line (used by tools/ab_lib.sh)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, bench
    os.environ["GA_SKIP_SAMPLER"] = "1"
    dev = torch.device("cuda:0")
    r = [bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-L", 30, 5)["ms_per_nfe"], bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-B", 30, 5)["ms_per_nfe"],
         bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-L", 20, 3, samples=4)["ms_per_nfe"]]
    print(" ".join(f"{v:.3f}" for v in r), flush=True)
    sys.exit(0)
