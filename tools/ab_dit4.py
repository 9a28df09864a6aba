#!/usr/bin/env python3
"""ms per DiT evaluation at the four shapes a released cascade runs (round 6), fresh process, one line:
L at CFG batch 2 | stage-2 L on the conditional sequence alone (batch 1) | L at CFG batch 4 (release stage 1: num_samples=2) | B at CFG batch 2.
usage (GPU box): [ENV=...] python tools/ab_dit4.py     (tools/ab_env.sh-style alternation: bash tools/ab_env4.sh A=1 A=0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
os.environ["GA_SKIP_SAMPLER"] = "1"
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = [bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-L", n, 5)["ms_per_nfe"],
     bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-stage2-L", n, 5, cond_only=True)["ms_per_nfe"],
     bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-L", n, 5, samples=2)["ms_per_nfe"],
     bench.bench_dit(dev, "DiT-PixArt-PCD-CLAY-B", n, 5)["ms_per_nfe"]]
print("L_cfg2 L_b1 L_cfg4 B_cfg2: " + " ".join(f"{v:.3f}" for v in r), flush=True)
