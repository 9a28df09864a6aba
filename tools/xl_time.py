"""ms per evaluation of DiT-PixArt-PCD-CLAY-XL (28 x 1152, 16 heads of 72: the head-dim-generic path) next to L, release shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
os.environ["GA_SKIP_SAMPLER"] = "1"
for arch in ("DiT-PixArt-PCD-CLAY-XL", "DiT-PixArt-PCD-CLAY-L"):
    r = bench.bench_dit(dev, arch, 10, 3)
    print(arch, r["ms_per_nfe"], flush=True)
