#!/usr/bin/env python3
"""ms per DiT-L evaluation with launch classes dropped (tools/dit_ablate.sh build; results are wrong by construction).
Trace durations over-count short dependent kernels; this measures what each class costs in WALL time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from gaussiananything_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libga_dit_ablate.so")
    import torch, bench
    os.environ["GA_SKIP_SAMPLER"] = "1"
    r = bench.bench_dit(torch.device("cuda:0"), "DiT-PixArt-PCD-CLAY-L", 30, 5)
    print(f"GA_DIT_SKIP={os.environ.get('GA_DIT_SKIP', '0'):>3s}  {r['ms_per_nfe']:.3f} ms")
    sys.exit(0)
NAMES = {0: "full", 1: "no self-attention", 2: "no cross-attention", 4: "no RMSNorm launches", 8: "no fc1 + fc2", 16: "no qkv + proj",
         32: "no CA q + out", 3: "no attention at all", 56: "no GEMMs", 63: "launch skeleton only (embed, conditioning, final)"}
for mask, name in NAMES.items():
    env = dict(os.environ, GA_DIT_SKIP=str(mask))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(f"{name:52s} {out[-1] if out else 'failed'}", flush=True)
