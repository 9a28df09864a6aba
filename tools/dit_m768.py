#!/usr/bin/env python3
"""The stage-2 evaluation as the cascade runs it: uc == c, so the denoiser sees the conditional sequence alone -- batch 1, every
GEMM at M = 768.  ms per evaluation (median of three runs); under rocprofv3 --kernel-trace --stats it gives the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd.dit import DiT_models

dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "DiT-PixArt-PCD-CLAY-stage2-L"
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
stage2 = "stage2" in arch
C = 10 if stage2 else 3
model = DiT_models[arch](input_size=16, in_channels=C, context_dim=1024, pooling_ctx_dim=768, num_classes=0, learn_sigma=False, roll_out=True)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p_ in model.parameters():
        if float(p_.abs().max()) == 0.0:
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
model.to(dev)
B, L, M = 1, 768, 1369
x = torch.randn(B, L, C, generator=g).to(dev)
ctx = {"img_crossattn": torch.randn(B, M, 1024, generator=g).to(dev), "img_vector": torch.randn(B, 1024, generator=g).to(dev)}
if stage2:
    ctx["fps-xyz"] = ((torch.rand(B, L, 3, generator=g) - 0.5) * 0.9).to(dev)
t = torch.full((B,), 0.5, device=dev)
with torch.no_grad():
    for _ in range(10):
        model(x, t, ctx)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.5:
        for _ in range(10):
            model(x, t, ctx)
        torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(nfe):
            model(x, t, ctx)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / nfe * 1e3)
print(f"{arch} batch 1: {sorted(reps)[1]:.4f} ms per evaluation  (runs {[round(r, 4) for r in reps]})")
