#!/usr/bin/env python3
"""Runs an N-step Euler sampler (or, 4th argument "dopri5", the adaptive default sampler) of one DiT arch (for rocprofv3 --kernel-trace):
prints wall time per step / per function evaluation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd.dit import DiT_models
from gaussiananything_amd.transport import Sampler, create_transport
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "DiT-PixArt-PCD-CLAY-L"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(0)
model = DiT_models[arch](input_size=16, in_channels=3, context_dim=1024, pooling_ctx_dim=768, num_classes=0,
                         learn_sigma=False, roll_out=True)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p_ in model.parameters():
        if float(p_.abs().max()) == 0.0:
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
model.to(dev)
B, L, M = 2, 768, 1369
x = torch.randn(B, L, 3, generator=g).to(dev)
ctx = {"img_crossattn": torch.randn(B, M, 1024, generator=g).to(dev), "img_vector": torch.randn(B, 1024, generator=g).to(dev)}
ctx["img_crossattn"][1] = 0   # the unconditional half of a CFG batch (sgm force_uc_zero_embeddings)
ctx["img_vector"][1] = 0
sampler = Sampler(create_transport("GVP", "velocity", None, None, None, snr_type="uniform"))
method = sys.argv[3] if len(sys.argv) > 3 else "euler"
fn = sampler.sample_ode(sampling_method=method, num_steps=steps)
with torch.no_grad():
    fn(x, model.forward_with_cfg, context=ctx, cfg_scale=4.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(x, model.forward_with_cfg, context=ctx, cfg_scale=4.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
st = sampler.last_ode.last_stats
if method == "euler":
    print(f"{arch}: {steps} grid points, {dt / (steps - 1) * 1e3:.3f} ms per step wall")
else:
    print(f"{arch}: {method}, {steps} output times, {st} -> {dt / st['nfe'] * 1e3:.3f} ms per function evaluation wall ({dt:.4f} s)")
