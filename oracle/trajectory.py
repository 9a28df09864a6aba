"""CPU oracle of a whole SAMPLING TRAJECTORY (BASELINE.json configs[2]: "DiT-B/2 point-cloud-latent, 250-step SiT ODE sampler"):
the fp32 denoiser oracle (oracle/dit.py, pinned to the reference's own classes) integrated by the float64 integrator oracle
(oracle/ode.py, torchdiffeq semantics of SURVEY.md A.3) -- what /root/reference/transport/integrators.py:100-119 computes when
the model is /root/reference/dit/dit_i23d.py:1537-1546 and nothing runs in bf16.

TEST INFRASTRUCTURE ONLY: used by tests/test_dit_gpu.py and by bench.py's `parity` leg to measure how far the bf16 HIP trajectory
drifts from the all-fp32 one over a real-depth, many-step integration.  Never imported by the product package."""
import numpy as np
import torch

from . import dit as od
from . import ode as oo


def release_model(arch, in_channels, seed=0, redraw_seed=1):
    """A release-size denoiser with the seeded weights of SURVEY.md 8d 'Config #3 input': torch.manual_seed(seed) default
    initialisation, then every zero-initialised tensor re-drawn from N(0, 0.02).  Returns (module on the CPU, fp32 state dict)."""
    from gaussiananything_amd.dit import DiT_models
    torch.manual_seed(seed)
    model = DiT_models[arch](input_size=16, in_channels=in_channels, context_dim=1024, pooling_ctx_dim=768, num_classes=0,
                             learn_sigma=False, roll_out=True)
    g = torch.Generator().manual_seed(redraw_seed)
    with torch.no_grad():
        for p in model.parameters():
            if float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model, {k: v.detach().clone().float() for k, v in model.state_dict().items()}


def release_inputs(in_channels, cfg=True, stage2=False, seed=7, tokens=768, ctx_tokens=1369):
    """start state and conditioning at the release shapes: CFG batch [conditional | unconditional (zeros)] or the conditional
    sequence alone"""
    g = torch.Generator().manual_seed(seed)
    B = 2 if cfg else 1
    x1 = torch.randn(1, tokens, in_channels, generator=g)
    x = torch.cat([x1, x1], 0) if cfg else x1        # (FlowMatchingEngine.sample: both halves start from the same noise)
    ctx = {"img_crossattn": torch.randn(B, ctx_tokens, 1024, generator=g), "img_vector": torch.randn(B, 1024, generator=g)}
    if cfg:
        ctx["img_crossattn"][1:] = 0
        ctx["img_vector"][1:] = 0
    if stage2:
        xyz = ((torch.rand(1, tokens, 3, generator=g) - 0.5) * 0.9) / 0.45
        ctx["fps-xyz"] = xyz.expand(B, -1, -1).contiguous()
    return x, ctx


def integrate(sd, x0, ctx, cfg_scale, method, num_points, cfg=True, atol=1e-6, rtol=1e-3, threads=None, stats=None):
    """states at linspace(0, 1, num_points) of dx/dt = forward_with_cfg(x, t) (cfg) or forward(x, t), fp32 model, fp64 integrator"""
    if threads:
        torch.set_num_threads(int(threads))
    cctx = {k: v.float() for k, v in ctx.items()}
    B = x0.shape[0]

    def f(ts, yy):
        with torch.no_grad():
            tt = torch.full((B,), float(ts))
            xx = torch.from_numpy(np.asarray(yy)).float()
            v = od.forward_with_cfg(sd, xx, tt, cctx, cfg_scale) if cfg else od.dit_forward(sd, xx, tt, cctx)
        return v.double().numpy()

    tgrid = np.linspace(0.0, 1.0, num_points)
    return oo.odeint(f, x0.double().numpy(), tgrid, method=method, atol=atol, rtol=rtol, stats=stats)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
