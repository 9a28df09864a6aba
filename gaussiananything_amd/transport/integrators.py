"""``ode`` solver class with the reference's surface (/root/reference/transport/integrators.py:83-119)."""
from __future__ import annotations

import os

import torch as th

from .odeint import odeint


class ode:
    """ODE solver class"""

    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol, plain_velocity_drift=False):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.drift = drift
        self.t = th.linspace(t0, t1, num_steps)
        self.atol = atol
        self.rtol = rtol
        self.sampler_type = sampler_type
        self.last_stats = {}
        self.plain_velocity_drift = plain_velocity_drift   # drift(x, t, model) == model(x, t): no wrapper arithmetic

    def sample(self, x, model, **model_kwargs):
        device = x.device
        # A denoiser of this package offers the whole fixed-grid Euler loop on the device (DiT.sample_euler_fused): taken
        # when `model` is its forward_with_cfg / forward_cond and the drift adds nothing.  GA_ODE_GRAPH=0 keeps the eager
        # loop (the parity tests compare the two bit by bit).
        owner, name = getattr(model, "__self__", None), getattr(model, "__name__", "")
        if (self.sampler_type == "euler" and self.plain_velocity_drift and device.type == "cuda" and len(self.t) > 4
                and name in ("forward_with_cfg", "forward_cond") and hasattr(owner, "sample_euler_fused")
                and "context" in model_kwargs and set(model_kwargs) <= {"context", "cfg_scale"}
                and os.environ.get("GA_ODE_GRAPH", "1") != "0"):
            out = owner.sample_euler_fused(x, self.t.tolist(), model_kwargs["context"],
                                           cfg_scale=model_kwargs.get("cfg_scale", 1.0), cfg=(name == "forward_with_cfg"))
            self.last_stats = {"nfe": len(self.t) - 1, "steps": len(self.t) - 1, "rejected": 0, "graph": True, "fused": True}
            return out

        def _fn(t, x):
            t = th.ones(x.size(0), device=device) * t
            return self.drift(x, t, model, **model_kwargs)

        self.last_stats = {}
        return odeint(_fn, x, self.t.to(device), method=self.sampler_type, atol=self.atol, rtol=self.rtol,
                      stats=self.last_stats)
