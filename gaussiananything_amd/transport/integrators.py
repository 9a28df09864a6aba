"""``ode`` solver class with the reference's surface (/root/reference/transport/integrators.py:83-119)."""
from __future__ import annotations

import torch as th

from .odeint import odeint


class ode:
    """ODE solver class"""

    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.drift = drift
        self.t = th.linspace(t0, t1, num_steps)
        self.atol = atol
        self.rtol = rtol
        self.sampler_type = sampler_type
        self.last_stats = {}

    def sample(self, x, model, **model_kwargs):
        device = x.device

        def _fn(t, x):
            t = th.ones(x.size(0), device=device) * t
            return self.drift(x, t, model, **model_kwargs)

        self.last_stats = {}
        return odeint(_fn, x, self.t.to(device), method=self.sampler_type, atol=self.atol, rtol=self.rtol,
                      stats=self.last_stats)
