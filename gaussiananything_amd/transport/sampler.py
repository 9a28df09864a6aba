"""The sampling entry of the denoise half, cut down to what the released models use, plus the hook that gives the REFERENCE'S OWN
``transport`` package the device-resident integrators.

The reference's ``transport/*.py`` (path plans, score / noise parametrisations, SDE samplers) is device-agnostic host Python and is
not rebuilt here: a deployment keeps importing it and calls ``bind_reference_transport(transport.integrators)`` once
(INTEGRATION.md section 3).  After that ``transport.integrators.ode.sample`` (/root/reference/transport/integrators.py:100-119)
hands fixed-grid Euler and dopri5 integrations of a HIP denoiser to this package instead of torchdiffeq; everything else of the
reference's transport code runs unchanged.

For ``bench.py`` / the tests / ``cascade.sample`` -- which must run on a GPU box that has no reference tree -- this module carries the
one configuration the release uses (sgm/configs/stage2-i23d.yaml: velocity prediction on the GVP path; the drift of the
probability-flow ODE is then the model output itself and the interval is [0, 1], /root/reference/transport/transport.py:85-112,
209-218): ``Sampler(create_transport("GVP", "velocity", ...)).sample_ode(...)`` with the reference's call signature."""
from __future__ import annotations

import os

import torch

from .odeint import odeint


def integrate(model, x, t_grid, method, atol, rtol, model_kwargs, stats, drift=None):
    """dx/dt = model(x, t 1_B, **model_kwargs) -- or ``drift(x, t 1_B, model, **model_kwargs)`` when the caller's parametrisation
    wraps the model call -- over ``t_grid``; returns the states at every grid time.  A HIP denoiser's
    ``forward_with_cfg`` / ``forward_cond`` gets the whole loop on the device: ``sample_euler_fused`` (fixed-grid Euler, one
    captured step replayed) or ``sample_dopri5_device`` (adaptive steps decided on the device); GA_ODE_GRAPH=0 keeps the eager
    loops (the parity tests compare the two)."""
    owner, name = getattr(model, "__self__", None), getattr(model, "__name__", "")
    fusable = (drift is None and x.device.type == "cuda" and len(t_grid) > 4 and name in ("forward_with_cfg", "forward_cond")
               and "context" in model_kwargs and set(model_kwargs) <= {"context", "cfg_scale"}
               and os.environ.get("GA_ODE_GRAPH", "1") != "0"
               # a sigma-predicting model (out_channels != in_channels) has no fused step: its output is not a velocity of the state's
               # shape -- the eager loop fails on the shapes as the reference's body_fn assert does (transport/transport.py:219-224)
               and getattr(owner, "out_channels", None) == getattr(owner, "in_channels", None))
    if fusable and method == "euler" and hasattr(owner, "sample_euler_fused"):
        out = owner.sample_euler_fused(x, t_grid.tolist(), model_kwargs["context"], cfg_scale=model_kwargs.get("cfg_scale", 1.0),
                                       cfg=(name == "forward_with_cfg"))
        stats.update(nfe=len(t_grid) - 1, steps=len(t_grid) - 1, rejected=0, graph=True, fused=True)
        return out
    if fusable and method == "dopri5" and hasattr(owner, "sample_dopri5_device"):
        return owner.sample_dopri5_device(x, t_grid.tolist(), model_kwargs["context"], cfg_scale=model_kwargs.get("cfg_scale", 1.0),
                                          cfg=(name == "forward_with_cfg"), atol=atol, rtol=rtol, stats=stats)

    def rhs(t, y):
        tv = torch.ones(y.size(0), device=y.device) * t
        return model(y, tv, **model_kwargs) if drift is None else drift(y, tv, model, **model_kwargs)

    return odeint(rhs, x, t_grid.to(x.device), method=method, atol=atol, rtol=rtol, stats=stats)


class VelocityTransport:
    """velocity prediction on a GVP / linear path: nothing to convert, the ODE runs over [0, 1]"""
    train_eps = 0
    sample_eps = 0

    def check_interval(self, *_, **kw):
        if kw.get("sde"):
            raise NotImplementedError("SDE sampling: use the reference's transport package (INTEGRATION.md section 3)")
        return (1, 0) if kw.get("reverse") else (0, 1)


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None, snr_type="uniform"):
    if prediction not in ("velocity", None) or path_type not in ("GVP", "Linear"):
        raise NotImplementedError(f"{prediction} prediction on the {path_type} path is served by the reference's own transport "
                                  "package with bind_reference_transport (INTEGRATION.md section 3)")
    return VelocityTransport()


class Sampler:
    def __init__(self, transport, guider_config=None):
        self.transport = transport
        self.last_ode = None

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False, cfg=False):
        """-> ``fn(x, model, **model_kwargs) -> Tensor[num_steps, *x.shape]`` (the caller takes ``[-1]``)"""
        if reverse:
            raise NotImplementedError("reverse-time integration: use the reference's transport package")
        t0, t1 = self.transport.check_interval(0, 0, sde=False, eval=True, reverse=False)
        run = _Run(torch.linspace(t0, t1, num_steps), sampling_method, atol, rtol)
        self.last_ode = run
        return run.sample


class _Run:
    def __init__(self, t, method, atol, rtol):
        self.t, self.method, self.atol, self.rtol, self.last_stats = t, method, atol, rtol, {}

    def sample(self, x, model, **model_kwargs):
        self.last_stats = {}
        return integrate(model, x, self.t, self.method, self.atol, self.rtol, model_kwargs, self.last_stats)


def _is_plain_velocity(drift):
    """the reference's ``Transport.get_drift`` (transport/transport.py:193-225) returns ``body_fn`` closing over ``velocity_ode`` for a
    velocity model: the drift is then the model call itself"""
    cells = getattr(drift, "__closure__", None) or ()
    inner = {getattr(c.cell_contents, "__name__", "") for c in cells}
    return getattr(drift, "__name__", "") == "body_fn" and "velocity_ode" in inner


def bind_reference_transport(ref_integrators):
    """Patch the reference's ``transport.integrators.ode`` in place: ``ode.sample`` integrates with this package instead of
    torchdiffeq -- the fused device loops when the drift is the plain velocity call of a HIP denoiser, the generic device-resident
    integrators (through the reference's own drift function) for every other parametrisation.  Tuple states (the likelihood path)
    stay with the reference's method."""
    ref_ode = ref_integrators.ode
    theirs = ref_ode.sample

    def sample(self, x, model, **model_kwargs):
        if isinstance(x, tuple):
            return theirs(self, x, model, **model_kwargs)
        self.last_stats = {}
        return integrate(model, x, self.t, self.sampler_type, self.atol, self.rtol, model_kwargs, self.last_stats,
                         drift=None if _is_plain_velocity(self.drift) else self.drift)

    ref_ode.sample = sample
    return ref_ode
