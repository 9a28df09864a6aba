"""``Transport`` / ``Sampler`` / ``create_transport`` with the reference's surface (/root/reference/transport/transport.py:45-112,
193-242, 255-431 and transport/__init__.py:4-72).  The released models are velocity predictors on the GVP path
(sgm/configs/stage2-i23d.yaml), for which the drift is the model output itself and the integration interval is [0, 1] -- the path
the HIP denoiser's fused sampler step serves; the score / noise parametrisations and the SDE samplers (Euler-Maruyama, Heun; last
step Mean / Tweedie / Euler) are host arithmetic around the same model call."""
from __future__ import annotations

import enum

import torch as th

from . import path
from .integrators import ode, sde


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


class SNRType(enum.Enum):
    UNIFORM = enum.auto()
    LOGNORM = enum.auto()


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps, snr_type):
        self.loss_type = loss_type
        self.model_type = model_type
        self.path_type = path_type
        self.train_eps = train_eps
        self.sample_eps = sample_eps
        self.snr_type = snr_type
        self.path_sampler = {PathType.LINEAR: path.ICPlan, PathType.GVP: path.GVPCPlan, PathType.VP: path.VPCPlan}[path_type]()

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False,
                       last_step_size=0.0):
        t0, t1 = 0, 1
        eps = train_eps if not eval else sample_eps
        if self.path_type == PathType.VP:
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        elif self.path_type in (PathType.LINEAR, PathType.GVP) and (self.model_type != ModelType.VELOCITY or sde):
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.model_type != ModelType.VELOCITY else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1

    def get_drift(self):
        """drift of the probability-flow ODE for the model's parametrisation: the output itself for a velocity model,
        -f + w * score for a score model, the same with score = -noise / sigma for a noise model"""
        plan = self.path_sampler

        def from_score(x, t, score):
            mean, var = plan.compute_drift(x, t)
            return var * score - mean

        if self.model_type == ModelType.VELOCITY:
            drift_fn = lambda x, t, model, **kw: model(x, t, **kw)                                                   # noqa: E731
        elif self.model_type == ModelType.SCORE:
            drift_fn = lambda x, t, model, **kw: from_score(x, t, model(x, t, **kw))                                 # noqa: E731
        else:
            drift_fn = lambda x, t, model, **kw: from_score(                                                         # noqa: E731
                x, t, model(x, t, **kw) / -plan.compute_sigma_t(path.expand_t_like_x(t, x))[0])

        def body_fn(x, t, model, **model_kwargs):
            model_output = drift_fn(x, t, model, **model_kwargs)
            assert model_output.shape == x.shape, "Output shape from ODE solver must match input shape"
            return model_output

        return body_fn

    def get_score(self):
        """score of x_t = alpha_t x + sigma_t eps from the model output"""
        plan = self.path_sampler
        if self.model_type == ModelType.NOISE:
            return lambda x, t, model, **kw: model(x, t, **kw) / -plan.compute_sigma_t(path.expand_t_like_x(t, x))[0]
        if self.model_type == ModelType.SCORE:
            return lambda x, t, model, **kw: model(x, t, **kw)
        return lambda x, t, model, **kw: plan.get_score_from_velocity(model(x, t, **kw), x, t)


class Sampler:
    """Sampler class for the transport model"""

    def __init__(self, transport, guider_config=None):
        self.transport = transport
        self.drift = self.transport.get_drift()
        self.score = self.transport.get_score()

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False, cfg=False):
        """returns ``fn(x, model, **model_kwargs) -> Tensor[num_steps, *x.shape]``; for fixed solvers ``num_steps`` grid
        points (num_steps - 1 function evaluations with euler), for dopri5 the number of interpolated outputs."""
        if reverse:
            drift = lambda x, t, model, **kwargs: self.drift(x, th.ones_like(t) * (1 - t), model, **kwargs)  # noqa: E731
        else:
            drift = self.drift
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True,
                                               reverse=reverse, last_step_size=0.0)
        self.last_ode = ode(drift=drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol,
                            rtol=rtol, plain_velocity_drift=not reverse and self.transport.model_type == ModelType.VELOCITY)
        return self.last_ode.sample

    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean",
                   last_step_size=0.04, num_steps=250):
        """returns ``fn(init, model, **model_kwargs) -> list of num_steps states``: num_steps - 1 stochastic steps on
        linspace(t0, t1, num_steps) plus the closing step (None: repeat, "Mean": one drift step of last_step_size,
        "Tweedie": denoise with the score, "Euler": one probability-flow step)."""
        if last_step is None:
            last_step_size = 0.0
        plan = self.transport.path_sampler

        def diffusion_fn(x, t):
            return plan.compute_diffusion(x, t, form=diffusion_form, norm=diffusion_norm)

        def sde_drift(x, t, model, **kw):
            return self.drift(x, t, model, **kw) + diffusion_fn(x, t) * self.score(x, t, model, **kw)

        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, diffusion_form=diffusion_form,
                                               sde=True, eval=True, reverse=False, last_step_size=last_step_size)
        solver = sde(sde_drift, diffusion_fn, t0=t0, t1=t1, num_steps=num_steps, sampler_type=sampling_method)

        if last_step is None:
            closing = lambda x, t, model, **kw: x                                                                   # noqa: E731
        elif last_step == "Mean":
            closing = lambda x, t, model, **kw: x + sde_drift(x, t, model, **kw) * last_step_size                   # noqa: E731
        elif last_step == "Tweedie":
            def closing(x, t, model, **kw):
                a, sg = plan.compute_alpha_t(t)[0][0], plan.compute_sigma_t(t)[0][0]
                return x / a + (sg ** 2) / a * self.score(x, t, model, **kw)
        elif last_step == "Euler":
            closing = lambda x, t, model, **kw: x + self.drift(x, t, model, **kw) * last_step_size                  # noqa: E731
        else:
            raise NotImplementedError()

        def _sample(init, model, **model_kwargs):
            xs = solver.sample(init, model, **model_kwargs)
            ts = th.ones(init.size(0), device=init.device) * t1
            xs.append(closing(xs[-1], ts, model, **model_kwargs))
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs

        return _sample


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     snr_type="uniform"):
    model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE}.get(prediction, ModelType.VELOCITY)
    loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(loss_weight, WeightType.NONE)
    if snr_type == "lognorm":
        snr = SNRType.LOGNORM
    elif snr_type == "uniform":
        snr = SNRType.UNIFORM
    else:
        raise ValueError(f"Invalid snr type {snr_type}")
    path = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    if path == PathType.VP:
        train_eps = 1e-5 if train_eps is None else train_eps
        sample_eps = 1e-3 if sample_eps is None else sample_eps
    elif path in (PathType.GVP, PathType.LINEAR) and model_type != ModelType.VELOCITY:
        train_eps = 1e-3 if train_eps is None else train_eps
        sample_eps = 1e-3 if sample_eps is None else sample_eps
    else:  # velocity & [GVP, LINEAR] is stable everywhere
        train_eps = 0
        sample_eps = 0
    return Transport(model_type=model_type, path_type=path, loss_type=loss_type, train_eps=train_eps,
                     sample_eps=sample_eps, snr_type=snr)
