"""``Transport`` / ``Sampler`` / ``create_transport`` with the reference's surface for the probability-flow ODE path
(/root/reference/transport/transport.py:45-112, 193-242, 322-431 and transport/__init__.py:4-72).  The released models are
velocity predictors on the GVP path (sgm/configs/stage2-i23d.yaml), for which the drift is the model output itself and the
integration interval is [0, 1]; score / noise parametrisations and the SDE samplers are outside this tier and raise."""
from __future__ import annotations

import enum

import torch as th

from .integrators import ode


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
        if self.model_type != ModelType.VELOCITY:
            raise NotImplementedError("only velocity-prediction models are on the GaussianAnything sampling path")

        def body_fn(x, t, model, **model_kwargs):
            model_output = model(x, t, **model_kwargs)
            assert model_output.shape == x.shape, "Output shape from ODE solver must match input shape"
            return model_output

        return body_fn


class Sampler:
    """Sampler class for the transport model"""

    def __init__(self, transport, guider_config=None):
        self.transport = transport
        self.drift = self.transport.get_drift()

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
                            rtol=rtol, plain_velocity_drift=not reverse)
        return self.last_ode.sample

    def sample_sde(self, *a, **k):
        raise NotImplementedError("SDE sampling is not on the released inference path (sample() uses sample_ode)")


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
