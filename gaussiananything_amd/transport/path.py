"""Interpolants x_t = alpha(t) x_1 + sigma(t) x_0 of the transport package and the quantities the samplers derive from them
(/root/reference/transport/path.py:18-192: ICPlan = linear, GVPCPlan, VPCPlan -- same class and method names, so that
``transport.path_sampler`` answers the same calls).  Host arithmetic on the ODE / SDE state, a handful of elementwise ops per step.

Everything follows from (alpha, alpha', sigma, sigma'):
    probability-flow drift of the data-prediction form   f(x, t) = (alpha'/alpha) x,   w(t) = (alpha'/alpha) sigma^2 - sigma sigma'
    score from a velocity                                 s = ((alpha/alpha') v - x) / (sigma^2 - (alpha/alpha') sigma' sigma)
"""
from __future__ import annotations

import math

import torch as th


def expand_t_like_x(t, x):
    """[B] time vector -> [B, 1, ..., 1] broadcastable against x"""
    return t.view(t.size(0), *([1] * (x.dim() - 1)))


class ICPlan:
    """linear interpolant: alpha = t, sigma = 1 - t"""

    def __init__(self, sigma=0.0):
        self.sigma = sigma

    def compute_alpha_t(self, t):
        return t, 1

    def compute_sigma_t(self, t):
        return 1 - t, -1

    def compute_d_alpha_alpha_ratio_t(self, t):
        return 1 / t

    def compute_drift(self, x, t):
        """(-f(x, t), w(t)) of the header"""
        t = expand_t_like_x(t, x)
        ratio = self.compute_d_alpha_alpha_ratio_t(t)
        s, ds = self.compute_sigma_t(t)
        return -(ratio * x), ratio * (s ** 2) - s * ds

    def compute_diffusion(self, x, t, form="constant", norm=1.0):
        t = expand_t_like_x(t, x)
        if form == "constant":
            return norm
        if form == "SBDM":
            return norm * self.compute_drift(x, t)[1]
        if form == "sigma":
            return norm * self.compute_sigma_t(t)[0]
        if form == "linear":
            return norm * (1 - t)
        if form == "decreasing":
            return 0.25 * (norm * th.cos(math.pi * t) + 1) ** 2
        if form == "inccreasing-decreasing":   # (the reference's spelling of the key)
            return norm * th.sin(math.pi * t) ** 2
        raise NotImplementedError(f"Diffusion form {form} not implemented")

    def _inverse_ratio(self, x, t):
        t = expand_t_like_x(t, x)
        a, da = self.compute_alpha_t(t)
        s, ds = self.compute_sigma_t(t)
        return a / da, s, ds

    def get_score_from_velocity(self, velocity, x, t):
        r, s, ds = self._inverse_ratio(x, t)
        return (r * velocity - x) / (s ** 2 - r * ds * s)

    def get_noise_from_velocity(self, velocity, x, t):
        r, s, ds = self._inverse_ratio(x, t)
        return (r * velocity - x) / (r * ds - s)

    def get_velocity_from_score(self, score, x, t):
        t = expand_t_like_x(t, x)
        drift, var = self.compute_drift(x, t)
        return var * score - drift

    def compute_mu_t(self, t, x0, x1):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[0] * x1 + self.compute_sigma_t(t)[0] * x0

    compute_xt = compute_mu_t

    def compute_ut(self, t, x0, x1, xt):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[1] * x1 + self.compute_sigma_t(t)[1] * x0

    def plan(self, t, x0, x1):
        xt = self.compute_xt(t, x0, x1)
        return t, xt, self.compute_ut(t, x0, x1, xt)


class GVPCPlan(ICPlan):
    """trigonometric ("generalised variance preserving") interpolant: alpha = sin(pi t / 2), sigma = cos(pi t / 2)"""

    def compute_alpha_t(self, t):
        return th.sin(t * math.pi / 2), math.pi / 2 * th.cos(t * math.pi / 2)

    def compute_sigma_t(self, t):
        return th.cos(t * math.pi / 2), -math.pi / 2 * th.sin(t * math.pi / 2)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return math.pi / (2 * th.tan(t * math.pi / 2))


class VPCPlan(ICPlan):
    """variance-preserving diffusion path with a linear beta schedule read backwards in t"""

    def __init__(self, sigma_min=0.1, sigma_max=20.0):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def log_mean_coeff(self, t):
        return -0.25 * ((1 - t) ** 2) * (self.sigma_max - self.sigma_min) - 0.5 * (1 - t) * self.sigma_min

    def d_log_mean_coeff(self, t):
        return 0.5 * (1 - t) * (self.sigma_max - self.sigma_min) + 0.5 * self.sigma_min

    def compute_alpha_t(self, t):
        a = th.exp(self.log_mean_coeff(t))
        return a, a * self.d_log_mean_coeff(t)

    def compute_sigma_t(self, t):
        p = 2 * self.log_mean_coeff(t)
        s = th.sqrt(1 - th.exp(p))
        return s, th.exp(p) * (2 * self.d_log_mean_coeff(t)) / (-2 * s)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return self.d_log_mean_coeff(t)

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        beta = self.sigma_min + (1 - t) * (self.sigma_max - self.sigma_min)
        return -0.5 * beta * x, beta / 2
