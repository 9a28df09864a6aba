"""ODE integrators behind ``transport.integrators.ode`` -- the arithmetic the reference obtains from the third-party
``torchdiffeq.odeint`` (call site /root/reference/transport/integrators.py:111-118; semantics restated from the public
project; torchdiffeq is unpinned in /root/reference/requirements.txt:40).

``odeint(func, y0, t, method=..., atol, rtol)`` returns the states at every requested time, ``[len(t), *y0.shape]``.
The state lives on the device of ``y0`` and is advanced in fp32 (the reference's state starts in bf16 under AMP and is
promoted to fp32 by the first update, because the model returns fp32: dit_i23d.py:565).

* fixed grid (``euler``, ``midpoint``, ``heun2``, ``heun3``, ``rk4``): the grid is exactly ``t``; no host synchronisation at
  all, so a whole sampling loop can be enqueued (or graph-captured) without the host waiting on the device;
* ``dopri5``: Dormand-Prince 5(4), FSAL, torchdiffeq's step-size controller (RMS norm over the WHOLE state tensor, so step
  sizes are coupled across the batch -- SURVEY.md section 7 hard part 5), 4th-order dense output.  The accept/reject
  decision needs one scalar on the host per attempted step, as in torchdiffeq.
"""
from __future__ import annotations

import math
import os
import warnings

import torch

FIXED = ("euler", "midpoint", "heun2", "heun3", "rk4")


def _rk_fixed(method, func, t0, dt, t1, y0):
    if method == "euler":
        return dt * func(t0, y0)
    if method == "midpoint":
        k1 = func(t0, y0)
        return dt * func(t0 + 0.5 * dt, y0 + (0.5 * dt) * k1)
    if method == "heun2":
        k1 = func(t0, y0)
        k2 = func(t1, y0 + dt * k1)
        return (0.5 * dt) * (k1 + k2)
    if method == "heun3":
        k1 = func(t0, y0)
        k2 = func(t0 + dt / 3, y0 + (dt / 3) * k1)
        k3 = func(t0 + dt * 2 / 3, y0 + (dt * 2 / 3) * k2)
        return dt * (0.25 * k1 + 0.75 * k3)
    if method == "rk4":  # torchdiffeq's default 3/8-rule variant
        k1 = func(t0, y0)
        k2 = func(t0 + dt / 3, y0 + (dt / 3) * k1)
        k3 = func(t0 + dt * 2 / 3, y0 + dt * (k2 - k1 / 3))
        k4 = func(t1, y0 + dt * (k1 - k2 + k3))
        return dt * 0.125 * (k1 + 3 * (k2 + k3) + k4)
    raise ValueError(method)


# Dormand-Prince tableau (torchdiffeq dopri5.py)
_A = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_B = [[1 / 5],
      [3 / 40, 9 / 40],
      [44 / 45, -56 / 15, 32 / 9],
      [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
      [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
      [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
          -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
          187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


_EVALS = {"euler": 1, "midpoint": 2, "heun2": 2, "heun3": 3, "rk4": 4}


def _fixed_grid_graph(method, func, y, tt, out):
    """Fixed-grid integration with ONE step captured into a HIP graph and replayed per grid interval.

    A function evaluation of the DiT is ~290 kernel launches; enqueued from Python/ctypes one by one the GPU idles ~9 % of
    the time between them (rocprofv3: 4.83 ms of kernels in a 5.29 ms step).  The step reads its time and step size from
    two device scalars that are refreshed (device-to-device, asynchronously) before every replay, so nothing is baked
    into the graph; the state is advanced in place.  For ``euler`` the arithmetic is identical to the eager loop; the
    multi-stage methods form their intermediate times in fp32 on the device instead of fp64 on the host."""
    dev = y.device
    t_dev = torch.tensor(tt[:-1], dtype=torch.float32, device=dev)
    dt_dev = torch.tensor([b - a for a, b in zip(tt[:-1], tt[1:])], dtype=torch.float32, device=dev)
    t_cur, dt_cur = torch.zeros((), device=dev), torch.zeros((), device=dev)
    y_st = y.clone()

    def ff(ts, yy):
        return func(ts, yy).float()

    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):  # warm-up outside the capture: lazy initialisation, workspace sizing, K/V caches
        t_cur.copy_(t_dev[0])
        dt_cur.copy_(dt_dev[0])
        _rk_fixed(method, ff, t_cur, dt_cur, t_cur + dt_cur, y_st)
    cur.wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y_st.add_(_rk_fixed(method, ff, t_cur, dt_cur, t_cur + dt_cur, y_st))
    for j in range(1, len(tt)):
        t_cur.copy_(t_dev[j - 1])
        dt_cur.copy_(dt_dev[j - 1])
        graph.replay()
        out[j].copy_(y_st)


def _rms(x):
    return float(torch.sqrt(torch.mean(x.float() ** 2)))


def _initial_step(func, t0, y0, f0, rtol, atol, order=4):
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y1 = y0 + h0 * f0
    f1 = func(t0 + h0, y1)
    d2 = _rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
    return min(100 * h0, h1)


def _optimal_step(last, ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    if ratio == 0:
        return last * ifactor
    if ratio < 1:
        dfactor = 1.0
    factor = min(ifactor, max(safety / ratio ** (1.0 / order), dfactor))
    return last * factor


def _interp_coeffs(y0, y1, ymid, f0, f1, dt):
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * ymid
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * ymid
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * ymid
    d = dt * f0
    return a, b, c, d, y0


def odeint(func, y0, t, method="dopri5", atol=1e-6, rtol=1e-3, stats=None, max_steps=1 << 20):
    """``func(t_scalar_tensor, y) -> dy/dt``.  ``t``: 1-D increasing tensor.  ``stats``: optional dict, receives nfe / steps."""
    if isinstance(atol, (list, tuple)):
        atol = atol[0]
    if isinstance(rtol, (list, tuple)):
        rtol = rtol[0]
    dev = y0.device
    y = y0.float()
    tt = [float(v) for v in t.detach().cpu().double()]  # time is kept in fp64 on the host
    out = torch.empty((len(tt),) + tuple(y.shape), dtype=torch.float32, device=dev)
    out[0] = y
    nfe = [0]

    def f(ts, yy):
        nfe[0] += 1
        return func(torch.tensor(ts, dtype=torch.float32, device=dev), yy).float()

    if method in FIXED:
        if dev.type == "cuda" and len(tt) > 4 and os.environ.get("GA_ODE_GRAPH", "1") != "0":
            try:
                _fixed_grid_graph(method, func, y, tt, out)
                if stats is not None:
                    stats.update(nfe=(len(tt) - 1) * _EVALS[method], steps=len(tt) - 1, rejected=0, graph=True)
                return out
            except Exception as e:  # capture refused (a host synchronisation inside ``func``, ...): eager loop below
                if os.environ.get("GA_ODE_GRAPH") == "1":
                    raise
                warnings.warn(f"ODE step could not be captured into a HIP graph ({type(e).__name__}: {e}); running eagerly")
                y = y0.float()
                out[0] = y
        for j in range(1, len(tt)):
            t0, t1 = tt[j - 1], tt[j]
            y = y + _rk_fixed(method, f, t0, t1 - t0, t1, y)
            out[j] = y
        if stats is not None:
            stats.update(nfe=nfe[0], steps=len(tt) - 1, rejected=0)
        return out
    if method != "dopri5":
        raise ValueError(f"unsupported ODE method {method!r} (have {FIXED + ('dopri5',)})")

    t0 = tt[0]
    f0 = f(t0, y)
    dt = _initial_step(f, t0, y, f0, rtol, atol)
    interp = None        # coefficients of the last accepted step and its interval
    j, steps, rejected = 1, 0, 0
    while j < len(tt):
        # advance until the next output time is inside the last accepted step
        while interp is None or tt[j] > interp[1]:
            k = [f0]
            for i in range(6):
                yi = y
                for c, kk in zip(_B[i], k):
                    if c != 0:
                        yi = yi + (dt * c) * kk
                k.append(f(t0 + _A[i] * dt, yi))
            y1 = yi  # the 7th stage is evaluated AT the 5th-order solution (FSAL)
            err = sum((dt * c) * kk for c, kk in zip(_C_ERR, k) if c != 0)
            tol = atol + rtol * torch.maximum(y.abs(), y1.abs())
            ratio = _rms(err / tol)
            steps += 1
            # torchdiffeq asserts on a non-finite or underflowing step; without this a NaN model output would make
            # `ratio <= 1` False forever while _optimal_step keeps growing dt
            if not math.isfinite(ratio):
                raise FloatingPointError(f"dopri5: non-finite error ratio at t = {t0} (dt = {dt}): the model returned NaN/inf")
            if not math.isfinite(dt) or t0 + dt == t0:
                raise FloatingPointError(f"dopri5: step size underflow at t = {t0} (dt = {dt})")
            if steps > max_steps:
                raise RuntimeError(f"dopri5: more than {max_steps} attempted steps")
            if ratio <= 1:
                ymid = y + sum((dt * c) * kk for c, kk in zip(_C_MID, k) if c != 0)
                interp = (t0, t0 + dt, _interp_coeffs(y, y1, ymid, k[0], k[6], dt))
                t0, y, f0 = t0 + dt, y1, k[6]
            else:
                rejected += 1
            dt = _optimal_step(dt, ratio)
        ta, tb, (a, b, c, d, e) = interp
        x = (tt[j] - ta) / (tb - ta)
        out[j] = e + x * (d + x * (c + x * (b + x * a)))
        j += 1
    if stats is not None:
        stats.update(nfe=nfe[0], steps=steps, rejected=rejected)
    return out
