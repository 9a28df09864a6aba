"""CPU oracle of the ODE integrators (numpy, float64 state): an independent restatement of the torchdiffeq semantics the
reference relies on at /root/reference/transport/integrators.py:111-118.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for the third-party arithmetic: torchdiffeq is absent from /root/reference and
unpinned in requirements.txt:40; the reference holds no test for it.  What IS pinned by tests/test_transport.py: the
reference's own transport plumbing (check_interval, velocity drift, time grid) -- the reference's transport/*.py is
imported with this module standing in for torchdiffeq -- and the analytic solutions of linear ODEs.  The Dormand-Prince stage
matrix, nodes and 5th-order weights are checked against SciPy's RK45 (an independent implementation of the same pair), one step
against its rk_step, and the error weights against the order conditions (tests/test_cpu_oracle_and_host.py::
test_ode_oracle_tableau_against_scipy_rk45); torchdiffeq's embedded 4th-order weights and its step-size controller are restated
from the published code and remain unpinned."""
import numpy as np

A = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
B = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
     [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
     [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
     [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
E = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
     11 / 84 - 649 / 6300, -1 / 60]
MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
       187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def rms(x):
    return float(np.sqrt(np.mean(np.square(x))))


def odeint(func, y0, t, method="dopri5", atol=1e-6, rtol=1e-3, stats=None):
    y = np.asarray(y0, np.float64)
    t = [float(v) for v in t]
    out = [y.copy()]
    nfe = [0]

    def f(ts, yy):
        nfe[0] += 1
        return np.asarray(func(ts, yy), np.float64)

    if method != "dopri5":
        for t0, t1 in zip(t[:-1], t[1:]):
            dt = t1 - t0
            if method == "euler":
                dy = dt * f(t0, y)
            elif method == "midpoint":
                dy = dt * f(t0 + dt / 2, y + dt / 2 * f(t0, y))
            elif method == "heun2":
                k1 = f(t0, y); dy = dt / 2 * (k1 + f(t1, y + dt * k1))
            elif method == "heun3":
                k1 = f(t0, y); k2 = f(t0 + dt / 3, y + dt / 3 * k1); k3 = f(t0 + 2 * dt / 3, y + 2 * dt / 3 * k2)
                dy = dt * (k1 / 4 + 3 * k3 / 4)
            elif method == "rk4":
                k1 = f(t0, y); k2 = f(t0 + dt / 3, y + dt / 3 * k1); k3 = f(t0 + 2 * dt / 3, y + dt * (k2 - k1 / 3))
                k4 = f(t1, y + dt * (k1 - k2 + k3)); dy = dt / 8 * (k1 + 3 * (k2 + k3) + k4)
            else:
                raise ValueError(method)
            y = y + dy
            out.append(y.copy())
        if stats is not None:
            stats.update(nfe=nfe[0], steps=len(t) - 1, rejected=0)
        return np.stack(out)

    t0 = t[0]
    f0 = f(t0, y)
    scale = atol + np.abs(y) * rtol
    d0, d1 = rms(y / scale), rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(t0 + h0, y + h0 * f0)
    d2 = rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
    dt = min(100 * h0, h1)
    seg = None
    steps = rejected = 0
    for tj in t[1:]:
        while seg is None or tj > seg[1]:
            k = [f0]
            for i in range(6):
                yi = y + dt * sum(c * kk for c, kk in zip(B[i], k))
                k.append(f(t0 + A[i] * dt, yi))
            y1 = yi
            err = dt * sum(c * kk for c, kk in zip(E, k))
            ratio = rms(err / (atol + rtol * np.maximum(np.abs(y), np.abs(y1))))
            steps += 1
            if ratio <= 1:
                ymid = y + dt * sum(c * kk for c, kk in zip(MID, k))
                seg = (t0, t0 + dt, y, y1, ymid, k[0], k[6], dt)
                t0, y, f0 = t0 + dt, y1, k[6]
            else:
                rejected += 1
            if ratio == 0:
                fac = 10.0
            else:
                fac = min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1 else 0.2))
            dt = dt * fac
        ta, tb, ya, yb, ym, fa, fb, h = seg
        a = 2 * h * (fb - fa) - 8 * (yb + ya) + 16 * ym
        b = h * (5 * fa - 3 * fb) + 18 * ya + 14 * yb - 32 * ym
        c = h * (fb - 4 * fa) - 11 * ya - 5 * yb + 16 * ym
        d = h * fa
        x = (tj - ta) / (tb - ta)
        out.append(ya + x * (d + x * (c + x * (b + x * a))))
    if stats is not None:
        stats.update(nfe=nfe[0], steps=steps, rejected=rejected)
    return np.stack(out)
