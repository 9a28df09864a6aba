"""Design aid: trip-count model of lanes = pixels blend schedules on the real tile lists of the bench scene
(tools/blend_sim.c).  Usage: python tools/blend_sim.py [surface|stress] [views]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussiananything_amd import synthetic  # noqa: E402
from tests import _util  # noqa: E402


class SimOut(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("pairs", "slots", "trips", "chunks", "waves", "max_wave_cost", "total_cost", "coupled_cost")]


def cull_extents(o, c2_rel=1.02, c2_abs=0.05, e_rel=1.01, e_abs=0.5):
    """numpy restatement of the cull half-extents of surfel_preprocess.hip (fp16, rounded up)."""
    T = o["trans"].astype(np.float32)
    Tu, Tv, Tw = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    opa = o["normal_opacity"][:, 3]
    cx, cy = o["xy"][:, 0], o["xy"][:, 1]
    with np.errstate(all="ignore"):
        c2 = (2.0 * np.log(255.0 * opa)) * c2_rel + c2_abs
        dd = (c2 * Tw[:, 0] ** 2 + c2 * Tw[:, 1] ** 2) - Tw[:, 2] ** 2
        iv = 1.0 / dd
        g0, g2 = iv * c2, -iv
        bx = g0 * Tu[:, 0] * Tw[:, 0] + g0 * Tu[:, 1] * Tw[:, 1] + g2 * Tu[:, 2] * Tw[:, 2]
        by = g0 * Tv[:, 0] * Tw[:, 0] + g0 * Tv[:, 1] * Tw[:, 1] + g2 * Tv[:, 2] * Tw[:, 2]
        hx = bx * bx - (g0 * Tu[:, 0] ** 2 + g0 * Tu[:, 1] ** 2 + g2 * Tu[:, 2] ** 2)
        hy = by * by - (g0 * Tv[:, 0] ** 2 + g0 * Tv[:, 1] ** 2 + g2 * Tv[:, 2] ** 2)
        e3x = np.sqrt(np.maximum(hx, 0)) * e_rel + e_abs
        e3y = np.sqrt(np.maximum(hy, 0)) * e_rel + e_abs
        r2 = np.sqrt(0.5 * c2) + e_abs
        xmin, xmax = np.minimum(bx - e3x, cx - r2), np.maximum(bx + e3x, cx + r2)
        ymin, ymax = np.minimum(by - e3y, cy - r2), np.maximum(by + e3y, cy + r2)
        rx = np.maximum(cx - xmin, xmax - cx)
        ry = np.maximum(cy - ymin, ymax - cy)
        ok = (dd < 0) & np.isfinite(rx) & np.isfinite(ry)
    rx = np.where(ok, rx, np.inf)
    ry = np.where(ok, ry, np.inf)
    never = opa < 1.0 / 255.0
    rx[never] = -1
    ry[never] = -1
    return rx.astype(np.float32), ry.astype(np.float32)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "surface"
    nviews = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libblend_sim.so"))
    cams = synthetic.eval_cameras(8)
    g = synthetic.surface_surfels(100_000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100_000, seed=0)[0]
    H = W = 512
    views = []
    for v in range(nviews):
        o = _util.oracle_view(g, cams, v, H, W)
        tight = os.environ.get("TIGHT")
        rx, ry = cull_extents(o, 1.001, 0.002, 1.001, float(tight)) if tight else cull_extents(o)
        views.append((o, rx, ry, np.ascontiguousarray(o["xy"][:, 0]), np.ascontiguousarray(o["xy"][:, 1])))
    P = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    # instruction model (wave-level VALU instructions): per staged chunk, per trip, per entry slot
    CS, CT, CE = 110.0, 12.0, 73.0
    configs = [("PROXY sorted groups, window 2 (cur/nxt), kU 4", dict(window=2, kU=4, map=4, seg_min=2048, nseg=8)),
               ("PROXY sorted groups, window 8, kU 4", dict(window=8, kU=4, map=4, seg_min=2048, nseg=8)),
               ("PROXY sorted groups + pointer scheme, window 8, kU 2", dict(window=-8, kU=2, map=4, seg_min=2048, nseg=8)),
               ("ORACLE sorted groups, window 8, kU 4", dict(window=8, kU=4, map=2, seg_min=2048, nseg=8)),
               ("ORACLE sorted groups, window 4, kU 4", dict(window=4, kU=4, map=2, seg_min=2048, nseg=8)),
               ("PROXY sorted groups, window 4, kU 4", dict(window=4, kU=4, map=4, seg_min=2048, nseg=8)),
               ("current: quadrants, window 2, kU 4, 4 seg >= 1024", dict(window=2, kU=4, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 1", dict(window=1, kU=4, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 3", dict(window=3, kU=4, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 4", dict(window=4, kU=4, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 8", dict(window=8, kU=4, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 4, kU 2", dict(window=4, kU=2, map=0, seg_min=1024, nseg=4)),
               ("quadrants, window 2, kU 2", dict(window=2, kU=2, map=0, seg_min=1024, nseg=4)),
               ("lattice, window 2", dict(window=2, kU=4, map=1, seg_min=1024, nseg=4)),
               ("lattice, window 4", dict(window=4, kU=4, map=1, seg_min=1024, nseg=4)),
               ("sorted-in-tile groups, window 8", dict(window=8, kU=4, map=2, seg_min=1024, nseg=4)),
               ("sorted-in-tile groups, window 2", dict(window=2, kU=4, map=2, seg_min=1024, nseg=4)),
               ("round-robin-by-rank, window 8", dict(window=8, kU=4, map=3, seg_min=1024, nseg=4)),
               ("pointer scheme, window 4, kU 4", dict(window=-4, kU=4, map=0, seg_min=1024, nseg=4)),
               ("pointer scheme, window 8, kU 4", dict(window=-8, kU=4, map=0, seg_min=1024, nseg=4)),
               ("pointer scheme, window 8, kU 2", dict(window=-8, kU=2, map=0, seg_min=1024, nseg=4)),
               ("pointer scheme, window 8, kU 3", dict(window=-8, kU=3, map=0, seg_min=1024, nseg=4)),
               ("sorted groups + pointer scheme, window 8, kU 2", dict(window=-8, kU=2, map=2, seg_min=2048, nseg=8)),
               ("sorted groups + pointer scheme, window 8, kU 3", dict(window=-8, kU=3, map=2, seg_min=2048, nseg=8)),
               ("sorted groups + pointer scheme, window 8, kU 4", dict(window=-8, kU=4, map=2, seg_min=2048, nseg=8)),
               ("sorted groups, window 2 (cur/nxt), kU 4", dict(window=2, kU=4, map=2, seg_min=2048, nseg=8)),
               ("quadrants, window 2, kU 4, 8 seg >= 2048 (today)", dict(window=2, kU=4, map=0, seg_min=2048, nseg=8)),
               ("quadrants + pointer scheme, window 8, kU 2", dict(window=-8, kU=2, map=0, seg_min=2048, nseg=8)),
               ("quadrants, window 2, no segments", dict(window=2, kU=4, map=0, seg_min=1 << 30, nseg=1)),
               ("quadrants, window 4, 8 seg >= 2048 (else 4 >= 1024 not modelled)", dict(window=4, kU=4, map=0, seg_min=2048, nseg=8)),
               ]
    for name, kw in configs:
        tot_w, tot_i = SimOut(), SimOut()
        mw = 0.0
        acc = {k: 0.0 for k, _ in SimOut._fields_}
        acci = dict(acc)
        for o, rx, ry, cx, cy in views:
            sw, si = SimOut(), SimOut()
            lib.blend_sim(H, W, W // 16, H // 16, P(o["ranges"]), P(o["point_list"]), P(cx), P(cy), P(rx), P(ry), P(o["n_walked"]),
                          kw["window"], kw["kU"], kw["map"], kw["seg_min"], kw["nseg"],
                          ctypes.c_double(CS), ctypes.c_double(CT), ctypes.c_double(CE), ctypes.byref(sw), ctypes.byref(si))
            for k, _ in SimOut._fields_:
                if k == "max_wave_cost":
                    acc[k] = max(acc[k], getattr(sw, k)); acci[k] = max(acci[k], getattr(si, k))
                else:
                    acc[k] += getattr(sw, k); acci[k] += getattr(si, k)
        sc = 8.0 / nviews
        print(f"{name:70s} util {acc['pairs'] / acc['slots']:.3f} (ideal {acci['pairs'] / acci['slots']:.3f})  "
              f"pairs {acc['pairs'] * sc / 1e6:6.2f}M  slots/64 {acc['slots'] * sc / 64e6:6.3f}M  chunks {acc['chunks'] * sc / 1e3:6.1f}K  "
              f"cost {acc['total_cost'] * sc / 1e6:6.1f}M (ideal {acci['total_cost'] * sc / 1e6:6.1f}M)  max wave {acc['max_wave_cost'] / 1e3:6.1f}K  quadrants in lock step {acc['coupled_cost'] * sc / 1e6:6.1f}M")


if __name__ == "__main__":
    main()
