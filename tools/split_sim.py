"""Design aid: lane-slot model of the split blend (tools/split_sim.c).  Usage: python tools/split_sim.py [surface|stress] [views]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussiananything_amd import synthetic  # noqa: E402
from tests import _util  # noqa: E402
from tools.blend_sim import cull_extents  # noqa: E402

FIELDS = ("cand_all", "pass_all", "cand_walk", "pass_walk", "candA", "passA", "slotsB_quad", "slotsB_sorted", "slotsB_sorted_tile",
          "groups", "itersA", "max_group_pairs", "entriesA")


NPOL = 12
POLK = (8, 8, 10, 10, 12, 12, 16, 16, 24, 32, 10, 12)
POLF = (4, 8, 4, 8, 4, 8, 4, 8, 8, 8, 1000, 1000)


class SplitOut(ctypes.Structure):
    _fields_ = [("polslots", ctypes.c_double * NPOL), ("polflush", ctypes.c_double * NPOL), ("quad_iters", ctypes.c_double),
                ("quad_chunks", ctypes.c_double)] + [(k, ctypes.c_double) for k in FIELDS]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "surface"
    nviews = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    so = os.path.join(ROOT, "tools", "_build", "libsplit_sim.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "split_sim.c"), "-lm"])
    lib = ctypes.CDLL(so)
    cams = synthetic.eval_cameras(8)
    g = synthetic.surface_surfels(100_000, seed=1)[0] if scene == "surface" else synthetic.random_surfels(100_000, seed=0)[0]
    H = W = 512
    P = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    views = []
    for v in range(nviews):
        o = _util.oracle_view(g, cams, v, H, W)
        rx, ry = cull_extents(o)
        views.append((o, rx, ry))
    sc = 8.0 / nviews
    for G in (64, 128, 256, 512, 1024):
        for kU in (1, 4):
            acc = dict.fromkeys(FIELDS, 0.0)
            pol = [0.0] * NPOL; polf = [0.0] * NPOL; qi = qc = 0.0
            for o, rx, ry in views:
                r = SplitOut()
                cx, cy = np.ascontiguousarray(o["xy"][:, 0]), np.ascontiguousarray(o["xy"][:, 1])
                opa = np.ascontiguousarray(o["normal_opacity"][:, 3])
                lib.split_sim(H, W, W // 16, H // 16, P(o["ranges"]), P(o["point_list"]), P(cx), P(cy), P(rx), P(ry), P(o["trans"]), P(opa),
                              P(o["n_walked"]), G, kU, ctypes.byref(r))
                for i in range(NPOL):
                    pol[i] += r.polslots[i]; polf[i] += r.polflush[i]
                qi += r.quad_iters; qc += r.quad_chunks
                for k in FIELDS:
                    acc[k] = max(acc[k], getattr(r, k)) if k == "max_group_pairs" else acc[k] + getattr(r, k)
            m = lambda k: acc[k] * sc / 1e6  # noqa: E731
            print(f"G {G:5d} kU {kU}: cand all {m('cand_all'):6.2f}M pass {m('pass_all'):6.2f}M | walked cand {m('cand_walk'):6.2f}M pass {m('pass_walk'):6.2f}M | "
                  f"A: cand {m('candA'):6.2f}M pass {m('passA'):6.2f}M iters {m('itersA') * 1e3:7.1f}K (util {acc['candA'] / 64 / acc['itersA']:.3f}) groups {m('groups') * 1e3:6.1f}K "
                  f"entries {m('entriesA'):5.2f}M maxpairs {acc['max_group_pairs']:.0f} | B slots/64: quad {m('slotsB_quad') / 64 * 1e3:7.1f}K sorted/group {m('slotsB_sorted') / 64 * 1e3:7.1f}K "
                  f"sorted/tile {m('slotsB_sorted_tile') / 64 * 1e3:7.1f}K  (ideal {m('pass_walk') / 64 * 1e3:7.1f}K)")
            if G == 64 and kU == 1:
                print(f"   per-quadrant streams: A iterations {qi * sc / 1e3:.1f}K over {qc * sc / 1e3:.1f}K (chunk, quadrant) visits")
                for i in range(NPOL):
                    print(f"   flush policy K {POLK[i]:3d} forced every {POLF[i]:4d} chunks: B slots/64 {pol[i] * sc / 64e3:7.1f}K  flushes {polf[i] * sc / 1e3:7.1f}K")


if __name__ == "__main__":
    main()
