"""ctypes front-end of oracle/surfel_raster.c (CPU restatement of the 2DGS surfel rasterizer forward).

TEST INFRASTRUCTURE ONLY -- see the header of surfel_raster.c ("PARITY UNPINNED" applies here as well).
Returns every intermediate integer artefact (radii, tile rects, tiles_touched, sorted point list, tile
ranges) next to the images, so the HIP path can be checked bit-exactly on indices and by MSE on pixels.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_surfel.so")
_LIB64_PATH = os.path.join(_HERE, "_build", "liboracle_surfel_f64.so")   # blend arithmetic in double (yardstick, see surfel_raster.c)
_lib = None
_lib64 = None


class _Pre(ctypes.Structure):
    _fields_ = [
        ("depths", ctypes.c_void_p),
        ("xy", ctypes.c_void_p),
        ("trans", ctypes.c_void_p),
        ("normal_opacity", ctypes.c_void_p),
        ("radii", ctypes.c_void_p),
        ("rect", ctypes.c_void_p),
        ("tiles_touched", ctypes.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms).  Building the checker is not using it."""
    src = os.path.join(_HERE, "surfel_raster.c")
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < os.path.getmtime(src) for q in (_LIB_PATH, _LIB64_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_count.restype = ctypes.c_int64
    return _lib


def lib_f64():
    global _lib64
    if _lib64 is None:
        build()
        _lib64 = ctypes.CDLL(_LIB64_PATH)
        _lib64.oracle_count.restype = ctypes.c_int64
    return _lib64


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def rasterize(means3D, opacities, colors, scales, rotations, viewmatrix, projmatrix, bg, H, W,
              scale_modifier=1.0, threads=None, blend_f64=False):
    """One view.  Matrices follow the reference's row-vector convention (cam_view / cam_view_proj).

    Returns a dict: color[3,H,W], allmap[7,H,W], radii[N] i32, rect[N,4] u32, tiles_touched[N] u32,
    depths/xy/trans/normal_opacity (f32), point_list[D] u32, keys[D] u64, ranges[tiles,2] u32,
    final_T[H,W], n_contrib[H,W], n_walked[H,W] (list entries each pixel visited), D, pairs.
    """
    L = lib_f64() if blend_f64 else lib()     # (preprocess and binning are the same fp32 code in both libraries)
    means3D = _f32(means3D, (-1, 3)); N = means3D.shape[0]
    opacities = _f32(opacities, (N,)); colors = _f32(colors, (N, 3))
    scales = _f32(scales, (N, 2)); rotations = _f32(rotations, (N, 4))
    vm = _f32(viewmatrix, (16,)); pm = _f32(projmatrix, (16,)); bg = _f32(bg, (3,))
    out = {
        "depths": np.zeros(N, np.float32), "xy": np.zeros((N, 2), np.float32),
        "trans": np.zeros((N, 9), np.float32), "normal_opacity": np.zeros((N, 4), np.float32),
        "radii": np.zeros(N, np.int32), "rect": np.zeros((N, 4), np.uint32),
        "tiles_touched": np.zeros(N, np.uint32),
    }
    pre = _Pre(*[_p(out[k]) for k in ("depths", "xy", "trans", "normal_opacity", "radii", "rect", "tiles_touched")])
    L.oracle_preprocess(ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(W), _p(means3D), _p(opacities),
                        _p(scales), _p(rotations), ctypes.c_float(scale_modifier), _p(vm), _p(pm),
                        ctypes.byref(pre))
    D = int(L.oracle_count(ctypes.c_int(N), _p(out["tiles_touched"])))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    keys = np.zeros(max(D, 1), np.uint64); vals = np.zeros(max(D, 1), np.uint32)
    ktmp = np.zeros(max(D, 1), np.uint64); vtmp = np.zeros(max(D, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.oracle_bin(ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(W), ctypes.byref(pre), ctypes.c_int64(D),
                 _p(keys), _p(vals), _p(ktmp), _p(vtmp), _p(ranges))
    color = np.zeros((3, H, W), np.float32); allmap = np.zeros((7, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32); n_contrib = np.zeros((H, W), np.uint32)
    n_walked = np.zeros((H, W), np.uint32)
    pairs = ctypes.c_int64(0)
    if threads is not None:
        L.oracle_set_threads(ctypes.c_int(int(threads)))
    L.oracle_blend(ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(W), ctypes.byref(pre), _p(colors), _p(bg),
                   _p(vals), _p(ranges), _p(color), _p(allmap), _p(final_T), _p(n_contrib), _p(n_walked), ctypes.byref(pairs))
    out.update(color=color, allmap=allmap, point_list=vals[:D], keys=keys[:D], ranges=ranges,
               final_T=final_T, n_contrib=n_contrib, n_walked=n_walked, D=D, pairs=int(pairs.value))
    return out
