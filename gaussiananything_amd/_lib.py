"""ctypes binding of the C-ABI in include/ga_surfel.h (libga_mi355.so, built in-tree by csrc/Makefile).

There is NO fallback: if the HIP library is missing or does not export a declared symbol this module raises, and
every product entry point that needs it fails loudly (the oracle under oracle/ is test infrastructure, never a
substitute).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libga_mi355.so")

GA_OK = 0
GA_STATUS_NUM_RENDERED, GA_STATUS_OVERFLOW, GA_STATUS_MAX_TILE, GA_STATUS_WORDS = 0, 1, 2, 16
GA_STATUS_SEG_WORK = 9
GA_SURFEL_FLAG_STATS, GA_SURFEL_FLAG_WORKSPACE_CLEAN, GA_SURFEL_FLAG_SPLIT_WALK, GA_SURFEL_FLAG_BG_IN_BLEND = 1, 2, 4, 8
GA_SEG_EPOCH_WORD = 96   # csrc/surfel_common.h: kSegEpochWord
GA_SURFEL_RECORD_FLOATS = 24
GA_SURFEL_STAGE_EVENTS = 5
_ERR = {-1: "GA_ERR_NULL_ARG", -2: "GA_ERR_BAD_SHAPE", -3: "GA_ERR_WORKSPACE", -4: "GA_ERR_LAUNCH"}


class GaSurfelForwardArgs(ctypes.Structure):
    _fields_ = [
        ("num_points", ctypes.c_int32), ("num_views", ctypes.c_int32),
        ("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32),
        ("scale_modifier", ctypes.c_float), ("flags", ctypes.c_int32),
        ("means3D", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("colors", ctypes.c_void_p),
        ("scales", ctypes.c_void_p), ("rotations", ctypes.c_void_p),
        ("viewmatrix", ctypes.c_void_p), ("projmatrix", ctypes.c_void_p), ("bg", ctypes.c_void_p),
        ("out_color", ctypes.c_void_p), ("out_others", ctypes.c_void_p), ("radii", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t), ("capacity", ctypes.c_int64),
        ("stage_events", ctypes.POINTER(ctypes.c_void_p)), ("seg_capacity", ctypes.c_int64),
        ("seg_T", ctypes.c_void_p), ("seg_T_floats", ctypes.c_int64),
    ]


class GaSurfelWorkspaceLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in (
        "status", "seg_sync", "tile_count", "tile_start", "tile_cursor", "tile_order", "run_table", "rect", "depth", "record", "keys",
        "point_list", "seg_table", "seg_scratch", "view_total", "total_bytes")]


class GaSurfelBackwardArgs(ctypes.Structure):
    """include/ga_surfel.h: GaSurfelBackwardArgs"""
    _fields_ = [("fwd", GaSurfelForwardArgs), ("grad_color", ctypes.c_void_p), ("grad_others", ctypes.c_void_p),
                ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t), ("grad_means3D", ctypes.c_void_p),
                ("grad_opacities", ctypes.c_void_p), ("grad_colors", ctypes.c_void_p), ("grad_scales", ctypes.c_void_p),
                ("grad_rotations", ctypes.c_void_p)]


class GaSurfelPostArgs(ctypes.Structure):
    """include/ga_surfel.h: GaSurfelPostArgs"""
    _fields_ = [("num_views", ctypes.c_int32), ("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32),
                ("color", ctypes.c_void_p), ("allmap", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
                ("image", ctypes.c_void_p), ("rend_normal", ctypes.c_void_p), ("depth", ctypes.c_void_p)]


class GaTsdfVolume(ctypes.Structure):
    """include/ga_tsdf.h: GaTsdfVolume"""
    _fields_ = [("units", ctypes.c_int32 * 3), ("unit0", ctypes.c_int32 * 3), ("voxel_length", ctypes.c_double),
                ("sdf_trunc", ctypes.c_double), ("tsdf", ctypes.c_void_p), ("weight", ctypes.c_void_p),
                ("color", ctypes.c_void_p), ("touched", ctypes.c_void_p), ("allocated", ctypes.c_void_p)]


class GaTsdfFrame(ctypes.Structure):
    """include/ga_tsdf.h: GaTsdfFrame"""
    _fields_ = [("height", ctypes.c_int32), ("width", ctypes.c_int32), ("rgb", ctypes.c_void_p), ("depth", ctypes.c_void_p),
                ("alpha", ctypes.c_void_p), ("alpha_thres", ctypes.c_float), ("depth_trunc", ctypes.c_float),
                ("fx", ctypes.c_double), ("fy", ctypes.c_double), ("cx", ctypes.c_double), ("cy", ctypes.c_double),
                ("extrinsic", ctypes.c_double * 16), ("pose", ctypes.c_double * 16), ("depth_sampling_stride", ctypes.c_int32)]


EXPORTS = ("ga_surfel_version", "ga_surfel_workspace_layout", "ga_surfel_workspace_layout2", "ga_surfel_forward", "ga_surfel_postprocess",
           "ga_surfel_backward", "ga_surfel_backward_scratch_bytes",
           "ga_tsdf_integrate", "ga_tsdf_mesh_scratch_bytes", "ga_tsdf_mesh_count", "ga_tsdf_mesh_emit", "ga_mesh_write_obj", "ga_mesh_cluster_labels")

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU) into LIB_PATH."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gaussiananything_amd/csrc`). "
                "There is no CPU fallback for the product path.")
        # torch first: the library's libamdhip64 dependency must resolve to the HIP runtime PyTorch has loaded (its streams and
        # allocations are handed to the kernels); loaded the other way round -- e.g. build() and then smoke() in one process --
        # the process ends up with the launches failing (GA_ERR_LAUNCH)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise RuntimeError(f"{LIB_PATH} does not export {name}")
        L.ga_surfel_version.restype = ctypes.c_char_p
        L.ga_surfel_workspace_layout.restype = ctypes.c_int
        L.ga_surfel_workspace_layout.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_int64,
                                                                       ctypes.POINTER(GaSurfelWorkspaceLayout)]
        L.ga_surfel_workspace_layout2.restype = ctypes.c_int
        L.ga_surfel_workspace_layout2.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_int64, ctypes.c_int64,
                                                                        ctypes.POINTER(GaSurfelWorkspaceLayout)]
        L.ga_surfel_forward.restype = ctypes.c_int
        L.ga_surfel_forward.argtypes = [ctypes.POINTER(GaSurfelForwardArgs), ctypes.c_void_p]
        L.ga_surfel_backward.restype = ctypes.c_int
        L.ga_surfel_backward.argtypes = [ctypes.POINTER(GaSurfelBackwardArgs), ctypes.c_void_p]
        L.ga_surfel_backward_scratch_bytes.restype = ctypes.c_size_t
        L.ga_surfel_backward_scratch_bytes.argtypes = [ctypes.POINTER(GaSurfelForwardArgs)]
        L.ga_surfel_postprocess.restype = ctypes.c_int
        L.ga_surfel_postprocess.argtypes = [ctypes.POINTER(GaSurfelPostArgs), ctypes.c_void_p]
        L.ga_tsdf_integrate.restype = ctypes.c_int
        L.ga_tsdf_integrate.argtypes = [ctypes.POINTER(GaTsdfVolume), ctypes.POINTER(GaTsdfFrame), ctypes.c_void_p]
        L.ga_tsdf_mesh_scratch_bytes.restype = ctypes.c_size_t
        L.ga_tsdf_mesh_scratch_bytes.argtypes = [ctypes.POINTER(GaTsdfVolume)]
        L.ga_tsdf_mesh_count.restype = ctypes.c_int
        L.ga_tsdf_mesh_count.argtypes = [ctypes.POINTER(GaTsdfVolume), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.ga_mesh_cluster_labels.restype = ctypes.c_int
        L.ga_mesh_cluster_labels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.ga_mesh_write_obj.restype = ctypes.c_int
        L.ga_mesh_write_obj.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        L.ga_tsdf_mesh_emit.restype = ctypes.c_int
        L.ga_tsdf_mesh_emit.argtypes = [ctypes.POINTER(GaTsdfVolume), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != GA_OK:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")
