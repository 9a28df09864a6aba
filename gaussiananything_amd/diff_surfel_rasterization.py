"""MI355X-native stand-in for the third-party ``diff_surfel_rasterization`` package.

Exports the two names the reference imports (``/root/reference/nsr/gs_surfel.py:15``) with the same call convention
(``:85-114``): ``GaussianRasterizationSettings`` (NamedTuple, 12 fields) and ``GaussianRasterizer(raster_settings)``
whose call returns ``(color[3,H,W], radii[N], allmap[7,H,W])``.  The arithmetic runs in hand-written HIP kernels behind
the C-ABI of ``include/ga_surfel.h``; there is no CPU path -- CPU tensors or a missing library raise.

``rasterize_views`` is the batched form the MI355X design is built around (all views of one Gaussian set in one
launch sequence); the per-view ``GaussianRasterizer`` call is a V=1 special case of it.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict
import warnings
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class SurfelWorkspace:
    """Caller-owned device scratch for ``ga_surfel_forward`` (the C-ABI allocates nothing).

    ``capacity`` is the number of binned (tile, splat) entries the buffers can hold, ``seg_capacity`` the number of
    (tile, segment) work items of the segmented blend (lists of 2048 entries or more) its exchange scratch holds (0 = the
    library's default, capacity / 2048 + 128; the worst case is capacity / 256).  The device reports the real counts and an
    overflow flag in ``status``; on overflow nothing was rendered and the caller grows the workspace and re-runs
    (``rasterize_views`` does that; ``grown`` sizes the replacement).
    """

    def __init__(self, device, num_points, num_views, height, width, capacity, seg_capacity=0):
        self.key = (num_points, num_views, height, width)
        self.device = device
        self.capacity = int(capacity)
        self.seg_capacity = int(seg_capacity)
        self.seg_items = self.seg_capacity if self.seg_capacity > 0 else self.capacity // 2048 + 128
        self.layout = _lib.GaSurfelWorkspaceLayout()
        _lib.check(_lib.lib().ga_surfel_workspace_layout2(num_points, num_views, height, width, self.capacity,
                                                          self.seg_capacity, ctypes.byref(self.layout)),
                   "ga_surfel_workspace_layout2")
        self.buffer = torch.empty(self.layout.total_bytes + 256, dtype=torch.uint8, device=device)
        self.offset = (-self.buffer.data_ptr()) % 256
        self.ptr = self.buffer.data_ptr() + self.offset
        # launch epoch of the segmented blend's exchange words: any start value will do, but recycled device memory may still
        # hold words of an earlier workspace -- start somewhere random so that they only match by a 2^-32 coincidence
        self.section("seg_table", torch.int32, 128)[_lib.GA_SEG_EPOCH_WORD] = int.from_bytes(os.urandom(4), "little") - (1 << 31)

    def section(self, name, dtype, count):
        """Typed view of one workspace section (tests read the integer artefacts through this)."""
        off = self.offset + getattr(self.layout, name)
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.buffer[off:off + nbytes].view(dtype)

    def status(self):
        return self.section("status", torch.int64, _lib.GA_STATUS_WORDS)

    def clean_flag(self):
        """GA_SURFEL_FLAG_WORKSPACE_CLEAN once a forward has been enqueued on this workspace (include/ga_surfel.h)."""
        return _lib.GA_SURFEL_FLAG_WORKSPACE_CLEAN if getattr(self, "clean", False) else 0

    def grown(self, st):
        """A replacement workspace for the counts an overflowed launch reported in its status words ``st`` (host tensor)."""
        need, seg_need = int(st[_lib.GA_STATUS_NUM_RENDERED]), int(st[_lib.GA_STATUS_SEG_WORK])
        if need > 0xFFFFFFFF:
            raise RuntimeError(f"{need} binned entries exceed the 2^32 limit of the tile ranges")
        cap = max(self.capacity, need + need // 4) if need > self.capacity else self.capacity
        seg = self.seg_capacity
        if seg_need > (seg if seg > 0 else cap // 2048 + 128):
            seg = seg_need + seg_need // 4 + 16
        n, v, h, w = self.key
        return SurfelWorkspace(self.device, n, v, h, w, cap, seg)


def _seg_T(ws):
    """The transmittance table a differentiable forward leaves for ``ga_surfel_backward`` (GaSurfelForwardArgs.seg_T): one row of
    256 floats per 128-entry list segment; kept with the workspace it was sized for.  ``GA_SURFEL_SEG_T=0``: do without (the
    backward then walks the lists once more for these products; A/B aid)."""
    if os.environ.get("GA_SURFEL_SEG_T", "1") == "0":
        return None
    t = getattr(ws, "seg_T", None)
    if t is None:
        n, v, h, w = ws.key
        rows = ws.capacity // 128 + v * ((h + 15) // 16) * ((w + 15) // 16) + 1
        t = ws.seg_T = torch.empty(rows * 256, dtype=torch.float32, device=ws.device)
    return t


def default_capacity(n, v):
    """Binned entries a fresh workspace is sized for: two tiles per (view, splat) on average -- BASELINE configs[1] needs 1.8;
    scenes that need more are reported by the device and the workspace is re-sized once."""
    return max(2 * n * v, 1 << 16)


EXTRA_FLAGS = int(os.environ.get("GA_SURFEL_FLAGS", "0"))   # GA_SURFEL_FLAG_* ORed into every forward (4: the split walk of the blend; A/B aid)
_WS_CACHE_SLOTS = 8          # most recently used (device, N, V, H, W) workspaces kept alive; older ones are released
_ws_cache = OrderedDict()
_AUTOGRAD_POOL_KEYS = 4       # shapes whose idle workspaces are kept (least recently used shape dropped first), two per shape
_autograd_pool = OrderedDict()  # (device, N, V, H, W) -> idle workspaces of the differentiable path


class _WorkspaceLease:
    """Ownership of one workspace by one differentiable forward.  The backward reads the tile lists and transmittances of exactly
    that forward, possibly several times (``retain_graph=True``: the reference's adaptive loss weight runs ``autograd.grad`` twice
    before the final backward, dnnlib/util.py ``calculate_adaptive_weight``), so the workspace goes back to the pool when the
    autograd node is FREED, not when a backward has run; stream order protects the reuse by a later forward."""

    def __init__(self, ws, key):
        self.ws, self.key = ws, key

    def __del__(self):
        try:
            pool = _autograd_pool.setdefault(self.key, [])
            _autograd_pool.move_to_end(self.key)
            if len(pool) < 2 and not any(w is self.ws for w in pool):
                pool.append(self.ws)
            while len(_autograd_pool) > _AUTOGRAD_POOL_KEYS:
                _autograd_pool.popitem(last=False)
        except Exception:   # interpreter shutdown
            pass


def clear_workspaces():
    """drop every cached / pooled rasterizer workspace (their device memory returns to the caching allocator)"""
    _ws_cache.clear()
    _autograd_pool.clear()


def _get_workspace(device, n, v, h, w, replacement=None):
    key = (str(device), n, v, h, w)
    ws = _ws_cache.pop(key, None)
    if replacement is not None:
        ws = replacement
    if ws is None:
        ws = SurfelWorkspace(device, n, v, h, w, default_capacity(n, v))
    _ws_cache[key] = ws          # most recently used last
    while len(_ws_cache) > _WS_CACHE_SLOTS:
        _ws_cache.popitem(last=False)
    return ws


def _f32c(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name} is on {t.device}: the surfel rasterizer only runs on an MI355X (HIP) device; "
                           "there is no CPU path")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    return t.detach().contiguous().float()




def postprocess_views(color, allmap, viewmatrix, image=None, rend_normal=None, depth=None):
    """The per-pixel post-processing of ``GaussianRenderer2DGS.render`` (nsr/gs_surfel.py:121-163) over the stacked outputs
    of ``rasterize_views`` in ONE pass (``ga_surfel_postprocess``): ``image = clamp(color, 0, 1)``, ``rend_normal`` =
    ``allmap[:, 2:5]`` rotated from view to world space, ``depth = nan_to_num(allmap[:, 5:6], 0, 0)``.  The optional
    outputs are written in place (contiguous ``[V,3,H,W]``, ``[V,3,H,W]``, ``[V,1,H,W]``)."""
    if color.device.type != "cuda":
        raise RuntimeError("gaussiananything_amd surfel post-processing only runs on an MI355X (HIP) device")
    V, _, H, W = color.shape
    assert allmap.shape == (V, 7, H, W) and color.is_contiguous() and allmap.is_contiguous()
    vm = viewmatrix.detach().to(device=color.device, dtype=torch.float32).reshape(V, 16).contiguous()
    image = torch.empty_like(color) if image is None else image
    rend_normal = torch.empty_like(color) if rend_normal is None else rend_normal
    depth = torch.empty((V, 1, H, W), dtype=torch.float32, device=color.device) if depth is None else depth
    for t, shp in ((image, (V, 3, H, W)), (rend_normal, (V, 3, H, W)), (depth, (V, 1, H, W))):
        assert t.shape == shp and t.is_contiguous() and t.dtype == torch.float32 and t.device == color.device
    args = _lib.GaSurfelPostArgs(V, H, W, color.data_ptr(), allmap.data_ptr(), vm.data_ptr(), image.data_ptr(),
                                 rend_normal.data_ptr(), depth.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(color.device).cuda_stream)
    _lib.check(_lib.lib().ga_surfel_postprocess(ctypes.byref(args), stream), "ga_surfel_postprocess")
    return image, rend_normal, depth


class _RasterizeViews(torch.autograd.Function):
    """Differentiable ``rasterize_views``: forward = ``ga_surfel_forward`` on a workspace of its own (the backward reads the
    tile ranges and point lists of exactly this call again), backward = ``ga_surfel_backward`` (gradients with respect to
    means3D, opacities, colours, scales and rotations, summed over the views; include/ga_surfel.h)."""

    @staticmethod
    def forward(ctx, means3D, opacities, colors, scales, rotations, vm, pm, bg, h, w, scale_modifier):
        n, v = means3D.shape[0], vm.shape[0]
        # a workspace of its own until the backward has read it; taken from / returned to a small pool instead of a fresh
        # allocation per call
        key = (str(means3D.device), n, v, h, w)
        pool = _autograd_pool.get(key) or []
        ws = pool.pop() if pool else SurfelWorkspace(means3D.device, n, v, h, w, default_capacity(n, v))
        while True:
            color, radii, allmap, _ = _rasterize_views_nograd(means3D, opacities, colors, scales, rotations, vm, pm, bg, h, w,
                                                              scale_modifier, workspace=ws, check_overflow=False, for_backward=True)
            st = ws.status().cpu()
            if int(st[_lib.GA_STATUS_OVERFLOW]) == 0:
                break
            ws = ws.grown(st)
        ctx.save_for_backward(means3D, opacities, colors, scales, rotations, vm, pm, bg, color, allmap, radii)
        ctx.lease, ctx.geom = _WorkspaceLease(ws, key), (n, v, h, w, float(scale_modifier))
        ctx.seg_T = getattr(ws, "seg_T", None) if os.environ.get("GA_SURFEL_SEG_T", "1") != "0" else None
        ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, g_color, g_radii, g_allmap):
        means3D, opacities, colors, scales, rotations, vm, pm, bg, color, allmap, radii = ctx.saved_tensors
        n, v, h, w, mod = ctx.geom
        ws, dev = ctx.lease.ws, means3D.device   # (owned by this node until it is freed: any number of backwards)
        g_color = torch.zeros_like(color) if g_color is None else g_color.detach().float().contiguous()
        g_allmap = torch.zeros_like(allmap) if g_allmap is None else g_allmap.detach().float().contiguous()
        L = _lib.lib()
        d_means, d_op = torch.empty_like(means3D), torch.empty_like(opacities)
        d_col, d_sc, d_rot = torch.empty_like(colors), torch.empty_like(scales), torch.empty_like(rotations)
        fwd = _lib.GaSurfelForwardArgs(
            n, v, h, w, mod, 0, means3D.data_ptr(), opacities.data_ptr(), colors.data_ptr(), scales.data_ptr(),
            rotations.data_ptr(), vm.data_ptr(), pm.data_ptr(), bg.data_ptr(), color.data_ptr(), allmap.data_ptr(),
            radii.data_ptr(), ws.ptr, ws.layout.total_bytes, ws.capacity, None, ws.seg_capacity,
            ctx.seg_T.data_ptr() if ctx.seg_T is not None else None, ctx.seg_T.numel() if ctx.seg_T is not None else 0)
        nbytes = int(L.ga_surfel_backward_scratch_bytes(ctypes.byref(fwd)))
        if nbytes == 0:
            raise RuntimeError("ga_surfel_backward_scratch_bytes: bad shape")
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        args = _lib.GaSurfelBackwardArgs(fwd, g_color.data_ptr(), g_allmap.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                         d_means.data_ptr(), d_op.data_ptr(), d_col.data_ptr(), d_sc.data_ptr(), d_rot.data_ptr())
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(L.ga_surfel_backward(ctypes.byref(args), stream), "ga_surfel_backward")
        return d_means, d_op, d_col, d_sc, d_rot, None, None, None, None, None, None


def rasterize_views(means3D, opacities, colors_precomp, scales, rotations, viewmatrix, projmatrix, bg,
                    image_height, image_width, scale_modifier=1.0, workspace: Optional[SurfelWorkspace] = None,
                    check_overflow: bool = True, stage_events=None):
    """``_rasterize_views_nograd`` (below), differentiable when autograd is recording and a Gaussian tensor requires grad:
    ``color`` and ``allmap`` then carry a grad_fn whose backward is ``ga_surfel_backward`` (``radii`` is not differentiable; the median-depth
    channel passes its gradient to the depth of the median contributor; the workspace of such a call is its own)."""
    if torch.is_grad_enabled() and any(getattr(t, "requires_grad", False) for t in
                                       (means3D, opacities, colors_precomp, scales, rotations)):
        device = means3D.device
        m = _f32g(means3D, "means3D", device)
        n = m.shape[0]
        o = _f32g(opacities, "opacities", device).reshape(-1)
        c, s, r = _f32g(colors_precomp, "colors_precomp", device), _f32g(scales, "scales", device), _f32g(rotations, "rotations", device)
        if m.shape != (n, 3) or o.shape != (n,) or c.shape != (n, 3) or s.shape != (n, 2) or r.shape != (n, 4):
            raise ValueError("expected means3D[N,3], opacities[N,1], colors_precomp[N,3], scales[N,2], rotations[N,4]")
        vm = _f32c(viewmatrix, "viewmatrix", device).reshape(-1, 16)
        pm = _f32c(projmatrix, "projmatrix", device).reshape(-1, 16)
        color, radii, allmap = _RasterizeViews.apply(m, o, c, s, r, vm, pm, _f32c(bg, "bg", device).reshape(3),
                                                     int(image_height), int(image_width), float(scale_modifier))
        return color, radii, allmap, None
    return _rasterize_views_nograd(means3D, opacities, colors_precomp, scales, rotations, viewmatrix, projmatrix, bg,
                                   image_height, image_width, scale_modifier, workspace, check_overflow, stage_events)


def _f32g(t: torch.Tensor, name: str, device) -> torch.Tensor:
    """``_f32c`` that keeps the autograd graph."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda" or t.device != device:
        raise RuntimeError(f"{name} is on {t.device}: the surfel rasterizer only runs on an MI355X (HIP) device; there is no CPU path")
    return t.contiguous().float()


def _rasterize_views_nograd(means3D, opacities, colors_precomp, scales, rotations, viewmatrix, projmatrix, bg,
                            image_height, image_width, scale_modifier=1.0, workspace: Optional[SurfelWorkspace] = None,
                            check_overflow: bool = True, stage_events=None, for_backward: bool = False):
    """Rasterize V views of one Gaussian set.  ``viewmatrix`` / ``projmatrix``: ``[V,4,4]`` row-vector matrices
    (``cam_view`` / ``cam_view_proj``).  Returns ``color [V,3,H,W]``, ``radii [V,N] int32``, ``allmap [V,7,H,W]`` and
    the workspace used (its ``status()`` holds D / overflow / longest tile list).

    ``check_overflow=True`` reads the device status word after the launch (one host sync, as upstream's read-back of
    ``num_rendered``) and transparently re-runs with a larger workspace; with ``False`` nothing synchronises and the
    caller inspects ``workspace.status()`` itself.  ``stage_events``: optional ctypes array of 5 ``hipEvent_t`` (see
    ``include/ga_surfel.h``), measurement only.
    """
    device = means3D.device
    means3D = _f32c(means3D, "means3D", device)
    n = means3D.shape[0]
    opacities = _f32c(opacities, "opacities", device).reshape(-1)
    colors = _f32c(colors_precomp, "colors_precomp", device)
    scales = _f32c(scales, "scales", device)
    rotations = _f32c(rotations, "rotations", device)
    if means3D.shape != (n, 3) or opacities.shape != (n,) or colors.shape != (n, 3) or scales.shape != (n, 2) \
            or rotations.shape != (n, 4):
        raise ValueError("expected means3D[N,3], opacities[N,1], colors_precomp[N,3], scales[N,2], rotations[N,4]")
    vm = _f32c(viewmatrix, "viewmatrix", device).reshape(-1, 16)
    pm = _f32c(projmatrix, "projmatrix", device).reshape(-1, 16)
    v = vm.shape[0]
    if pm.shape[0] != v:
        raise ValueError("viewmatrix and projmatrix must hold the same number of views")
    bg = _f32c(bg, "bg", device).reshape(3)
    h, w = int(image_height), int(image_width)

    color = torch.empty((v, 3, h, w), dtype=torch.float32, device=device)
    allmap = torch.empty((v, 7, h, w), dtype=torch.float32, device=device)
    radii = torch.empty((v, n), dtype=torch.int32, device=device)
    L = _lib.lib()
    stream = torch.cuda.current_stream(device).cuda_stream
    ws = workspace if workspace is not None else _get_workspace(device, n, v, h, w)
    if ws.key != (n, v, h, w):
        raise ValueError("workspace was laid out for a different problem size")
    with torch.cuda.device(device):
        while True:
            seg_T = _seg_T(ws) if for_backward else None     # (the forward of a differentiable call leaves it for the backward)
            args = _lib.GaSurfelForwardArgs(
                n, v, h, w, float(scale_modifier), ws.clean_flag() | EXTRA_FLAGS, means3D.data_ptr(), opacities.data_ptr(), colors.data_ptr(),
                scales.data_ptr(), rotations.data_ptr(), vm.data_ptr(), pm.data_ptr(), bg.data_ptr(),
                color.data_ptr(), allmap.data_ptr(), radii.data_ptr(), ws.ptr, ws.layout.total_bytes, ws.capacity,
                stage_events, ws.seg_capacity, seg_T.data_ptr() if seg_T is not None else None,
                seg_T.numel() if seg_T is not None else 0)
            _lib.check(L.ga_surfel_forward(ctypes.byref(args), ctypes.c_void_p(stream)), "ga_surfel_forward")
            ws.clean = True      # its tile scan leaves the workspace head ready for the next forward
            if not check_overflow:
                break
            st = ws.status().cpu()
            if int(st[_lib.GA_STATUS_OVERFLOW]) == 0:
                break
            if workspace is not None:
                raise RuntimeError(f"workspace capacity {ws.capacity} / {ws.seg_items} segment work items < "
                                   f"{int(st[_lib.GA_STATUS_NUM_RENDERED])} binned entries / {int(st[_lib.GA_STATUS_SEG_WORK])} work items")
            ws = _get_workspace(device, n, v, h, w, replacement=ws.grown(st))
    return color, radii, allmap, ws


class GaussianRasterizer(nn.Module):
    """Same surface as ``diff_surfel_rasterization.GaussianRasterizer``; differentiable with respect to means3D, opacities,
    colors_precomp, scales and rotations (``means2D`` is accepted and ignored: the reference passes zeros without
    requires_grad, /root/reference/nsr/gs_surfel.py:104-106, and never reads its gradient)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is not None:
            raise NotImplementedError("spherical-harmonics colours are not on the GaussianAnything path "
                                      "(nsr/gs_surfel.py always passes colors_precomp, sh_degree=0)")
        if cov3D_precomp is not None:
            raise NotImplementedError("precomputed transforms are not on the GaussianAnything path "
                                      "(nsr/gs_surfel.py always passes scales/rotations)")
        color, radii, allmap, _ = rasterize_views(
            means3D, opacities, colors_precomp, scales, rotations, rs.viewmatrix.reshape(1, 4, 4),
            rs.projmatrix.reshape(1, 4, 4), rs.bg, rs.image_height, rs.image_width, rs.scale_modifier)
        return color[0], radii[0], allmap[0]


class SurfelForwardPlan:
    """Pre-bound ``ga_surfel_forward`` call for repeated rendering of a fixed problem shape: inputs are converted
    once, outputs and workspace are allocated once, ``run()`` is a single C-ABI call that enqueues the five kernels on
    the current stream without any host synchronisation (so it can also be captured into a HIP graph).

    The sampling scripts render 50 cameras x 4 LoDs per sample (/root/reference/nsr/lsgm/flow_matching_trainer.py:
    1545-1616); a plan per LoD removes the per-call tensor juggling of the reference's Python loop.
    """

    def __init__(self, means3D, opacities, colors_precomp, scales, rotations, viewmatrix, projmatrix, bg,
                 image_height, image_width, scale_modifier=1.0, capacity=None, flags=0):
        device = means3D.device
        self.device = device
        self.means3D = _f32c(means3D, "means3D", device)
        self.n = n = self.means3D.shape[0]
        self.opacities = _f32c(opacities, "opacities", device).reshape(-1)
        self.colors = _f32c(colors_precomp, "colors_precomp", device)
        self.scales = _f32c(scales, "scales", device)
        self.rotations = _f32c(rotations, "rotations", device)
        self.vm = _f32c(viewmatrix, "viewmatrix", device).reshape(-1, 16)
        self.pm = _f32c(projmatrix, "projmatrix", device).reshape(-1, 16)
        self.v = v = self.vm.shape[0]
        self.bg = _f32c(bg, "bg", device).reshape(3)
        self.h, self.w = int(image_height), int(image_width)
        self.scale_modifier = float(scale_modifier)
        self.flags = int(flags) | EXTRA_FLAGS
        self.color = torch.empty((v, 3, self.h, self.w), dtype=torch.float32, device=device)
        self.allmap = torch.empty((v, 7, self.h, self.w), dtype=torch.float32, device=device)
        self.radii = torch.empty((v, n), dtype=torch.int32, device=device)
        self.ws = SurfelWorkspace(device, n, v, self.h, self.w, capacity or default_capacity(n, v))
        self._L = _lib.lib()
        self._bind(None)

    def _bind(self, stage_events):
        ws = self.ws
        self._args = _lib.GaSurfelForwardArgs(
            self.n, self.v, self.h, self.w, self.scale_modifier, self.flags | ws.clean_flag(), self.means3D.data_ptr(),
            self.opacities.data_ptr(), self.colors.data_ptr(), self.scales.data_ptr(), self.rotations.data_ptr(),
            self.vm.data_ptr(), self.pm.data_ptr(), self.bg.data_ptr(), self.color.data_ptr(),
            self.allmap.data_ptr(), self.radii.data_ptr(), ws.ptr, ws.layout.total_bytes, ws.capacity, stage_events,
            ws.seg_capacity)
        self._argp = ctypes.byref(self._args)

    def set_stage_events(self, stage_events):
        self._bind(stage_events)

    def run(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.ga_surfel_forward(self._argp, ctypes.c_void_p(stream)), "ga_surfel_forward")
        if not self.ws.clean_flag():     # from the second forward on the clearing memset is not needed any more
            self.ws.clean = True
            self._args.flags = self.flags | _lib.GA_SURFEL_FLAG_WORKSPACE_CLEAN

    def ensure_capacity(self):
        """One synchronising check (call once after the first run): grow the workspace and re-run on overflow."""
        while True:
            st = self.ws.status().cpu()
            if int(st[_lib.GA_STATUS_OVERFLOW]) == 0:
                return int(st[_lib.GA_STATUS_NUM_RENDERED])
            self.ws = self.ws.grown(st)
            self._bind(None)
            self.run()
