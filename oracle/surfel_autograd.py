"""Differentiable float64 restatement of the 2DGS surfel forward in PyTorch -- the BACKWARD oracle for SURVEY.md section
8(f)-4 (rasterizer backward, not built in round 1) and a second, independent forward statement.

TEST INFRASTRUCTURE ONLY.  Written from the published method (Huang et al. 2024, 2D Gaussian Splatting: ray-splat
intersection, object-space low-pass filter, front-to-back alpha blending, depth distortion, sec. 4-5) with the constants
and output channels of the reference's consumer (/root/reference/nsr/gs_surfel.py:85-142), as a
plain per-pixel solve: for every pixel and surfel it finds the (u, v) whose world point p0 + su u tu + sv v tv projects
onto the pixel centre -- no homography / cross-product trick, no lists, no culling; of the rasterizer's tile machinery only
what changes results is kept: the CENTRE of the screen-space low-pass filter follows the bounding-box formula, and a splat
reaches only the pixels of its tile rectangle (see below).  Gradients come from autograd, so the
backward oracle is BY CONSTRUCTION the derivative of this forward; tests/test_cpu_oracle_and_host.py checks (a) the
forward against oracle/surfel_raster.c and (b) the gradients against finite differences (torch.autograd.gradcheck).

PARITY UNPINNED with respect to upstream's backward.cu (third-party, absent): the piecewise-constant choices are the
natural ones -- the selection min(rho3d, rho2d), the alpha >= 1/255 and T >= 1e-4 tests, the 0.99 clamp and the choice of
the median contributor are treated as constants of the gradient (the median depth itself is the depth of that pair and
carries its gradient there, as upstream's ``dL_dmedian_depth`` does); upstream additionally reports an absolute screen-space gradient for
densification (``means2D``), which is a training heuristic, not a derivative, and is not restated.
"""
from __future__ import annotations

import torch

NEAR, FAR = 0.2, 100.0


def quat_to_rot(q):
    """Rotation matrix of a (w, x, y, z) quaternion, re-normalised first (as the rasterizer does)."""
    w, x, y, z = (q / q.norm(dim=-1, keepdim=True)).unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


def render(means3D, opacities, colors, scales, rotations, cam_view, cam_view_proj, bg, H, W, scale_modifier=1.0):
    """All tensors float64; matrices in the reference's row-vector convention.  Returns (color [3,H,W], allmap [7,H,W]) =
    (expected depth, alpha, normal xyz, median depth, distortion), differentiable w.r.t. the five Gaussian tensors."""
    dt = torch.float64
    V, VP = cam_view.to(dt), cam_view_proj.to(dt)
    py, px = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    ndcx, ndcy = (2 * px + 1) / W - 1, (2 * py + 1) / H - 1            # pixel = ((ndc + 1) * size - 1) / 2
    N = means3D.shape[0]
    hom = torch.cat([means3D, torch.ones(N, 1, dtype=dt)], 1)
    zc_all = (hom @ V)[:, 2]
    order = torch.argsort(zc_all.detach(), stable=True)
    R = quat_to_rot(rotations)
    T = torch.ones(H, W, dtype=dt)
    C = torch.zeros(3, H, W, dtype=dt)
    Dp = torch.zeros(H, W, dtype=dt); Nrm = torch.zeros(3, H, W, dtype=dt); dist = torch.zeros(H, W, dtype=dt)
    M1 = torch.zeros(H, W, dtype=dt); M2 = torch.zeros(H, W, dtype=dt); median = torch.zeros(H, W, dtype=dt)
    for i in order.tolist():
        tu, tv, n = R[i, :, 0] * scales[i, 0] * scale_modifier, R[i, :, 1] * scales[i, 1] * scale_modifier, R[i, :, 2]
        p0 = means3D[i]
        z0 = torch.zeros(1, dtype=dt)
        c0, cu, cv = torch.cat([p0, z0 + 1]) @ VP, torch.cat([tu, z0]) @ VP, torch.cat([tv, z0]) @ VP   # clip = c0 + u cu + v cv
        a11, a12, b1 = cu[0] - ndcx * cu[3], cv[0] - ndcx * cv[3], -(c0[0] - ndcx * c0[3])
        a21, a22, b2 = cu[1] - ndcy * cu[3], cv[1] - ndcy * cv[3], -(c0[1] - ndcy * c0[3])
        det = a11 * a22 - a12 * a21
        u, v = (b1 * a22 - a12 * b2) / det, (a11 * b2 - a21 * b1) / det
        rho3d = u * u + v * v
        # centre of the low-pass filter: the rasterizer uses the centre of the splat's screen-space bounding box (SURVEY.md
        # A.1 step 5: from the homography columns Tu, Tv, Tw with t = (9, 9, -1)), which for a tilted splat under perspective
        # is NOT the projection of p0 (the paper's wording); with the latter a strongly tilted 0.11-scale splat differs by
        # 0.02 in alpha at its centre pixel
        Hm = torch.stack([torch.cat([tu, z0]), torch.cat([tv, z0]), torch.cat([p0, z0 + 1])], 0)          # 3 x 4
        Npix = torch.tensor([[W / 2, 0, 0], [0, H / 2, 0], [0, 0, 0], [(W - 1) / 2, (H - 1) / 2, 1]], dtype=dt)
        Mh = Hm @ VP @ Npix                                                                              # columns Tu, Tv, Tw
        tvec = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
        fvec = tvec / (Mh[:, 2] * Mh[:, 2] * tvec).sum()
        xc, yc = (fvec * Mh[:, 0] * Mh[:, 2]).sum(), (fvec * Mh[:, 1] * Mh[:, 2]).sum()
        rho2d = 2.0 * ((xc - px) ** 2 + (yc - py) ** 2)                                         # low-pass filter
        # The rasterizer's tile rectangle (oracle/surfel_raster.c: compute_aabb + getRect): a splat only
        # reaches the pixels of the 16 x 16 tiles its rect covers.  getRect's upper edge, trunc((c + radius + 15) / 16), leaves
        # out the tile that begins at 16 k when c + radius lies in [16 k, 16 k + 1) -- a pixel row / column inside the radius is
        # then NOT touched.  A constant of the gradient like the other selections.
        with torch.no_grad():
            ex = torch.sqrt(torch.clamp(xc * xc - (fvec * Mh[:, 0] * Mh[:, 0]).sum(), min=1e-4))
            ey = torch.sqrt(torch.clamp(yc * yc - (fvec * Mh[:, 1] * Mh[:, 1]).sum(), min=1e-4))
            radius = torch.ceil(torch.maximum(torch.maximum(ex, ey), torch.tensor(3.0 * 0.70710678118654752, dtype=dt)))
            gx, gy = (W + 15) // 16, (H + 15) // 16
            clampi = lambda v, hi_: min(hi_, max(0, int(v)))    # noqa: E731   (int(): truncation towards zero, as the C cast)
            rminx, rmaxx = clampi((xc - radius) / 16, gx), clampi((xc + radius + 15) / 16, gx)
            rminy, rmaxy = clampi((yc - radius) / 16, gy), clampi((yc + radius + 15) / 16, gy)
            in_rect = (px >= 16 * rminx) & (px < 16 * rmaxx) & (py >= 16 * rminy) & (py < 16 * rmaxy)
        zc = (torch.cat([p0, z0 + 1]) @ V)[2]
        use3d = (rho3d <= rho2d).detach()
        depth = torch.where(use3d, zc + u * (torch.cat([tu, z0]) @ V)[2] + v * (torch.cat([tv, z0]) @ V)[2], zc.expand(H, W))
        raw = opacities[i] * torch.exp(-0.5 * torch.where(use3d, rho3d, rho2d))
        alpha = torch.where(raw.detach() > 0.99, torch.full_like(raw, 0.99), raw)              # clamp: no gradient above it
        nv = n @ V[:3, :3]
        nv = nv * torch.sign(-torch.dot((torch.cat([p0, z0 + 1]) @ V)[:3], nv)).detach()        # facing the camera
        ok = ((alpha >= 1.0 / 255.0) & (depth >= NEAR) & (T * (1 - alpha) >= 1e-4)).detach() & in_rect
        wgt = torch.where(ok, alpha * T, torch.zeros_like(T))
        m = FAR / (FAR - NEAR) * (1 - NEAR / depth)
        dist = dist + torch.where(ok, (m * m * (1 - T) + M2 - 2 * m * M1) * wgt, torch.zeros_like(T))
        Dp = Dp + depth * wgt; M1 = M1 + m * wgt; M2 = M2 + m * m * wgt
        median = torch.where(ok & (T.detach() > 0.5), depth, median)          # (the selection is a constant; the depth is not)
        Nrm = Nrm + nv[:, None, None] * wgt
        C = C + colors[i][:, None, None] * wgt
        T = torch.where(ok, T * (1 - alpha), T)
    color = C + T[None] * bg.to(dt)[:, None, None]
    allmap = torch.stack([Dp, 1 - T, Nrm[0], Nrm[1], Nrm[2], median, dist], 0)
    return color, allmap
