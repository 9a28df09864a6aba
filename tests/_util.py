"""Shared helpers of the parity tests: run the oracle and the HIP path on the same seeded inputs."""
import numpy as np
import torch

from gaussiananything_amd import synthetic


def oracle_view(g, cams, v, H, W, bg=(1.0, 1.0, 1.0), scale_modifier=1.0, blend_f64=False):
    from oracle import surfel as osurf
    m, o, s, r, c = synthetic.split_gaussians(g)
    return osurf.rasterize(m.numpy(), o.numpy(), c.numpy(), s.numpy(), r.numpy(), cams["cam_view"][v].numpy(),
                           cams["cam_view_proj"][v].numpy(), np.asarray(bg, np.float32), H, W,
                           scale_modifier=scale_modifier, blend_f64=blend_f64)


def hip_views(g, cams, views, H, W, device, bg=(1.0, 1.0, 1.0), scale_modifier=1.0):
    from gaussiananything_amd.diff_surfel_rasterization import rasterize_views
    m, o, s, r, c = [t.to(device) for t in synthetic.split_gaussians(g)]
    vm = cams["cam_view"][views].to(device)
    pm = cams["cam_view_proj"][views].to(device)
    color, radii, allmap, ws = rasterize_views(m, o, c, s, r, vm, pm, torch.tensor(bg, device=device), H, W,
                                               scale_modifier)
    torch.cuda.synchronize()
    return color, radii, allmap, ws


def ws_artifacts(ws, N, V, H, W):
    """Integer artefacts out of the HIP workspace, as numpy."""
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    st = ws.status().cpu().numpy()
    D = int(st[0])
    tile_start = ws.section("tile_start", torch.int32, V * tiles + 1).cpu().numpy().astype(np.int64)
    rect = ws.section("rect", torch.int16, V * N * 4).cpu().numpy().view(np.uint16).reshape(V, N, 4)
    point_list = ws.section("point_list", torch.int32, max(D, 1)).cpu().numpy()[:D]
    # cull box = centre (record floats 8, 9) +- the two fp16 half-extents packed into record float 15 (surfel_common.h)
    rec = ws.section("record", torch.float32, V * N * 24).cpu().numpy().reshape(V, N, 24)
    ext = np.ascontiguousarray(rec[..., 15]).view(np.uint32)
    rx = (ext & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    ry = (ext >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    with np.errstate(invalid="ignore"):
        bbox = np.stack([rec[..., 8] - rx, rec[..., 9] - ry, rec[..., 8] + rx, rec[..., 9] + ry], -1)
    return dict(D=D, overflow=int(st[1]), max_tile=int(st[2]), tile_start=tile_start, rect=rect,
                point_list=point_list, bbox=bbox, tiles=tiles)
