"""Distribution of the absolute pixel differences HIP vs C oracle (beside the MSE): per scene the largest difference per channel and
the number of pixels beyond 1e-5 / 1e-4 / 1e-3 -- what the max-abs bars of tests/test_surfel_gpu.py were set with."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiananything_amd import synthetic  # noqa: E402
from tests import _util  # noqa: E402

dev = torch.device("cuda:0")
cams = synthetic.eval_cameras(8)
cases = [("surface 100k 512", synthetic.surface_surfels(100_000, seed=1)[0], list(range(8)), 512, 512, 1.0),
         ("stress 100k 512", synthetic.random_surfels(100_000, seed=0)[0], list(range(8)), 512, 512, 1.0),
         ("1k 256", synthetic.random_surfels(1000, seed=0)[0], [0], 256, 256, 1.0),
         ("2k 250x300", synthetic.random_surfels(2000, seed=3)[0], [1, 5], 250, 300, 1.0),
         ("4k 1080p", synthetic.random_surfels(4000, seed=41)[0], [0, 3], 1080, 1920, 1.0),
         ("4k 2048x1024 x2", synthetic.random_surfels(4000, seed=41)[0], [1], 1024, 2048, 2.0),
         ("big splats 3k x3", synthetic.random_surfels(3000, seed=5)[0], [0, 5], 200, 296, 3.0)]
for name, g, views, H, W, sm in cases:
    color, radii, allmap, ws = _util.hip_views(g, cams, views, H, W, dev, scale_modifier=sm)
    art = _util.ws_artifacts(ws, g.shape[0], len(views), H, W)
    color, allmap, radii = color.cpu().numpy(), allmap.cpu().numpy(), radii.cpu().numpy()
    worst = np.zeros(10)
    cnt = np.zeros((3, 10), np.int64)
    cnt64 = np.zeros((3, 10), np.int64)      # the fp32 oracle against the same loop in double: what ANY fp32 order may differ by
    both = np.zeros(10, np.int64)
    tile_worst = 0
    ok = True
    tot = 0
    for k, v in enumerate(views):
        o = _util.oracle_view(g, cams, v, H, W, scale_modifier=sm)
        ok &= bool(np.array_equal(radii[k], o["radii"]))
        ts = art["tile_start"][k * art["tiles"]:(k + 1) * art["tiles"] + 1]
        ok &= bool(np.array_equal(art["point_list"][ts[0]:ts[0] + o["D"]].astype(np.uint32), o["point_list"]))
        tot += o["D"]
        d = np.concatenate([np.abs(color[k] - o["color"]), np.abs(allmap[k] - o["allmap"])], 0).reshape(10, -1)
        worst = np.maximum(worst, d.max(1))
        for i, th in enumerate((1e-5, 1e-4, 1e-3)):
            cnt[i] += (d > th).sum(1)
        o64 = _util.oracle_view(g, cams, v, H, W, scale_modifier=sm, blend_f64=True)
        d64 = np.concatenate([np.abs(o["color"] - o64["color"]), np.abs(o["allmap"] - o64["allmap"])], 0).reshape(10, -1)
        for i, th in enumerate((1e-5, 1e-4, 1e-3)):
            cnt64[i] += (d64 > th).sum(1)
        both += ((d > 1e-4) & (d64 > 1e-5)).sum(1)
        bad = (d > 1e-4).reshape(10, H, W)
        Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
        pad = np.zeros((10, Hp, Wp), bool)
        pad[:, :H, :W] = bad
        tile_worst = max(tile_worst, int(pad.reshape(10, Hp // 16, 16, Wp // 16, 16).sum((2, 4)).max()))
    print(f"{name}: bins identical {ok and tot == art['D']}, pixels {len(views) * H * W}")
    print("   max abs per channel (rgb | depth alpha nx ny nz median dist):", " ".join(f"{x:.2e}" for x in worst))
    for i, th in enumerate((1e-5, 1e-4, 1e-3)):
        print(f"   pixels beyond {th:g}:", cnt[i].tolist(), "  fp32 oracle vs fp64 blend:", cnt64[i].tolist())
    print("   HIP pixels beyond 1e-4 where fp32 and fp64 oracle also differ by > 1e-5:", both.tolist())
    print(f"   most pixels beyond 1e-4 in one 16x16 tile (any channel): {tile_worst}")
