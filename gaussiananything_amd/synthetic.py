"""Seeded synthetic inputs of the BASELINE.json configurations.

No reference asset is read at run time: the two data fixtures this module needs were exported once by
``tests/golden/make_golden.py`` into ``tests/golden/`` (camera poses ``eval_pose.pt[:8]`` and the eight in-tree
``fps-4096.ply`` clouds) and travel with the repository.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def fixture_path(name: str) -> str:
    return os.path.join(_GOLDEN, name)


def eval_cameras(num_views: int = 8):
    """Row-vector camera matrices of ``eval_pose.pt[:num_views]`` (exported with the reference's own camera code)."""
    z = np.load(fixture_path("cameras_eval8.npz"))
    assert num_views <= z["cam_view"].shape[0]
    return dict(poses=z["poses"][:num_views], cam_view=torch.from_numpy(z["cam_view"][:num_views]),
                cam_view_proj=torch.from_numpy(z["cam_view_proj"][:num_views]),
                cam_pos=torch.from_numpy(z["cam_pos"][:num_views]), tanfov=float(z["tanfov"]))


def random_surfels(n: int = 1000, seed: int = 0) -> torch.Tensor:
    """Config #1 / the uniform 'stress' scene: returns the reference's packed ``[1,N,13]`` Gaussian tensor
    (xyz, opacity, scale(2), quat wxyz(4), rgb(3); /root/reference/nsr/gs_surfel.py:68-72)."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * 0.9
    opacity = 0.05 + 0.95 * torch.rand(n, 1, generator=g)
    lo, hi = math.log(1e-3), math.log(2e-2)
    scales = torch.exp(lo + (hi - lo) * torch.rand(n, 2, generator=g))
    quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    rgb = torch.rand(n, 3, generator=g)
    return torch.cat([xyz, opacity, scales, quat, rgb], dim=1).unsqueeze(0).contiguous()


def surface_surfels(n: int = 100_000, seed: int = 1) -> torch.Tensor:
    """Config #2 'surface-like' scene: the eight in-tree FPS clouds (32 768 points) tiled up to ``n`` with Gaussian
    jitter (sigma 0.004), clipped to +-0.45; scales follow the decoder's activation/initialisation
    (softplus(N(-2.5,0.5)) * 0.0045/0.6931, /root/reference/vit/vit_triplane.py:1304-1309,315-322)."""
    clouds = np.load(fixture_path("fps_clouds.npz"))["xyz"].reshape(-1, 3)
    g = torch.Generator().manual_seed(seed)
    base = torch.from_numpy(clouds).float()
    reps = (n + base.shape[0] - 1) // base.shape[0]
    xyz = base.repeat(reps, 1)[:n]
    xyz = (xyz + 0.004 * torch.randn(n, 3, generator=g)).clamp(-0.45, 0.45)
    scales = torch.nn.functional.softplus(-2.5 + 0.5 * torch.randn(n, 2, generator=g)) * (0.0045 / 0.6931)
    opacity = torch.sigmoid(2.0 * torch.randn(n, 1, generator=g))
    quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    rgb = torch.rand(n, 3, generator=g)
    return torch.cat([xyz, opacity, scales, quat, rgb], dim=1).unsqueeze(0).contiguous()


def split_gaussians(gaussians_b: torch.Tensor):
    """[N,13] -> means3D, opacity[N,1], scales[N,2], rotations[N,4], rgbs[N,3] (nsr/gs_surfel.py:68-72)."""
    return (gaussians_b[:, 0:3].contiguous().float(), gaussians_b[:, 3:4].contiguous().float(),
            gaussians_b[:, 4:6].contiguous().float(), gaussians_b[:, 6:10].contiguous().float(),
            gaussians_b[:, 10:13].contiguous().float())


def recipe_state_dict(keys_shapes, seed: int):
    """Seeded weights for a state dict given as [(key, shape), ...] -- what a fixture stores INSTEAD of megabytes of weights when the
    reference outputs were computed on exactly these values (tests/golden/make_dit_golden.py: make_hd72).  CPU generator, one stream
    per tensor: matrices N(0, 1 / fan_in), norm weights 1 + 0.1 N(0, 1), tables 0.1 N(0, 1), biases 0.02 N(0, 1)."""
    out = {}
    for i, (key, shape) in enumerate(keys_shapes):
        g = torch.Generator().manual_seed(int(seed) * 100003 + i)
        t = torch.randn(tuple(shape), generator=g)
        if len(shape) >= 2 and "table" not in key:
            t = t / float(shape[-1]) ** 0.5
        elif "norm" in key and key.endswith("weight") or key.endswith(".0.weight") and len(shape) == 1:
            t = 1.0 + 0.1 * t
        elif "table" in key:
            t = 0.1 * t
        else:
            t = 0.02 * t
        out[key] = t
    return out
