"""CPU suite (-m "not gpu"): the oracles against their golden vectors, the host-side logic, and the C-ABI surface."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from gaussiananything_amd import synthetic
from tests import _util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- cameras: pinned against matrices computed by the reference's own graphics_utils -------------------------------
def test_cameras_match_reference_fixture():
    from gaussiananything_amd import cameras
    z = np.load(synthetic.fixture_path("cameras_eval8.npz"))
    for v in range(8):
        c = cameras.c_to_3dgs_format(z["poses"][v])
        assert np.array_equal(c["cam_view"].numpy(), z["cam_view"][v])
        assert np.array_equal(c["cam_view_proj"].numpy(), z["cam_view_proj"][v])
        assert np.array_equal(c["cam_pos"].numpy(), z["cam_pos"][v])
        assert c["tanfov"] == float(z["tanfov"])
    b = cameras.c_to_3dgs_format_batched(z["poses"])
    assert np.array_equal(b["cam_view"].numpy(), z["cam_view"])
    assert abs(float(z["tanfov"]) - 0.36) < 1e-3     # tanfov = 1/(2 fx), fx = 1.3889


# ---- surfel oracle ------------------------------------------------------------------------------------------------
def _cfg1():
    from oracle import surfel as osurf
    cams = synthetic.eval_cameras(1)
    g = synthetic.random_surfels(1000, seed=0)[0]
    m, o, s, r, c = synthetic.split_gaussians(g)
    return osurf.rasterize(m.numpy(), o.numpy(), c.numpy(), s.numpy(), r.numpy(), cams["cam_view"][0].numpy(),
                           cams["cam_view_proj"][0].numpy(), np.ones(3, np.float32), 256, 256)


def test_surfel_oracle_matches_frozen_golden():
    z = np.load(synthetic.fixture_path("surfel_cfg1_oracle.npz"))
    out = _cfg1()
    for k in ("radii", "rect", "tiles_touched", "point_list", "ranges"):
        assert np.array_equal(out[k], z[k]), k
    assert out["D"] == int(z["D"]) and out["pairs"] == int(z["pairs"])
    assert np.allclose(out["color"].astype(np.float64).sum((1, 2)), z["color_sum"], rtol=1e-6)
    assert np.allclose(out["allmap"].astype(np.float64).sum((1, 2)), z["allmap_sum"], rtol=1e-6)
    assert np.abs(out["color"] - z["color_f16"].astype(np.float32)).max() < 2e-3


def test_surfel_oracle_invariants():
    out = _cfg1()
    # sortedness: keys non-decreasing; inside a tile, depth-ordered with index as the tie-break
    assert np.all(np.diff(out["keys"].astype(np.uint64)) >= 0) or np.all(out["keys"][1:] >= out["keys"][:-1])
    tiles = (out["keys"] >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        lo, hi = out["ranges"][t]
        assert np.all(tiles[lo:hi] == t)
        d = out["depths"][out["point_list"][lo:hi]]
        assert np.all(np.diff(d) >= 0)
    # counts: D = sum of tiles touched = sum of tile range lengths
    assert out["D"] == int(out["tiles_touched"].sum()) == int((out["ranges"][:, 1] - out["ranges"][:, 0]).sum())
    # compositing: alpha in [0,1), colour = C + T*bg with white bg => every channel >= 1 - alpha - eps ... <= 1 + eps
    alpha = out["allmap"][1]
    assert alpha.min() >= 0 and alpha.max() <= 1.0
    assert np.all(out["color"] <= 1.0 + 1e-5) and np.all(out["color"] >= (1 - alpha) - 1e-5)
    # linearity in the background colour
    from oracle import surfel as osurf
    cams = synthetic.eval_cameras(1)
    m, o, s, r, c = [t.numpy() for t in synthetic.split_gaussians(synthetic.random_surfels(1000, seed=0)[0])]
    black = osurf.rasterize(m, o, c, s, r, cams["cam_view"][0].numpy(), cams["cam_view_proj"][0].numpy(),
                            np.zeros(3, np.float32), 256, 256)
    assert np.allclose(out["color"] - black["color"], (1 - alpha)[None], atol=1e-6)


def test_surfel_oracle_empty_and_culled():
    from oracle import surfel as osurf
    cams = synthetic.eval_cameras(1)
    m = np.full((4, 3), 50.0, np.float32)
    out = osurf.rasterize(m, np.ones(4), np.ones((4, 3)), np.full((4, 2), 0.01), np.tile([1, 0, 0, 0], (4, 1)),
                          cams["cam_view"][0].numpy(), cams["cam_view_proj"][0].numpy(), np.array([0.1, 0.2, 0.3]), 32, 48)
    assert out["D"] == 0 and np.all(out["radii"] == 0) and np.allclose(out["color"][2], 0.3)


def _analytic_surfel_render(means, opac, rgb, scales, quats, cam_view, cam_view_proj, bg, H, W):
    """Independent float64 statement of the 2DGS forward for a FEW surfels, from the published method (Huang et al. 2024,
    2D Gaussian Splatting, sec. 4: ray-splat intersection, object-space low-pass filter, front-to-back alpha blending,
    depth distortion) and NOT from the homography / cross-product formulation the oracle restates: for every pixel and
    surfel it solves directly for the (u, v) whose world point p0 + su u tu + sv v tv projects onto the pixel centre.
    No tile culling: callers keep opacity small enough that everything outside the 3-sigma box is below 1/255 anyway."""
    V, VP = cam_view.astype(np.float64), cam_view_proj.astype(np.float64)
    py, px = np.mgrid[0:H, 0:W].astype(np.float64)
    ndcx, ndcy = (2 * px + 1) / W - 1, (2 * py + 1) / H - 1        # pixel = ((ndc + 1) * size - 1) / 2
    near, far = 0.2, 100.0
    order = np.argsort([(np.append(m, 1.0) @ V)[2] for m in means], kind="stable")
    T = np.ones((H, W)); C = np.zeros((3, H, W)); am = np.zeros((7, H, W)); M1 = np.zeros((H, W)); M2 = np.zeros((H, W))
    for i in order:
        w, x, y, z = quats[i].astype(np.float64) / np.linalg.norm(quats[i])
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        tu, tv, n = R[:, 0] * scales[i][0], R[:, 1] * scales[i][1], R[:, 2]
        p0 = means[i].astype(np.float64)
        c0, cu, cv = np.append(p0, 1.0) @ VP, np.append(tu, 0.0) @ VP, np.append(tv, 0.0) @ VP   # clip = c0 + u cu + v cv
        # (c.x - ndcx c.w) = 0 and (c.y - ndcy c.w) = 0: two linear equations in (u, v) per pixel
        a11, a12, b1 = cu[0] - ndcx * cu[3], cv[0] - ndcx * cv[3], -(c0[0] - ndcx * c0[3])
        a21, a22, b2 = cu[1] - ndcy * cu[3], cv[1] - ndcy * cv[3], -(c0[1] - ndcy * c0[3])
        det = a11 * a22 - a12 * a21
        u, v = (b1 * a22 - a12 * b2) / det, (a11 * b2 - a21 * b1) / det
        rho3d = u * u + v * v
        xc, yc = ((c0[0] / c0[3] + 1) * W - 1) / 2, ((c0[1] / c0[3] + 1) * H - 1) / 2
        rho2d = 2.0 * ((xc - px) ** 2 + (yc - py) ** 2)                                    # low-pass filter, 1/sqrt(2) px
        zc = (np.append(p0, 1.0) @ V)[2]
        depth = np.where(rho3d <= rho2d, zc + u * (np.append(tu, 0) @ V)[2] + v * (np.append(tv, 0) @ V)[2], zc)
        alpha = np.minimum(0.99, opac[i] * np.exp(-0.5 * np.minimum(rho3d, rho2d)))
        nv = n @ V[:3, :3]
        nv = nv * np.sign(-np.dot((np.append(p0, 1.0) @ V)[:3], nv))                        # facing the camera
        ok = (alpha >= 1.0 / 255.0) & (depth >= near) & (T * (1 - alpha) >= 1e-4)
        wgt = np.where(ok, alpha * T, 0.0)
        m = far / (far - near) * (1 - near / depth)
        am[6] += np.where(ok, (m * m * (1 - T) + M2 - 2 * m * M1) * wgt, 0.0)
        am[0] += depth * wgt; M1 += m * wgt; M2 += m * m * wgt
        am[5] = np.where(ok & (T > 0.5), depth, am[5])
        for k in range(3):
            am[2 + k] += nv[k] * wgt
            C[k] += rgb[i][k] * wgt
        T = np.where(ok, T * (1 - alpha), T)
    am[1] = 1 - T
    return C + T[None] * np.asarray(bg, np.float64)[:, None, None], am


def test_surfel_oracle_against_the_published_method():
    """Known-answer test from the 2DGS paper's equations (the rasterizer itself is third-party and unavailable: 'parity
    unpinned'): a fronto-parallel surfel, a tilted anisotropic one and three overlapping ones, 96 x 96, against an
    independent float64 ray-splat solve.  Pins the oracle's homography, filter, sign conventions, blending order,
    median depth and distortion to the method as published."""
    from oracle import surfel as osurf
    cams = synthetic.eval_cameras(3)
    H = W = 96
    q_tilt = np.array([np.cos(0.45), 0.3 * np.sin(0.45), 0.8 * np.sin(0.45), 0.52 * np.sin(0.45)])
    for view, scene in ((0, "facing"), (1, "tilted"), (2, "stack")):
        cv, cvp = cams["cam_view"][view].numpy(), cams["cam_view_proj"][view].numpy()
        Rc = cv[:3, :3].astype(np.float64)                    # world -> view rotation (row-vector convention)
        if scene == "facing":      # tangents = the camera's x / y axes: the surfel faces the camera on its axis
            Rw = np.stack([Rc[:, 0], Rc[:, 1], np.cross(Rc[:, 0], Rc[:, 1])], axis=1)
            w_ = 0.5 * np.sqrt(max(1 + np.trace(Rw), 1e-12))
            quat = np.array([w_, (Rw[2, 1] - Rw[1, 2]) / (4 * w_), (Rw[0, 2] - Rw[2, 0]) / (4 * w_), (Rw[1, 0] - Rw[0, 1]) / (4 * w_)])
            means, quats = np.zeros((1, 3)), quat[None]
            opac, scales, rgb = np.array([0.3]), np.array([[0.05, 0.05]]), np.array([[0.9, 0.2, 0.1]])
        elif scene == "tilted":
            means, quats = np.array([[0.05, -0.03, 0.02]]), q_tilt[None]
            opac, scales, rgb = np.array([0.3]), np.array([[0.09, 0.03]]), np.array([[0.1, 0.8, 0.3]])
        else:
            means = np.array([[0.0, 0.0, 0.0], [0.03, 0.01, 0.1], [-0.02, 0.02, -0.12]])
            quats = np.stack([q_tilt, np.array([1.0, 0.1, -0.2, 0.05]), np.array([0.7, -0.4, 0.1, 0.3])])
            quats /= np.linalg.norm(quats, axis=1, keepdims=True)   # unit quaternions, as rot_act = F.normalize hands them over
            opac, scales = np.array([0.3, 0.25, 0.3]), np.array([[0.08, 0.05], [0.06, 0.09], [0.1, 0.04]])
            rgb = np.array([[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9]])
        bg = np.array([0.2, 0.5, 0.7], np.float32)
        out = osurf.rasterize(means.astype(np.float32), opac.astype(np.float32), rgb.astype(np.float32),
                              scales.astype(np.float32), quats.astype(np.float32), cv, cvp, bg, H, W)
        color, am = _analytic_surfel_render(means, opac, rgb, scales, quats, cv, cvp, bg, H, W)
        assert am[1].max() > 0.2, scene                                      # the splats are on screen
        assert np.abs(out["color"] - color).max() < 5e-5, scene        # fp32 oracle vs float64 statement
        for ch, name in enumerate(("depth", "alpha", "nx", "ny", "nz", "median", "dist")):
            assert np.abs(out["allmap"][ch] - am[ch]).max() < (5e-4 if name in ("depth", "median") else 5e-5), (scene, name)
        if scene == "facing":     # closed form on the optical axis: alpha = o exp(-r^2 / 2 s^2), normal = -z, no distortion
            assert abs(am[1].max() - 0.3) < 1e-2 and np.abs(out["allmap"][6]).max() < 1e-9   # centre lies between four pixel centres
            cy, cx = np.unravel_index(np.argmax(out["allmap"][1]), (H, W))
            nrm = out["allmap"][2:5, cy, cx] / out["allmap"][1, cy, cx]
            assert np.allclose(nrm, [0, 0, -1], atol=1e-4)
            assert abs(out["allmap"][0, cy, cx] / out["allmap"][1, cy, cx] - np.linalg.norm(cams["cam_pos"][view].numpy())) < 1e-3
        if scene == "stack":      # distortion of two layers at the busiest pixel: w_i w_j (m_i - m_j)^2 summed over pairs > 0
            assert out["allmap"][6].max() > 1e-7


def test_surfel_backward_oracle_forward_and_gradcheck():
    """oracle/surfel_autograd.py (the backward oracle of section 8(f)-4): its forward equals the C oracle, and its
    autograd gradients w.r.t. means, opacity, colour, scales and rotations equal finite differences of that forward."""
    from oracle import surfel as osurf
    from oracle import surfel_autograd as oag
    cams = synthetic.eval_cameras(2)
    H = W = 48
    g = torch.Generator().manual_seed(5)
    n = 6
    means = ((torch.rand(n, 3, generator=g) - 0.5) * 0.3).double()
    opac = (0.1 + 0.2 * torch.rand(n, generator=g)).double()           # < 0.35: nothing outside the 3-sigma box reaches 1/255
    rgb = torch.rand(n, 3, generator=g).double()
    scales = (0.04 + 0.08 * torch.rand(n, 2, generator=g)).double()
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1).double()
    cv, cvp = cams["cam_view"][1].double(), cams["cam_view_proj"][1].double()
    bg = torch.tensor([0.3, 0.6, 0.1], dtype=torch.float64)
    color, am = oag.render(means, opac, rgb, scales, quats, cv, cvp, bg, H, W)
    ref = osurf.rasterize(means.float().numpy(), opac.float().numpy(), rgb.float().numpy(), scales.float().numpy(),
                          quats.float().numpy(), cv.float().numpy(), cvp.float().numpy(), bg.float().numpy(), H, W)
    assert float(am[1].max()) > 0.2
    assert np.abs(ref["color"] - color.numpy()).max() < 5e-5
    for ch in range(7):
        assert np.abs(ref["allmap"][ch] - am[ch].numpy()).max() < (5e-4 if ch in (0, 5) else 5e-5), ch

    # gradients: a scalar functional of every differentiable output (random fixed weights), checked by finite differences
    wc = torch.rand(3, H, W, generator=g).double()
    wa = torch.rand(7, H, W, generator=g).double()
    # (channel 5, the median depth: the depth of the pair that crosses T = 0.5 -- differentiable through that depth, the choice
    #  of the pair being a constant, like the other selections)

    def f(m_, o_, c_, s_, q_):
        col, a_ = oag.render(m_, o_, c_, s_, q_, cv, cvp, bg, H, W)
        return (col * wc).sum() + (a_ * wa).sum()
    inputs = [t.clone().requires_grad_(True) for t in (means, opac, rgb, scales, quats)]
    assert torch.autograd.gradcheck(f, inputs, eps=1e-6, atol=2e-5, rtol=2e-4, nondet_tol=0.0)
    grads = torch.autograd.grad(f(*inputs), inputs)
    assert all(torch.isfinite(gr).all() and float(gr.abs().max()) > 0 for gr in grads)


# ---- DiT oracle: pinned against the reference's own model code ----------------------------------------------------
@pytest.mark.parametrize("stage", [1, 2])
def test_dit_oracle_matches_reference_golden(stage):
    from oracle import dit as od
    z = torch.load(synthetic.fixture_path(f"dit_ref_stage{stage}.pt"))
    y = od.dit_forward(z["state_dict"], z["x"], z["t"], z["context"])
    assert float((y - z["y"]).abs().max()) <= 1e-6
    ycfg = od.forward_with_cfg(z["state_dict"], z["x"], z["t"], z["context"], z["cfg_scale"])
    assert float((ycfg - z["y_cfg"]).abs().max()) <= 1e-5


def test_dit_oracle_and_module_with_heads_of_72_against_the_reference_golden():
    """DiT-PixArt-PCD-CLAY-XL's head geometry (/root/reference/dit/dit_i23d.py:1526-1535: 16 heads of 72 at width 1152): the fixture
    tests/golden/dit_ref_hd72.pt (8 heads of 72, depth 2) holds outputs of the REFERENCE'S classes on synthetic.recipe_state_dict weights --
    the oracle reproduces them, the module takes that state dict strictly, and the registry entry has the reference's shape."""
    from gaussiananything_amd.dit import DiT_I23D_PCD_PixelArt_noclip, DiT_models
    from oracle import dit as od
    z = torch.load(synthetic.fixture_path("dit_ref_hd72.pt"))
    sd = synthetic.recipe_state_dict(z["keys"], z["recipe_seed"])
    assert od.config_from_state_dict(sd)["num_heads"] == 8 and sd["blocks.0.attn.q_norm.weight"].shape[0] == 72
    y = od.dit_forward(sd, z["x"], z["t"], z["context"])
    assert float((y - z["y"]).abs().max()) <= 2e-5 * float(z["y"].abs().max())
    ycfg = od.forward_with_cfg(sd, z["x"], z["t"], z["context"], z["cfg_scale"])
    assert float((ycfg - z["y_cfg"]).abs().max()) <= 1e-4 * float(z["y_cfg"].abs().max())
    model = DiT_I23D_PCD_PixelArt_noclip(**z["kwargs"])
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == [(k, tuple(sh)) for k, sh in z["keys"]]
    model.load_state_dict(sd, strict=True)
    with torch.device("meta"):     # the registry entry itself (0.7 G parameters: shapes only)
        xl = DiT_models["DiT-PixArt-PCD-CLAY-XL"](input_size=16, in_channels=3, context_dim=1024, pooling_ctx_dim=768, num_classes=0,
                                                   learn_sigma=False, roll_out=True)
    assert (xl.depth, xl.embed_dim, xl.num_heads) == (28, 1152, 16) and xl.blocks[0].attn.q_norm.weight.shape == (72,)


@pytest.mark.parametrize("stage", [1, 2])
def test_dit_module_state_dict_is_reference_compatible(stage):
    from gaussiananything_amd.dit import DiT_I23D_PCD_PixelArt_noclip, DiT_I23D_PCD_PixelArt_noclip_clay_stage2, DiT_models
    z = torch.load(synthetic.fixture_path(f"dit_ref_stage{stage}.pt"))
    kw = dict(z["kwargs"])
    cls = DiT_I23D_PCD_PixelArt_noclip
    if stage == 2:
        kw["use_pe_cond"] = True
        cls = DiT_I23D_PCD_PixelArt_noclip_clay_stage2
    model = cls(**kw)
    sd = model.state_dict()
    assert list(sorted(sd)) == list(sorted(z["state_dict"]))
    assert all(sd[k].shape == v.shape for k, v in z["state_dict"].items())
    model.load_state_dict(z["state_dict"], strict=True)
    assert set(DiT_models) >= {"DiT-PixArt-PCD-CLAY-L", "DiT-PixArt-PCD-CLAY-B", "DiT-PixArt-PCD-CLAY-stage2-L"}
    with pytest.raises(RuntimeError):          # no CPU fallback
        model(z["x"], z["t"], z["context"])


# ---- transport -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["euler", "midpoint", "heun2", "heun3", "rk4", "dopri5"])
def test_odeint_matches_oracle_and_analytic(method):
    import scipy.linalg as sl
    from gaussiananything_amd.transport.odeint import odeint
    from oracle import ode as oo
    A = torch.tensor([[-0.5, 2.0], [-2.0, -0.5]])
    y0 = torch.tensor([[1.0, 0.0], [0.3, -0.7]])
    t = torch.linspace(0, 1, 40)
    s1, s2 = {}, {}
    a = odeint(lambda ts, y: y @ A.T * (1 + ts), y0, t, method=method, stats=s1)
    b = oo.odeint(lambda ts, y: (y @ A.numpy().astype(np.float64).T) * (1 + ts), y0.numpy(), t.numpy(), method=method, stats=s2)
    assert s1 == s2                                          # same NFE / accepted / rejected steps
    assert np.abs(a.numpy() - b).max() < 2e-5
    exact = np.stack([y0.numpy().astype(np.float64) @ sl.expm(A.numpy().astype(np.float64) * (x + x * x / 2)).T for x in t.numpy()])
    tol = {"euler": 0.1, "midpoint": 5e-3, "heun2": 5e-3, "heun3": 1e-4, "rk4": 1e-5, "dopri5": 5e-3}[method]
    assert np.abs(b - exact).max() < tol
    if method == "euler":
        assert s1["nfe"] == len(t) - 1                       # "250 steps" = 249 function evaluations


def _reference_transport():
    """the reference's transport package, imported with the oracle integrator standing in for torchdiffeq"""
    import types
    from oracle import ode as oo
    ref_root = "/root/reference"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not present (GPU box)")

    def fake_odeint(fn, x, t, method, atol, rtol):
        return torch.from_numpy(oo.odeint(lambda ts, y: fn(torch.tensor(ts, dtype=torch.float32),
                                                            torch.from_numpy(y).float()).numpy(),
                                          x.numpy(), t.numpy(), method=method, atol=atol[0], rtol=rtol[0])).float()

    saved = {k: sys.modules.get(k) for k in ("torchdiffeq", "sgm", "sgm.util", "transport")}
    sys.modules["torchdiffeq"] = types.SimpleNamespace(odeint=fake_odeint)
    sys.modules["sgm"] = types.ModuleType("sgm")
    sys.modules["sgm.util"] = types.SimpleNamespace(instantiate_from_config=lambda *a, **k: None)
    sys.path.insert(0, ref_root)
    for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
        del sys.modules[k]
    import transport as ref_transport

    def restore():
        sys.path.remove(ref_root)
        for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    return ref_transport, restore


@pytest.mark.parametrize("path_type", ["Linear", "GVP", "VP"])
@pytest.mark.parametrize("prediction", ["velocity", "score", "noise"])
def test_bound_reference_transport_integrates_with_this_package(path_type, prediction):
    """the REFERENCE'S OWN transport package (every parametrisation, every path plan -- none of it is
    rebuilt here) with ``bind_reference_transport`` applied keeps its results while ``ode.sample`` no longer reaches torchdiffeq:
    euler and dopri5, state by state against the unpatched reference driven by the oracle integrator."""
    from gaussiananything_amd.transport import bind_reference_transport
    ref_transport, restore = _reference_transport()
    try:
        model = lambda x, t, scale=1.0: torch.tanh(scale * x) * (0.5 + t.view(-1, 1, 1)) - 0.3 * x  # noqa: E731
        x0 = torch.randn(3, 4, 2, generator=torch.Generator().manual_seed(5))
        rt = ref_transport.create_transport(path_type, prediction, None, None, None, snr_type="uniform")
        # (score / noise drifts are singular at an end of the interval: the adaptive controller is exercised on the velocity models)
        methods = ("euler", "dopri5") if prediction == "velocity" else ("euler", "heun3")
        want = {m: ref_transport.Sampler(rt).sample_ode(sampling_method=m, num_steps=9)(x0, model, scale=1.5) for m in methods}
        theirs = ref_transport.integrators.ode.sample
        bind_reference_transport(ref_transport.integrators)
        import torchdiffeq                 # (sys.modules holds the stand-in of _reference_transport here, never the real package)
        stand_in = torchdiffeq.odeint
        try:
            torchdiffeq.odeint = None     # a call would now fail
            for m in methods:
                sampler = ref_transport.Sampler(rt)
                got = sampler.sample_ode(sampling_method=m, num_steps=9)(x0, model, scale=1.5)
                assert got.shape == want[m].shape and torch.allclose(got, want[m], rtol=1e-4, atol=2e-5, equal_nan=True), (m, float((got - want[m]).abs().max()))
        finally:
            torchdiffeq.odeint = stand_in
            ref_transport.integrators.ode.sample = theirs
    finally:
        restore()


def test_minimal_sampler_refuses_what_it_does_not_carry():
    from gaussiananything_amd.transport import Sampler, create_transport
    with pytest.raises(NotImplementedError):
        create_transport("VP", "velocity")
    with pytest.raises(NotImplementedError):
        create_transport("GVP", "score")
    with pytest.raises(NotImplementedError):
        Sampler(create_transport("GVP", "velocity")).sample_ode(reverse=True)


def test_transport_surface_and_reference_plumbing():
    """Sampler(create_transport(GVP, velocity)).sample_ode(...) against the REFERENCE'S transport/*.py imported with the
    oracle integrator standing in for torchdiffeq (and a stub for sgm.util): interval, grid and drift must agree."""
    from gaussiananything_amd.transport import Sampler, create_transport
    f = lambda x, t, scale=1.0: -scale * x * (1 + t.view(-1, 1))  # noqa: E731
    x0 = torch.tensor([[1.0, 2.0], [0.5, -1.0]])
    tr = create_transport("GVP", "velocity", None, None, None, snr_type="uniform")
    assert tr.check_interval(tr.train_eps, tr.sample_eps, sde=False, eval=True) == (0, 1)
    ours = Sampler(tr).sample_ode(sampling_method="euler", num_steps=25)(x0, f, scale=2.0)
    assert ours.shape == (25, 2, 2)
    ref_root = "/root/reference"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not present (GPU box)")
    import types
    from oracle import ode as oo

    def fake_odeint(fn, x, t, method, atol, rtol):
        return torch.from_numpy(oo.odeint(lambda ts, y: fn(torch.tensor(ts, dtype=torch.float32),
                                                            torch.from_numpy(y).float()).numpy(),
                                          x.numpy(), t.numpy(), method=method, atol=atol[0], rtol=rtol[0])).float()

    saved = {k: sys.modules.get(k) for k in ("torchdiffeq", "sgm", "sgm.util", "transport")}
    sys.modules["torchdiffeq"] = types.SimpleNamespace(odeint=fake_odeint)
    sys.modules["sgm"] = types.ModuleType("sgm")
    sys.modules["sgm.util"] = types.SimpleNamespace(instantiate_from_config=lambda *a, **k: None)
    sys.path.insert(0, ref_root)
    try:
        for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
            del sys.modules[k]
        import transport as ref_transport
        rt = ref_transport.create_transport("GVP", "velocity", None, None, None, snr_type="uniform")
        try:
            sampler = ref_transport.Sampler(rt, guider_config=None)
        except TypeError:
            sampler = ref_transport.Sampler(rt)
        ref = sampler.sample_ode(sampling_method="euler", num_steps=25)(x0, f, scale=2.0)
    finally:
        sys.path.remove(ref_root)
        for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    assert float((ours - ref).abs().max()) < 1e-5


# ---- C-ABI surface -------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from gaussiananything_amd import _lib, dit_ops
    L = _lib.lib()
    declared = set()
    for hdr in ("ga_surfel.h", "ga_dit.h", "ga_decode.h", "ga_tsdf.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"^\s*(?:int|size_t|const char \*)\s*(ga_[a-z0-9_]+)\s*\(", src, flags=re.M))
    assert {"ga_surfel_forward", "ga_surfel_workspace_layout", "ga_dit_forward", "ga_gemm_bf16",
            "ga_attention_bf16", "ga_tiny_attention", "ga_surfel_head", "ga_layernorm_modulate", "ga_tsdf_integrate",
            "ga_tsdf_mesh_count", "ga_tsdf_mesh_emit", "ga_tsdf_mesh_scratch_bytes"} <= declared
    for name in declared:
        assert hasattr(L, name), name
    assert b"surfel" in L.ga_surfel_version() and b"dit" in dit_ops.lib().ga_dit_version()


def test_workspace_layout_and_argument_errors_without_a_gpu():
    from gaussiananything_amd import _lib
    L = _lib.lib()
    lay = _lib.GaSurfelWorkspaceLayout()
    assert L.ga_surfel_workspace_layout(100000, 8, 512, 512, 3_200_000, ctypes.byref(lay)) == 0
    offs = [getattr(lay, n) for n, _ in lay._fields_]
    assert len(set(offs)) == len(offs) and all(o % 256 == 0 for o in offs) and lay.total_bytes == max(offs) < 2 ** 31
    # the words a forward accumulates into sit in front of tile_start: one memset clears them (or the previous forward left them clean)
    head = ("status", "seg_sync", "tile_count", "view_total", "tile_cursor")
    assert max(getattr(lay, n) for n in head) < lay.tile_start == min(getattr(lay, n) for n, _ in lay._fields_ if n not in head)
    assert L.ga_surfel_workspace_layout(-1, 8, 512, 512, 10, ctypes.byref(lay)) == -2       # GA_ERR_BAD_SHAPE
    assert L.ga_surfel_workspace_layout(10, 1, 16 * 70000, 16, 10, ctypes.byref(lay)) == -2
    assert L.ga_surfel_forward(None, None) == -1                                            # GA_ERR_NULL_ARG
    args = _lib.GaSurfelForwardArgs()
    args.num_points, args.num_views, args.image_height, args.image_width = 10, 1, 64, 64
    assert L.ga_surfel_forward(ctypes.byref(args), None) == -1


def test_rasterizer_refuses_cpu_tensors():
    from gaussiananything_amd.diff_surfel_rasterization import rasterize_views
    g = synthetic.random_surfels(10, seed=0)[0]
    m, o, s, r, c = synthetic.split_gaussians(g)
    cams = synthetic.eval_cameras(1)
    with pytest.raises(RuntimeError):
        rasterize_views(m, o, c, s, r, cams["cam_view"], cams["cam_view_proj"], torch.ones(3), 64, 64)
    # the multi-set entry point of the renderer (round 6) refuses the same way, before it touches a stream
    from gaussiananything_amd.gs_surfel import GaussianRenderer2DGS
    rd = GaussianRenderer2DGS.__new__(GaussianRenderer2DGS)      # (the constructor places its background tensor on the GPU)
    rd.bg_color = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rd.render_levels([g[None], g[None]], [32, 64], cams["cam_view"][None], cams["cam_view_proj"][None], cams["cam_pos"][None], cams["tanfov"])


def test_shard_samples_partition():
    from gaussiananything_amd.distributed import shard_samples
    for n, w in ((8, 8), (8, 3), (5, 8), (1, 2)):
        parts = [shard_samples(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_decode_oracle_matches_reference_golden():
    """oracle/decode.py against outputs of the reference's own surfel-decode classes / methods
    (tests/golden/make_decode_golden.py): all four Gaussian levels and the decoder features, exactly."""
    import torch
    from gaussiananything_amd import synthetic
    from oracle import decode as od
    z = torch.load(synthetic.fixture_path("decode_ref.pt"))
    out = od.decode(z["state_dict"], z["latent"], z["xyz"])
    for k in ("latent_from_vit", "gaussians_base", "gaussians_upsampled", "gaussians_upsampled_2", "gaussians_upsampled_3"):
        assert out[k].shape == z[k].shape
        assert float((out[k] - z[k]).abs().max()) <= 1e-6, k
    assert out["gaussians_upsampled_3"].shape[1] == z["latent"].shape[1] * 8 * 4 * 3


def test_batched_device_camera_conversion_matches_the_per_view_one():
    import torch
    from gaussiananything_amd import cameras
    poses = cameras.orbit_poses(12, seed=3)
    ref = cameras.c_to_3dgs_format_batched(poses)
    got = cameras.c_to_3dgs_format_device(torch.from_numpy(poses).reshape(3, 4, 25))
    for k in ("cam_view", "cam_view_proj", "cam_pos"):
        assert got[k].shape[:2] == (3, 4)
        assert float((got[k].reshape(12, *ref[k].shape[1:]) - ref[k]).abs().max()) < 2e-6, k
    assert abs(got["tanfov"] - ref["tanfov"]) < 1e-7


def test_ply_and_npy_handoff_formats(tmp_path):
    from gaussiananything_amd import io_formats as io
    rng = np.random.default_rng(0)
    xyz = (rng.random((768, 3), dtype=np.float32) - 0.5) * 1.2
    p = tmp_path / "stage1.ply"
    io.save_points_ply(p, xyz)
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 768\nproperty float x\n")
    assert np.array_equal(io.load_points_ply(p), xyz)
    got = io.load_stage1_points(p)
    assert got.shape == (1, 768, 3) and float(np.abs(got).max()) <= 0.45 and np.array_equal(got[0], np.clip(xyz, -0.45, 0.45))
    # ascii PLY with extra properties in another order (what other tools write)
    a = tmp_path / "ascii.ply"
    with open(a, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 3\nproperty uchar red\nproperty double z\nproperty float x\n"
                "property float y\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        f.write("255 3.0 1.0 2.0\n0 6.0 4.0 5.0\n7 9.0 7.0 8.0\n")
    assert np.array_equal(io.load_points_ply(a), np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32))
    g = rng.random((100, 13), dtype=np.float32)
    io.save_gaussians_npy(tmp_path / "g.npy", g)
    assert io.load_gaussians_npy(tmp_path / "g.npy").shape == (1, 100, 13)


def test_viewer_exports_glb_ply_obj_round_trip(tmp_path):
    """Section 8(f)-4: the files the engine leaves for viewers (flow_matching_trainer.py:1452-1475, 1742-1753,
    utils/mesh_util.py:113-136).  Parity unpinned against trimesh's byte layout; checked: the glTF container's invariants,
    the transforms of the call sites on known points, and that everything written reads back."""
    import json
    import struct
    from gaussiananything_amd import io_formats as io
    rng = np.random.default_rng(1)
    g = rng.random((1, 500, 13)).astype(np.float32)
    g[0, :, :3] -= 0.5
    g[0, 0, 10:13] = [1.7, -0.2, 0.5]                        # decoder colours are not clamped
    glb, ply, npy = io.export_gaussian_point_cloud(g, str(tmp_path), "sample0")
    assert glb.endswith("sample0-gaussian-pcd.glb") and ply.endswith("sample0-gaussian-pcd.ply") and npy.endswith("sample0-gaussian.npy")
    raw = open(glb, "rb").read()
    magic, version, total = struct.unpack_from("<4sII", raw, 0)
    assert magic == b"glTF" and version == 2 and total == len(raw) and total % 4 == 0
    jl, jt = struct.unpack_from("<I4s", raw, 12)
    assert jt == b"JSON" and jl % 4 == 0
    doc = json.loads(raw[20:20 + jl])
    assert doc["asset"]["version"] == "2.0" and doc["meshes"][0]["primitives"][0]["mode"] == 0     # POINTS
    bl, bt = struct.unpack_from("<I4s", raw, 20 + jl)
    assert bt == b"BIN\x00" and bl == doc["buffers"][0]["byteLength"] and 28 + jl + bl == total
    back = io.read_glb(glb)
    # R_x(-90 deg) on the points, then @ R_y(pi).T:  (x, y, z) -> (x, z, -y) -> (-x, z, y)
    want = np.stack([-g[0, :, 0], g[0, :, 2], g[0, :, 1]], 1)
    assert back["faces"] is None and np.abs(back["positions"] - want).max() < 1e-6
    acc = doc["accessors"][doc["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
    assert np.allclose(acc["min"], back["positions"].min(0)) and np.allclose(acc["max"], back["positions"].max(0))
    rgb8 = np.clip(np.round(g[0, :, 10:13].astype(np.float64) * 255), 0, 255).astype(np.uint8)
    assert np.array_equal(back["colors"][:, :3], rgb8) and (back["colors"][:, 3] == 255).all()
    assert tuple(back["colors"][0]) == (255, 0, 128, 255)
    xyz, rgba = io.load_colored_points_ply(ply)
    assert np.array_equal(xyz, back["positions"]) and np.array_equal(rgba, back["colors"])
    assert np.array_equal(io.load_points_ply(ply), xyz)         # the generic reader skips the colour properties
    assert np.array_equal(np.load(npy), g)
    # stage-1 hand-off: display copy rotated and grey, the stage-2 input file un-rotated
    pts = (rng.random((768, 3)).astype(np.float32) - 0.5) * 0.9
    glb1, ply1 = io.export_stage1_point_cloud(pts, str(tmp_path), "stage1")
    b1 = io.read_glb(glb1)
    assert np.abs(b1["positions"] - np.stack([pts[:, 0], pts[:, 2], -pts[:, 1]], 1)).max() < 1e-6   # v @ R_x(-90).T
    assert (b1["colors"] == np.array([26, 26, 26, 255], np.uint8)).all()                        # round(0.1 * 255)
    assert np.array_equal(io.load_stage1_points(ply1)[0], pts)
    # meshes: a tetrahedron
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64)
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], np.float32)
    io.save_glb(v, f, c, str(tmp_path / "m.glb"))
    m = io.read_glb(str(tmp_path / "m.glb"))
    assert m["mode"] == 4 and np.array_equal(m["faces"], f) and np.array_equal(m["positions"], v * np.array([-1, 1, -1], np.float32))
    assert np.array_equal(m["colors"][:, :3], (c * 255).astype(np.uint8))
    io.save_obj(v, f, c, str(tmp_path / "m.obj"))
    lines = open(tmp_path / "m.obj").read().splitlines()
    vs = np.array([[float(t) for t in l.split()[1:]] for l in lines if l.startswith("v ")])
    fs = np.array([[int(t) for t in l.split()[1:]] for l in lines if l.startswith("f ")])
    assert np.allclose(vs[:, :3], v * np.array([1, 1, -1])) and np.allclose(vs[:, 3:], c)
    assert np.array_equal(fs, f[:, ::-1] + 1)
    with pytest.raises(ValueError):
        io.write_glb(str(tmp_path / "bad.glb"), v, faces=np.array([[0, 1, 4]]))
    io.write_glb(str(tmp_path / "empty.glb"), np.zeros((0, 3), np.float32))
    assert io.read_glb(str(tmp_path / "empty.glb"))["positions"].shape == (0, 3)


def test_save_2dgs_ply_round_trip(tmp_path):
    """nsr/gs_surfel.py:206-265 (upstream body does not run: undefined names); the file it describes, read back."""
    from gaussiananything_amd import io_formats as io
    from gaussiananything_amd.gs_surfel import GaussianRenderer2DGS
    rng = np.random.default_rng(2)
    g = np.concatenate([rng.random((1, 300, 3)) - 0.5, rng.random((1, 300, 1)) * 0.98 + 0.01, rng.random((1, 300, 2)) * 0.02 + 1e-4,
                        rng.standard_normal((1, 300, 4)), rng.random((1, 300, 3))], -1).astype(np.float32)
    r = GaussianRenderer2DGS.__new__(GaussianRenderer2DGS)          # the constructor places a tensor on the GPU
    for compatible in (True, False):
        p = str(tmp_path / "sub" / f"surfels_{int(compatible)}.ply")
        r.save_2dgs_ply(p, torch.from_numpy(g), compatible=compatible)
        head = open(p, "rb").read(400).decode("ascii", "ignore")
        assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 300\nproperty float x\n")
        names = [l.split()[2] for l in head.splitlines() if l.startswith("property float")]
        assert names == list(io._2DGS_PLY_FIELDS) and len(names) == 16
        back = r.load_2dgs_ply(p, compatible=compatible)
        assert back.shape == (1, 300, 13) and float((back - torch.from_numpy(g)).abs().max()) < 2e-6
    raw = io.load_2dgs_ply(str(tmp_path / "sub" / "surfels_1.ply"), compatible=False)[0]
    assert np.allclose(raw[:, 3], np.log(g[0, :, 3] / (1 - g[0, :, 3])), atol=1e-5)            # logit
    assert np.allclose(raw[:, 4:6], np.log(g[0, :, 4:6] + 1e-8), atol=1e-5)
    assert np.allclose(raw[:, 10:13], (g[0, :, 10:13] - 0.5) / 0.28209479177387814, atol=1e-5)
    with pytest.raises(AssertionError):
        io.save_2dgs_ply(str(tmp_path / "b2.ply"), np.zeros((2, 4, 13), np.float32))


def test_conditioner_oracle_and_host_surface():
    """Section 8(f)-3 (parity unpinned): the preprocess restatement behaves as specified, the parameter container has the
    DINOv2 state-dict layout, and the product path refuses to run without the GPU."""
    from oracle import dinov2 as od
    from gaussiananything_amd.conditioner import FrozenDinov2ImageEmbedder
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 3, 28, 28, generator=g) * 2 - 1
    same = od.preprocess(img, 28)
    mean = torch.tensor(od.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(od.IMAGENET_STD).view(1, 3, 1, 1)
    assert torch.allclose(same, ((img + 1) / 2 - mean) / std, atol=1e-6)          # same size: no blur, no resampling
    up = od.resize_bicubic(img, 56)
    assert up.shape[-1] == 56 and torch.allclose(up[..., ::55, ::55], img[..., ::27, ::27], atol=1e-5)   # align_corners=True
    const = od.resize_bicubic(torch.full((1, 3, 64, 64), 0.25), 28)
    assert torch.allclose(const, torch.full_like(const, 0.25), atol=1e-5)         # blur and bicubic both preserve constants
    e = FrozenDinov2ImageEmbedder(arch="vitl", output_cls=True, inp_size=28, _vit_kwargs=dict(embed_dim=128, depth=2, num_heads=2, img_size=28))
    x = torch.rand(1, 3, 50, 50, generator=g) * 2 - 1
    assert torch.allclose(e.preprocess(x), od.preprocess(x, 28), atol=1e-6)
    keys = set(e.model.state_dict())
    want = {"cls_token", "pos_embed", "register_tokens", "mask_token", "patch_embed.proj.weight", "patch_embed.proj.bias",
            "norm.weight", "norm.bias"}
    for i in range(2):
        for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                  "ls1.gamma", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                  "mlp.fc2.bias", "ls2.gamma"):
            want.add(f"blocks.{i}.{k}")
    assert keys == want
    full = FrozenDinov2ImageEmbedder.__init__.__code__.co_varnames
    for arg in ("arch", "version", "device", "max_length", "freeze", "antialias", "ucg_rate", "unsqueeze_dim",
                "repeat_to_max_len", "num_image_crops", "output_tokens", "output_cls", "init_device", "inp_size"):
        assert arg in full                                                         # modules.py:797-812
    tok, cls = od.embed(e.model.state_dict(), img, 28)
    assert tok.shape == (2, 4, 128) and cls.shape == (2, 128)
    with pytest.raises(RuntimeError):
        e(img)                                                                    # no CPU fallback


def test_committed_bench_line_follows_the_contract():
    """profiles/r*_bench.json (the last full `python bench.py` line of the round) carries every field of the driver's
    contract with the right types, and the derived numbers are consistent with each other."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_bench.json")))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["unit"] == "Msplats/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    cfg = d["config"]
    assert abs(d["value"] - cfg["points"] * cfg["views"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    # ("valu": round 3's review asked for the roof that binds; achieved / peak / frac stay the contract's HBM pair, valu_issue_frac beside)
    assert r["bound"] in ("hbm", "mfma", "valu") and "valu_issue_frac" in r and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    # round 5: the absolute pixel bars and the full-depth trajectory on `parity`; the compact summary is the LAST key of the line
    p = d["parity"]
    assert p["pass"] is True and p["bins_bit_identical"] is True and p["max_abs"] <= 0.25 and p["pixels_beyond_1e-4_fraction"] <= 3e-4
    # round 6: the CPU trajectory oracle left the default run (it doubled the driver's run time); the line quotes the committed
    # full-length run instead, or carries the 25-point measurement when --trajectory-parity was given
    if "trajectory" in p:
        assert 0 < p["trajectory_rel_l2"] < 3e-2 and p["trajectory"]["pass"] is True
    elif "trajectory_250_points" in p:
        assert 0 < p["trajectory_250_points"]["end_state_rel_l2"] < 3e-2 and os.path.exists(os.path.join(ROOT, "profiles", "r6_traj250.txt"))
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) <= 1500
    assert d["summary"]["sec_per_sample"] == d["sec_per_sample"] and d["summary"]["parity"]["traj_rel_l2"] == p.get("trajectory_rel_l2")


def test_dopri5_refuses_non_finite_models_and_step_underflow():
    """torchdiffeq asserts on a non-finite / underflowing step; our loop must not spin forever on a NaN model output."""
    from gaussiananything_amd.transport.odeint import odeint
    with pytest.raises(FloatingPointError):
        odeint(lambda t, y: y * float("nan"), torch.ones(2, 2), torch.linspace(0, 1, 5))
    with pytest.raises(RuntimeError):
        odeint(lambda t, y: -y, torch.ones(2, 2), torch.linspace(0, 1, 5), max_steps=1)


def test_surfel_oracle_renormalises_quaternions():
    """scaled quaternions give the same splats (integer artefacts equal, pixels to rounding)."""
    cams = synthetic.eval_cameras(1)
    g = synthetic.random_surfels(800, seed=4)[0].clone()
    g2 = g.clone()
    g2[:, 6:10] *= torch.exp(2.0 * torch.randn(800, 1, generator=torch.Generator().manual_seed(1)))
    a, b = _util.oracle_view(g, cams, 0, 96, 96), _util.oracle_view(g2, cams, 0, 96, 96)
    assert float(np.mean(a["radii"] != b["radii"])) < 0.01 and abs(int(a["D"]) - int(b["D"])) <= 8
    assert float(np.mean((a["color"] - b["color"]) ** 2)) < 1e-8


def test_stage2_conditioning_follows_the_release_config():
    """sgm/configs/stage2-i23d.yaml: the 'fps-xyz' embedder is PCD_Scaler(0.45) -> the stage-2 denoiser sees xyz / 0.45;
    cond_key 'img-xyz' matches no embedder input key, so uc == c (guidance is a no-op); the decoder gets the raw cloud.
    Checked on the orchestration itself with recording stand-ins for the two denoisers and the decoder."""
    from gaussiananything_amd import cascade

    class Den(torch.nn.Module):
        def __init__(self, C):
            super().__init__()
            self.in_channels = C
            self.w = torch.nn.Parameter(torch.zeros(1))
            self.seen = []

        def forward_with_cfg(self, x, t, context=None, cfg_scale=1.0):
            self.seen.append({k: v.clone() for k, v in context.items()} | {"x0": x.clone()})
            return -x

        def forward(self, x, t, context=None):
            self.seen.append({k: v.clone() for k, v in context.items()} | {"x0": x.clone(), "cond_only": True})
            return -x

    class Dec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vit_decoder = torch.nn.Module()
            self.vit_decoder.pos_embed = torch.nn.Parameter(torch.zeros(1, 16, 4))

        def decode(self, latent, xyz):
            return {"latent": latent, "query_pcd_xyz": xyz}

    s1, s2, dec = Den(3), Den(10), Dec()
    cond = {"img_crossattn": torch.randn(1, 5, 8), "img_vector": torch.randn(1, 8)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    out = cascade.cascade(s1, s2, dec, cond, uc, num_steps=4, sampling_method="euler", seed=7, dedup_noop_cfg=False)
    first1, first2 = s1.seen[0], s2.seen[0]
    # stage 1: [cond | zero uncond]; initial state = CPU-seeded noise rounded to bf16 (flow_matching_trainer.py:720)
    assert torch.equal(first1["img_crossattn"][1], torch.zeros(5, 8)) and torch.equal(first1["img_crossattn"][0], cond["img_crossattn"][0])
    torch.manual_seed(7)
    z = torch.randn(1, 16, 3)
    assert torch.equal(first1["x0"][0], z[0].bfloat16().float()) and torch.equal(first1["x0"][0], first1["x0"][1])
    # stage 2: scaled cloud for the denoiser, both CFG halves conditional, raw cloud for the decoder
    raw = out["query_pcd_xyz"]
    assert float(raw.abs().max()) <= 0.45
    assert torch.allclose(first2["fps-xyz"][0], raw[0] / 0.45) and torch.equal(first2["fps-xyz"][0], first2["fps-xyz"][1])
    assert torch.equal(first2["img_crossattn"][0], first2["img_crossattn"][1]) and torch.equal(first2["img_crossattn"][0], cond["img_crossattn"][0])
    # default: the no-op guidance of stage 2 is recognised and the denoiser runs on the conditional half alone -- same result
    n_before = len(s2.seen)
    out_d = cascade.cascade(s1, s2, dec, cond, uc, num_steps=4, sampling_method="euler", seed=7)
    assert s2.seen[n_before].get("cond_only") and s2.seen[n_before]["x0"].shape[0] == 1
    assert torch.allclose(s2.seen[n_before]["fps-xyz"][0], raw[0] / 0.45) and torch.equal(out_d["latent"], out["latent"])
    out0 = cascade.cascade(s1, s2, dec, cond, uc, num_steps=4, sampling_method="euler", seed=7, stage2_zero_image_uc=True)
    assert torch.equal(s2.seen[-1]["img_crossattn"][1], torch.zeros(5, 8)) and out0["latent"].shape == (1, 16, 10)


# ---- conditioner oracle against an independent published implementation ---------------------------------------------------
def test_dinov2_oracle_against_the_transformers_implementation():
    """oracle/dinov2.py::vit_forward restates facebookresearch/dinov2 (torch.hub code, absent here).  Hugging Face transformers
    ships an independent implementation of the same published architecture (Dinov2WithRegistersModel); with the same randomly
    drawn weights, mapped from its parameter names to the hub layout, both must produce the same tokens -- this pins the
    oracle's encoder arithmetic (patch embedding order, cls / register / patch token order, position embedding, pre-norm blocks
    with LayerScale, erf-GELU MLP, final LayerNorm, head split of the fused qkv) to published code.  (The kornia resize in
    front of it stays unpinned.)"""
    tr = pytest.importorskip("transformers")
    if not hasattr(tr, "Dinov2WithRegistersModel"):
        pytest.skip("transformers without Dinov2WithRegistersModel")
    from oracle import dinov2 as od
    torch.manual_seed(0)
    D, depth, heads, R, P, grid = 128, 3, 2, 4, 14, 5
    cfg = tr.Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4,
                                       image_size=P * grid, patch_size=P, num_register_tokens=R, layerscale_value=1.0,
                                       hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True, use_swiglu_ffn=False,
                                       attn_implementation="eager")
    hf = tr.Dinov2WithRegistersModel(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():          # every parameter away from its initial value (LayerScale = 1, zero biases, ...)
            p.copy_(torch.randn(p.shape) * (0.3 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 and p.numel() == D else 0.0))
    h = hf.state_dict()
    sd = {"cls_token": h["embeddings.cls_token"], "pos_embed": h["embeddings.position_embeddings"],
          "register_tokens": h["embeddings.register_tokens"],
          "patch_embed.proj.weight": h["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": h["embeddings.patch_embeddings.projection.bias"],
          "norm.weight": h["layernorm.weight"], "norm.bias": h["layernorm.bias"]}
    for i in range(depth):
        a, b = f"encoder.layer.{i}.", f"blocks.{i}."
        for n in ("weight", "bias"):
            sd[b + "norm1." + n] = h[a + "norm1." + n]
            sd[b + "norm2." + n] = h[a + "norm2." + n]
            sd[b + "attn.qkv." + n] = torch.cat([h[a + f"attention.attention.{w}.{n}"] for w in ("query", "key", "value")], 0)
            sd[b + "attn.proj." + n] = h[a + "attention.output.dense." + n]
            sd[b + "mlp.fc1." + n] = h[a + "mlp.fc1." + n]
            sd[b + "mlp.fc2." + n] = h[a + "mlp.fc2." + n]
        sd[b + "ls1.gamma"] = h[a + "layer_scale1.lambda1"]
        sd[b + "ls2.gamma"] = h[a + "layer_scale2.lambda1"]
    img = torch.randn(2, 3, P * grid, P * grid)
    with torch.no_grad():
        ref = hf(pixel_values=img).last_hidden_state
    out = od.vit_forward(sd, img)
    assert out["x_norm_patchtokens"].shape == (2, grid * grid, D)
    scale = float(ref.abs().max())
    assert float((out["x_norm_clstoken"] - ref[:, 0]).abs().max()) < 2e-5 * scale
    assert float((out["x_norm_patchtokens"] - ref[:, 1 + R:]).abs().max()) < 2e-5 * scale


def test_ode_oracle_tableau_against_scipy_rk45():
    """oracle/ode.py restates torchdiffeq's dopri5 (absent here).  SciPy's RK45 is an independent implementation of the same
    Dormand-Prince 5(4) pair: nodes, stage matrix and the 5th-order weights must agree, one step from the same state must give
    the same y1, and the oracle's error weights (torchdiffeq pairs the 5th-order solution with a different 4th-order one than
    SciPy) must at least be the difference of two order->=4 quadratures: sum E c^q = 0 for q = 0..3 -- a mistyped coefficient
    would break one of these.  (The step-size controller stays unpinned.)"""
    sp = pytest.importorskip("scipy.integrate")
    from scipy.integrate._ivp.rk import RK45, rk_step
    from oracle import ode as oo
    c = np.array([0.0] + oo.A[:5])                       # nodes of stages 1..6 (the 7th, FSAL, stage sits at 1)
    assert np.allclose(c, RK45.C, rtol=0, atol=1e-15)
    for i in range(5):
        assert np.allclose(oo.B[i], RK45.A[i + 1][:i + 1], rtol=0, atol=1e-15), i
    assert np.allclose(oo.B[5], RK45.B, rtol=0, atol=1e-15)
    cs = np.array([0.0] + oo.A)                          # all seven stages
    for q in range(4):
        assert abs(float(np.dot(oo.E, cs ** q))) < 1e-15, q
    assert abs(float(np.dot(oo.E, cs ** 4))) > 1e-4      # ... and differ at order 5
    f = lambda t, y: np.array([y[1], -y[0] + 0.1 * np.sin(3 * t), -0.5 * y[2] * y[0]])   # noqa: E731
    y0, t0, h = np.array([1.0, 0.3, -0.7]), 0.2, 0.05
    K = np.empty((7, 3))
    y_sp, _ = rk_step(f, t0, y0, f(t0, y0), h, RK45.A, RK45.B, RK45.C, K)
    seen = {}
    oo.odeint(lambda t, y: f(t, y), y0, [t0, t0 + h], method="dopri5", atol=1e9, rtol=1e9, stats=seen)   # (never rejects)
    k = [f(t0, y0)]
    for i in range(6):
        yi = y0 + h * sum(cc * kk for cc, kk in zip(oo.B[i], k))
        k.append(f(t0 + oo.A[i] * h, yi))
    assert np.allclose(yi, y_sp, rtol=1e-14, atol=1e-15)


def test_backward_segment_numbers_in_closed_form_do_not_collide():
    """csrc/surfel_backward.hip numbers the 128-entry segments of tile t's list from floor(tile_start[t] / 128) + t (and the blend
    that leaves the transmittance table for it, csrc/surfel_blend.hip, does the same): no scan over the tiles.  The numbers of
    consecutive tiles must not overlap, leave at most one number unused between two tiles, and stay below the row count both
    sides allocate (capacity / 128 + tiles + 1) -- checked here on random list lengths including empty and exactly-full ones."""
    rng = np.random.default_rng(7)
    for trial in range(200):
        tiles = int(rng.integers(1, 400))
        kind = trial % 4
        if kind == 0:
            n = rng.integers(0, 700, tiles)
        elif kind == 1:
            n = rng.integers(0, 3, tiles) * 128            # empty and exactly full segments
        elif kind == 2:
            n = (rng.random(tiles) < 0.1) * rng.integers(1, 6000, tiles)     # mostly empty, a few long lists
        else:
            n = rng.integers(120, 137, tiles)
        start = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        first = start[:-1] // 128 + np.arange(tiles)
        nseg = (n + 127) // 128
        nxt = start[1:] // 128 + np.arange(tiles) + 1       # the next tile's first number (one past the last for the last tile)
        assert np.all(first + nseg <= nxt)                   # no overlap
        assert np.all(nxt - (first + nseg) <= 1)             # at most one unused number
        capacity = int(start[-1])
        assert int(nxt[-1]) <= capacity // 128 + tiles       # < the capacity / 128 + tiles + 1 rows allocated


def test_ctypes_mirrors_have_the_layout_of_the_c_headers(tmp_path):
    """Every argument struct the Python host passes through the C-ABI is declared twice: in include/*.h and as a ctypes.Structure.
    A C program compiled against the headers prints sizeof and every field's offsetof; they must be what ctypes computes for the
    mirror (same field names, same order, same padding) -- a field appended on one side only would otherwise shift silently."""
    import ctypes
    import importlib
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mirrors = []
    for mod in ("gaussiananything_amd._lib", "gaussiananything_amd.dit_ops", "gaussiananything_amd.decode_ops"):
        m = importlib.import_module(mod)
        for name in dir(m):
            obj = getattr(m, name)
            if isinstance(obj, type) and issubclass(obj, ctypes.Structure) and name.startswith("Ga") and obj.__module__ == mod:
                mirrors.append(obj)
    assert len(mirrors) >= 20
    lines = ["#include <stddef.h>", "#include <stdio.h>"] + [f'#include "{h}"' for h in sorted(os.listdir(os.path.join(root, "include")))]
    lines.append("int main(void) {")
    for cls in mirrors:
        lines.append(f'  printf("{cls.__name__} %zu\\n", sizeof({cls.__name__}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cls.__name__}.{fname} %zu\\n", offsetof({cls.__name__}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cls in mirrors:
        assert int(got[cls.__name__]) == ctypes.sizeof(cls), cls.__name__
        for fname, _ in cls._fields_:
            assert int(got[f"{cls.__name__}.{fname}"]) == getattr(cls, fname).offset, (cls.__name__, fname)


def test_the_algebra_of_the_folded_modulated_prenorm():
    """What the GEMM epilogues of the denoiser rely on (include/ga_dit.h, 'folding a MODULATED RMSNorm'), in fp64 on the oracle's own
    RMSNorm: (norm(x) w (1 + scale_b) + shift_b) W^T + bias  ==  rsqrt(mean x^2 + eps) ((x w (1 + scale_b)) W^T) + (bias + shift_b W^T),
    with the row sums of squares taken in 64-column groups of the RAW x, and a row that skipped the producer's product (k_rows)."""
    from oracle import dit as od
    g = torch.Generator().manual_seed(5)
    B, L, D, N = 3, 7, 256, 192
    x = torch.randn(B * L, D, generator=g, dtype=torch.float64) * 3
    w = 1 + 0.2 * torch.randn(D, generator=g, dtype=torch.float64)
    scale, shift = (0.3 * torch.randn(2, B, D, generator=g, dtype=torch.float64)).unbind(0)
    W = torch.randn(N, D, generator=g, dtype=torch.float64) / D ** 0.5
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    rep = lambda v: v.repeat_interleave(L, 0)
    normed = od.rmsnorm(x, w)
    ref = (normed * (1 + rep(scale)) + rep(shift)) @ W.T + bias
    emit = x * (w[None] * (1 + rep(scale)))                                   # producer: emit_x (before the bf16 rounding)
    ss = x.pow(2).reshape(B * L, D // 64, 64).sum(-1)                         # producer: emit_ss partial sums
    rs = torch.rsqrt(ss.sum(-1, keepdim=True) / D + 1e-5)                     # consumer: row_ss
    bias_rows = bias[None] + shift @ W.T                                      # ga_dit_shift_bias
    out = rs * (emit @ W.T) + rep(bias_rows)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)        # (the oracle's norm takes its mean in fp32)
