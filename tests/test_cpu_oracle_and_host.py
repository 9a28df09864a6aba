"""CPU suite (-m "not gpu"): the oracles against their golden vectors, the host-side logic, and the C-ABI surface."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from gaussiananything_amd import synthetic

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


# ---- DiT oracle: pinned against the reference's own model code ----------------------------------------------------
@pytest.mark.parametrize("stage", [1, 2])
def test_dit_oracle_matches_reference_golden(stage):
    from oracle import dit as od
    z = torch.load(synthetic.fixture_path(f"dit_ref_stage{stage}.pt"))
    y = od.dit_forward(z["state_dict"], z["x"], z["t"], z["context"])
    assert float((y - z["y"]).abs().max()) <= 1e-6
    ycfg = od.forward_with_cfg(z["state_dict"], z["x"], z["t"], z["context"], z["cfg_scale"])
    assert float((ycfg - z["y_cfg"]).abs().max()) <= 1e-5


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
    for hdr in ("ga_surfel.h", "ga_dit.h", "ga_decode.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"^\s*(?:int|size_t|const char \*)\s*(ga_[a-z0-9_]+)\s*\(", src, flags=re.M))
    assert {"ga_surfel_forward", "ga_surfel_workspace_layout", "ga_dit_forward", "ga_gemm_bf16",
            "ga_attention_bf16", "ga_tiny_attention", "ga_surfel_head", "ga_layernorm_modulate"} <= declared
    for name in declared:
        assert hasattr(L, name), name
    assert b"surfel" in L.ga_surfel_version() and b"dit" in dit_ops.lib().ga_dit_version()


def test_workspace_layout_and_argument_errors_without_a_gpu():
    from gaussiananything_amd import _lib
    L = _lib.lib()
    lay = _lib.GaSurfelWorkspaceLayout()
    assert L.ga_surfel_workspace_layout(100000, 8, 512, 512, 3_200_000, ctypes.byref(lay)) == 0
    offs = [getattr(lay, n) for n, _ in lay._fields_]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and lay.total_bytes < 2 ** 31
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
