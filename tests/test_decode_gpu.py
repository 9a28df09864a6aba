"""GPU parity of the HIP surfel decode through the C-ABI of include/ga_decode.h, against
(i) plain PyTorch fp32 references of each new operator and (ii) the golden vectors produced by the REFERENCE'S OWN decoder
classes (tests/golden/decode_ref.pt, tests/golden/make_decode_golden.py).

Tolerances: row operators on fp32 data are compared after the bf16 rounding of their outputs (<= 1e-2 relative L2, the
bf16 quantum); the whole decode (bf16 MFMA inputs, fp32 accumulation and residual streams) to <= 2e-2 relative L2 per
Gaussian level and <= 4e-3 absolute on positions (scene extent 0.9; a position is the anchor plus up to four
tanh offsets of amplitude 0.225 whose arguments carry bf16-level relative error, 2^-8 * 0.225 ~ 1e-3 each).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("M,D,affine,mod", [(37, 128, False, True), (200, 768, True, False), (9, 1024, True, True)])
def test_layernorm_modulate(gpu_device, M, D, affine, mod):
    from gaussiananything_amd import decode_ops as dops
    g = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(gpu_device)
    w = torch.randn(D, generator=g).to(gpu_device) if affine else None
    b = torch.randn(D, generator=g).to(gpu_device) if affine else None
    modt = torch.randn(M, 6 * D, generator=g).to(gpu_device) if mod else None
    sc, sh = (modt[:, D:2 * D], modt[:, 0:D]) if mod else (None, None)
    eps = 1e-5 if affine else 1e-6
    y = dops.layernorm_modulate(x, eps, weight=w, bias=b, scale=sc, shift=sh)
    ref = F.layer_norm(x, (D,), w, b, eps)
    if mod:
        ref = ref * (1 + sc) + sh
    assert y.dtype == torch.bfloat16 and rel_l2(y.float(), ref) < 1e-2


@pytest.mark.parametrize("groups,S,heads", [(50, 9, 2), (333, 5, 12), (1000, 4, 12), (3, 16, 1)])
def test_tiny_attention(gpu_device, groups, S, heads):
    from gaussiananything_amd import decode_ops as dops
    g = torch.Generator().manual_seed(groups + S)
    C = heads * 64
    qkv = torch.randn(groups * S, 3 * C, generator=g).to(gpu_device).bfloat16()
    out = dops.tiny_attention(qkv, groups, S, heads)
    q, k, v = qkv.float().reshape(groups, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(groups * S, C)
    assert rel_l2(out.float(), ref) < 1e-2


def test_assemble_tokens_and_tiny_mlp(gpu_device):
    from gaussiananything_amd import decode_ops as dops
    g = torch.Generator().manual_seed(5)
    D, P0, f0, f1 = 128, 10, 8, 4
    feat = torch.randn(P0, D, generator=g).to(gpu_device)
    emb0 = torch.randn(f0, D, generator=g).to(gpu_device)
    t0 = dops.assemble_tokens(feat, emb0, P0, f0, 0).reshape(P0, 1 + f0, D)
    assert torch.equal(t0[:, 0], feat) and torch.equal(t0[:, 1:], emb0.expand(P0, -1, -1))
    emb1 = torch.randn(f1, D, generator=g).to(gpu_device)
    t1 = dops.assemble_tokens(t0.reshape(-1, D), emb1, P0 * f0, f1, f0).reshape(P0 * f0, 1 + f1, D)
    assert torch.equal(t1[:, 0], t0[:, 1:].reshape(P0 * f0, D)) and torch.equal(t1[:, 1:], emb1.expand(P0 * f0, -1, -1))
    x = torch.randn(77, 10, generator=g).to(gpu_device)
    w1, b1 = torch.randn(10, 10, generator=g).to(gpu_device), torch.randn(10, generator=g).to(gpu_device)
    w2, b2 = torch.randn(D, 10, generator=g).to(gpu_device), torch.randn(D, generator=g).to(gpu_device)
    y = dops.tiny_mlp_silu(x, w1, b1, w2, b2)
    ref = F.silu(F.linear(F.gelu(F.linear(x, w1, b1), approximate="tanh"), w2, b2))
    assert rel_l2(y.float(), ref) < 1e-2


def test_surfel_head_both_modes(gpu_device):
    from gaussiananything_amd import decode_ops as dops
    from oracle import decode as od
    g = torch.Generator().manual_seed(9)
    D, P, f = 128, 40, 3
    x = torch.randn(P, D, generator=g)
    w, b = torch.randn(13, D, generator=g) * 0.2, torch.randn(13, generator=g)
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * 0.9
    pre_ref = F.linear(F.silu(x), w, b)
    g_ref = od.activate(od.offset_act(pre_ref[..., :3]) * 0.1 + xyz, pre_ref)
    dev = lambda t: t.to(gpu_device).contiguous()
    gg, pre = dops.surfel_head(dev(x), dev(w), dev(b), dev(xyz), P, 0, skip_weight=0.1)
    assert rel_l2(pre.cpu(), pre_ref) < 1e-5 and float((gg.cpu() - g_ref).abs().max()) < 1e-5
    tok = torch.randn(P * (1 + f), D, generator=g)
    lw, lb = torch.randn(D, generator=g), torch.randn(D, generator=g)
    emb = tok.reshape(P, 1 + f, D)[:, 1:]
    res = F.linear(F.layer_norm(emb, (D,), lw, lb, 1e-5), w, b)
    pos = od.offset_act(res[..., :3]) + g_ref[:, None, :3]
    res = res + pre_ref[:, None]
    g2_ref = od.activate(pos, res).reshape(P * f, 13)
    g2, pre2 = dops.surfel_head(dev(tok), dev(w), dev(b), dev(g_ref), P * f, 1, f=f, ln_weight=dev(lw), ln_bias=dev(lb),
                                base_pre=dev(pre_ref))
    assert rel_l2(pre2.cpu(), res.reshape(P * f, 13)) < 1e-5 and float((g2.cpu() - g2_ref).abs().max()) < 1e-5


def test_decode_matches_reference_golden(gpu_device):
    from gaussiananything_amd import synthetic
    from gaussiananything_amd.decode import SurfelDecoder
    z = torch.load(synthetic.fixture_path("decode_ref.pt"))
    cfg = z["config"]
    model = SurfelDecoder(embed_dim=cfg["D"], depth=cfg["depth"], num_heads=cfg["heads"], tokens=cfg["tokens"],
                          ldm_z_channels=cfg["z_channels"])
    model.load_state_dict(z["state_dict"], strict=True)
    model.to(gpu_device)
    out = model.decode(z["latent"].to(gpu_device), z["xyz"].to(gpu_device))
    lat = model.vit_decode_backbone({"latent_normalized": z["latent"].to(gpu_device)})["latent_from_vit"]
    assert rel_l2(lat.cpu(), z["latent_from_vit"]) < 2e-2
    for k in ("gaussians_base", "gaussians_upsampled", "gaussians_upsampled_2", "gaussians_upsampled_3"):
        got, ref = out[k].cpu(), z[k]
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert rel_l2(got, ref) < 2e-2, (k, rel_l2(got, ref))
        assert float((got[..., :3] - ref[..., :3]).abs().max()) < 4e-3, k
    assert torch.equal(out["gaussians"], out["gaussians_upsampled"]) and out["pos"].shape[-1] == 3
    # the residual streams are updated in place: a second call must see untouched parameters and give the same surfels
    again = model.decode(z["latent"].to(gpu_device), z["xyz"].to(gpu_device))
    assert all(torch.equal(out[k], again[k]) for k in ("gaussians_base", "gaussians_upsampled_3"))
    assert all(torch.equal(v.cpu(), z["state_dict"][k]) for k, v in model.state_dict().items())
    one = model.decode(z["latent"][:1].to(gpu_device), z["xyz"][:1].to(gpu_device))      # batch of one: same as row 0
    assert torch.equal(one["gaussians_upsampled_3"][0], out["gaussians_upsampled_3"][0])
    assert all(torch.equal(v.cpu(), z["state_dict"][k]) for k, v in model.state_dict().items())


def test_triplane_decode_renders_every_level_like_the_oracle(gpu_device):
    """End of the cascade (vit_triplane.py:1550-1591): each decoded level rendered at its own resolution; pixels against
    the CPU raster oracle on the very same surfels (MSE <= 1e-5, the rasterizer's bar)."""
    import numpy as np
    from gaussiananything_amd import synthetic
    from gaussiananything_amd.decode import SurfelDecoder
    from tests import _util
    z = torch.load(synthetic.fixture_path("decode_ref.pt"))
    cfg = z["config"]
    model = SurfelDecoder(embed_dim=cfg["D"], depth=cfg["depth"], num_heads=cfg["heads"], tokens=cfg["tokens"],
                          ldm_z_channels=cfg["z_channels"])
    model.load_state_dict(z["state_dict"])
    model.to(gpu_device)
    ret = model.decode(z["latent"][:1].to(gpu_device), z["xyz"][:1].to(gpu_device))
    cams = synthetic.eval_cameras(2)
    c = {"cam_view": cams["cam_view"][None].to(gpu_device), "cam_view_proj": cams["cam_view_proj"][None].to(gpu_device),
         "cam_pos": cams["cam_pos"][None].to(gpu_device), "tanfov": cams["tanfov"]}
    res = model.triplane_decode(ret, c, render_all_scale=True)
    assert list(res) == ["gaussians_base", "gaussians_upsampled", "gaussians_upsampled_2", "gaussians_upsampled_3"]
    for key, size in model.output_size.items():
        img = res[key]["image"]
        assert img.shape == (1, 2, 3, size, size) and bool(torch.isfinite(img).all())
        assert torch.equal(res[key]["image_raw"], img * 2 - 1) and res[key]["image_mask"].shape == (1, 2, 1, size, size)
        o = _util.oracle_view(ret[key][0].cpu(), cams, 1, size, size)
        mse = float(np.mean((img[0, 1].cpu().numpy() - np.clip(o["color"], 0, 1)) ** 2))
        assert mse <= 1e-5, (key, mse)
    sub = model.triplane_decode(ret, c)                   # the reference's default: one random coarse level + the finest
    assert len(sub) == 2 and "gaussians_upsampled_3" in sub


def test_cascade_equals_its_parts(gpu_device):
    """stage 1 -> x0.164 / clip -> stage 2 (xyz-conditioned) -> decode -> renders, on the small golden models: the driver
    must give exactly what the pieces give when called one after the other."""
    from gaussiananything_amd import cascade, synthetic
    from gaussiananything_amd.decode import SurfelDecoder
    from gaussiananything_amd.dit import DiT_I23D_PCD_PixelArt_noclip, DiT_I23D_PCD_PixelArt_noclip_clay_stage2
    zs = [torch.load(synthetic.fixture_path(f"dit_ref_stage{k}.pt")) for k in (1, 2)]
    m1 = DiT_I23D_PCD_PixelArt_noclip(**zs[0]["kwargs"])
    m1.load_state_dict(zs[0]["state_dict"])
    kw2 = dict(zs[1]["kwargs"], use_pe_cond=True)
    m2 = DiT_I23D_PCD_PixelArt_noclip_clay_stage2(**kw2)
    m2.load_state_dict(zs[1]["state_dict"])
    zd = torch.load(synthetic.fixture_path("decode_ref.pt"))
    cfg = zd["config"]
    dec = SurfelDecoder(embed_dim=cfg["D"], depth=cfg["depth"], num_heads=cfg["heads"], tokens=cfg["tokens"],
                        ldm_z_channels=cfg["z_channels"])
    dec.load_state_dict(zd["state_dict"])
    for m in (m1, m2, dec):
        m.to(gpu_device)
    ctx = zs[0]["context"]
    cond = {k: v[:1].to(gpu_device) for k, v in ctx.items()}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    cams = synthetic.eval_cameras(2)
    c = {"cam_view": cams["cam_view"][None].to(gpu_device), "cam_view_proj": cams["cam_view_proj"][None].to(gpu_device),
         "cam_pos": cams["cam_pos"][None].to(gpu_device), "tanfov": cams["tanfov"]}
    out = cascade.cascade(m1, m2, dec, cond, uc, cameras=c, num_steps=6, sampling_method="euler", seed=3)
    L = cfg["tokens"]
    xyz = cascade.sample(m1, cond, uc, (L, 3), 1, 4.0, 3, 6, "euler")
    fps = (xyz * 0.164).clip(-0.45, 0.45)
    # stage-2 conditioning of the release (sgm/configs/stage2-i23d.yaml): PCD_Scaler feeds the denoiser xyz / 0.45, and
    # with cond_key 'img-xyz' the unconditional branch equals the conditional one; the decoder gets the raw cloud
    c2 = dict(cond, **{"fps-xyz": fps / 0.45})
    lat = cascade.sample(m2, c2, dict(c2), (L, 10), 1, 4.0, 3, 6, "euler")
    ref = dec.decode(lat, fps)
    assert torch.equal(out["query_pcd_xyz"], fps) and torch.equal(out["gaussians_upsampled_3"], ref["gaussians_upsampled_3"])
    assert out["gaussians_upsampled_3"].shape == (1, L * 96, 13)
    assert set(out["renders"]) == set(dec.output_size) and out["renders"]["gaussians_upsampled_3"]["image"].shape[-1] == 512
    assert bool(torch.isfinite(out["renders"]["gaussians_upsampled_3"]["image"]).all())
