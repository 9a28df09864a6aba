#!/usr/bin/env python3
"""Generates tests/golden/decode_ref.pt from the REFERENCE'S OWN surfel-decode code.
Run once in the build container:  python tests/golden/make_decode_golden.py

What is the reference here and what is not:
  * `DiT2` / `DiTBlock2` (dit/dit_decoder.py), `Transformer` / `PreNorm` (nsr/srt/layers.py), `MemEffAttention`
    (vit/vision_transformer.py) and the ldm attention classes are IMPORTED from /root/reference (third-party xformers /
    timm pieces replaced by the stand-ins of ref_dit_loader.py);
  * `surfel_prediction`, `init_gaussian_prediction` and the upsampler hierarchy `GS_Adaptive_Read_Write_CA` ->
    `_adaptive` -> `_adaptive_f14_prepend` -> `_adaptive_2dgs` are EXEC'D VERBATIM out of vit/vit_triplane.py (the file as
    a whole imports half of the project);
  * the decoder's own methods `_get_base_gaussians`, `_gaussian_pred_activations`, `vit_decode_backbone`,
    `vit_decode_postprocess` (both the base and the cascaded one), `forward_vit_decoder` are likewise exec'd verbatim into
    two shim classes; only the constructor is restated (vit_triplane.py:1274-1343, 1602-1640: activations, the
    `superresolution` ModuleDict) because the real one goes through `vae_3d` / the triplane decoder.
Small shapes (width 128 = 2 heads of 64, 64 anchor tokens, DiT2 depth 2, upsamplers f = 8/4/3 with depth 2/1/1 as in the
release) keep the fixture at a few MB; the arithmetic path is the release's
`pcd_structured_latent_space_vae_decoder_cascaded` with `--arch_dit_decoder DiT2-B/2 --in_plane_attention False`.
"""
import ast
import contextlib
import io
import os
import sys
import textwrap

import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_dit_loader as L  # noqa: E402

REF_FILE = os.path.join(L.REF, "vit/vit_triplane.py")


def _segments():
    src = open(REF_FILE).read()
    tree = ast.parse(src)
    top = {n.name: n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))}

    def seg(node):
        return ast.get_source_segment(src, node)

    def methods(cls, names):
        body = {n.name: n for n in top[cls].body if isinstance(n, ast.FunctionDef)}
        return "\n\n".join(textwrap.indent(textwrap.dedent("    " + seg(body[m])), "    ") for m in names)

    return top, seg, methods


def build_reference_decoder(D=128, tokens=64, heads=2, depth=2, z_channels=10):
    L.install()
    import einops
    from einops import rearrange
    import types
    import dit.dit_decoder as dd
    for pkg in ("nsr", "nsr.srt"):  # nsr/__init__.py imports the trainers (mcubes, ...): bind the bare packages instead
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(L.REF, *pkg.split("."))]
            sys.modules[pkg] = m
    from dit.dit_models_xformers import DiTBlock
    from nsr.srt.layers import PreNorm, Transformer as SRT_TX
    from ldm.modules.attention import CrossAttention, MemoryEfficientCrossAttention
    from timm.models.vision_transformer import Mlp

    top, seg, methods = _segments()
    ns = dict(nn=nn, torch=torch, F=F, einops=einops, rearrange=rearrange, DiTBlock2=dd.DiTBlock2, SRT_TX=SRT_TX,
              PreNorm=PreNorm, CrossAttention=CrossAttention, MemoryEfficientCrossAttention=MemoryEfficientCrossAttention,
              st=lambda: None)
    for name in ("init_gaussian_prediction", "surfel_prediction", "GS_Adaptive_Read_Write_CA",
                 "GS_Adaptive_Read_Write_CA_adaptive", "GS_Adaptive_Read_Write_CA_adaptive_f14_prepend",
                 "GS_Adaptive_Read_Write_CA_adaptive_2dgs"):
        exec(seg(top[name]), ns)

    base_methods = methods("pcd_structured_latent_space_vae_decoder",
                           ["_get_base_gaussians", "vit_decode_backbone", "_gaussian_pred_activations",
                            "vit_decode_postprocess", "forward_gaussians", "forward_vit_decoder"])
    casc_methods = methods("pcd_structured_latent_space_vae_decoder_cascaded", ["vit_decode_postprocess"])
    exec("class RefDecoderBase(nn.Module):\n" + base_methods + "\n\nclass RefDecoder(RefDecoderBase):\n" + casc_methods, ns)

    with contextlib.redirect_stdout(io.StringIO()):
        vit_decoder = dd.DiT2(input_size=16, patch_size=2, in_channels=D, hidden_size=D, depth=depth, num_heads=heads,
                              num_classes=0, learn_sigma=False, mixed_prediction=False, context_dim=None, roll_out=True,
                              plane_n=3, return_all_layers=False, in_plane_attention=False, vit_blk=DiTBlock)
    dec = ns["RefDecoder"]()
    # ---- constructor restated from vit_triplane.py:1274-1343 and :1602-1640 ------------------------------------------
    dec.vit_decoder = vit_decoder
    dec.embed_dim = D
    dec.ldm_z_channels = z_channels
    bbox_max = 0.45                                              # rendering_kwargs['sampler_bbox_max']
    dec.scene_range = [-bbox_max, bbox_max]
    dec.skip_weight = torch.tensor(0.1)
    dec.offset_act = lambda x: torch.tanh(x) * (dec.scene_range[1]) * 0.5
    dec.vit_decoder.pos_embed = nn.Parameter(torch.zeros(1, tokens, D))
    dec.rot_act = lambda x: F.normalize(x, dim=-1)
    dec.scene_extent = bbox_max * 0.01
    scaling_factor = dec.scene_extent / F.softplus(torch.tensor(0.0))
    dec.scale_act = lambda x: F.softplus(x) * scaling_factor
    dec.rgb_act = lambda x: 0.5 * torch.tanh(x) + 0.5
    dec.pos_act = lambda x: x.clamp(-0.45, 0.45)
    dec.opacity_act = lambda x: torch.sigmoid(x)
    approx_gelu = lambda: nn.GELU(approximate="tanh")            # vit_triplane.py:88-89
    up = ns["GS_Adaptive_Read_Write_CA_adaptive_2dgs"]
    dec.superresolution = nn.ModuleDict(dict(
        conv_sr=ns["surfel_prediction"](query_dim=D),
        post_quant_conv=Mlp(in_features=z_channels, out_features=D, act_layer=approx_gelu, drop=0),
        ada_CA_f4_1=up(D, D, vit_heads=heads, mlp_ratio=4, depth=2, f=8, heads=8),
        ada_CA_f4_2=up(D, D, vit_heads=heads, mlp_ratio=4, depth=1, f=4, heads=8, no_flash_op=True, cross_attention=False),
        ada_CA_f4_3=up(D, D, vit_heads=heads, mlp_ratio=4, depth=1, f=3, heads=8, no_flash_op=True, cross_attention=False),
    ))
    return dec


def randomize(dec, seed=5):
    """Seeded, non-degenerate weights everywhere (the reference zero-initialises the residual heads and adaLN)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in dec.named_parameters():
            if "norm" in name and name.endswith("weight") and p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("pos_embed") or "latent_embedding" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / p.shape[-1] ** 0.5))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return dec


def main():
    torch.manual_seed(0)
    cfg = dict(D=128, tokens=64, heads=2, depth=2, z_channels=10)
    dec = randomize(build_reference_decoder(**cfg)).eval()
    g = torch.Generator().manual_seed(3)
    B = 2
    latent = torch.randn(B, cfg["tokens"], cfg["z_channels"], generator=g)
    xyz = (torch.rand(B, cfg["tokens"], 3, generator=g) - 0.5) * 0.9
    with torch.no_grad():
        ret = {"latent_normalized": latent, "query_pcd_xyz": xyz}       # script_util.py:268-275 -> vit_triplane.py:1415-1427
        lat = dec.vit_decode_backbone(ret, None)
        out = dec.forward_gaussians(dec.vit_decode_postprocess(lat, ret))
    keys = ("gaussians_base", "gaussians_upsampled", "gaussians_upsampled_2", "gaussians_upsampled_3")
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    torch.save(dict(config=cfg, state_dict=sd, latent=latent, xyz=xyz, latent_from_vit=lat["latent_from_vit"],
                    **{k: out[k].float() for k in keys}), os.path.join(HERE, "decode_ref.pt"))
    for k in keys:
        print(k, tuple(out[k].shape), f"|.| mean {out[k].abs().mean():.4f}")
    print("params", sum(v.numel() for v in sd.values()), "bytes", os.path.getsize(os.path.join(HERE, "decode_ref.pt")))
    print("state-dict keys (sample):", [k for k in sd][:6], "...", len(sd), "tensors")


if __name__ == "__main__":
    main()
