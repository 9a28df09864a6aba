"""CPU oracle of the surfel decode (latent + anchor points -> four levels of surfel Gaussians),
plain PyTorch fp32, a restatement of

  pcd_structured_latent_space_vae_decoder.vit_decode_backbone        /root/reference/vit/vit_triplane.py:1415-1427
  ....._get_base_gaussians / _gaussian_pred_activations               vit_triplane.py:1385-1412, 1430-1440
  .....vit_decode_postprocess (base + _cascaded)                      vit_triplane.py:1467-1501, 1645-1676
  constructor constants (activations, skip_weight)                    vit_triplane.py:1289-1313
  surfel_prediction                                                   vit_triplane.py:287-341
  GS_Adaptive_Read_Write_CA_adaptive_2dgs.forward (cross_attention=False)   vit_triplane.py:995-1064
  DiT2.forward (roll_out, in_plane_attention=False) / DiTBlock2       /root/reference/dit/dit_decoder.py:19-35, 99-176
  DiTBlock constructor (LayerNorm no affine eps 1e-6, qk-norm attention, FusedMLP, adaLN)  dit/dit_models_xformers.py:232-289
  SRT Transformer / PreNorm                                           /root/reference/nsr/srt/layers.py:82-92, 146-190
  MemEffAttention.forward                                             /root/reference/vit/vision_transformer.py:215-303
plus the third-party pieces restated in oracle/dit.py (xformers attention, FusedMLP) and timm's Mlp.

TEST INFRASTRUCTURE ONLY.  Works on a reference-format state dict (keys ``vit_decoder.*``, ``superresolution.*``).

PARITY PINNED: tests/golden/decode_ref.pt was produced by the reference's own classes and methods (imported / exec'd
verbatim from /root/reference by tests/golden/make_decode_golden.py; only the decoder's constructor is restated there);
tests/test_cpu_oracle_and_host.py checks this file against it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .dit import fused_mlp, self_attn

BBOX_MAX = 0.45            # rendering_kwargs['sampler_bbox_max']
SKIP_WEIGHT = 0.1
SCENE_EXTENT = BBOX_MAX * 0.01


def offset_act(x):
    return torch.tanh(x) * BBOX_MAX * 0.5


def scale_act(x):
    return F.softplus(x) * (SCENE_EXTENT / F.softplus(torch.tensor(0.0)))


def activate(pos, x):
    """_gaussian_pred_activations: [pos, sigmoid(opacity), softplus-scale(2), normalised quaternion(4), rgb(3)]."""
    return torch.cat([pos, torch.sigmoid(x[..., 3:4]), scale_act(x[..., 4:6]), F.normalize(x[..., 6:10], dim=-1),
                      0.5 * torch.tanh(x[..., 10:]) + 0.5], dim=-1).float()


def dit2_block(sd, p, x, c, heads):
    mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=-1)      # per token
    ln = lambda t: F.layer_norm(t, t.shape[-1:], None, None, 1e-6)
    x = x + gate_msa * self_attn(sd, p + "attn.", ln(x) * (1 + scale_msa) + shift_msa, heads)
    x = x + gate_mlp * fused_mlp(sd, p + "mlp.", ln(x) * (1 + scale_mlp) + shift_mlp)
    return x


def srt_layer(sd, p, x, heads):
    ln = lambda t, q: F.layer_norm(t, t.shape[-1:], sd[q + "norm.weight"], sd[q + "norm.bias"], 1e-5)
    x = self_attn(sd, p + "0.fn.", ln(x, p + "0."), heads) + x
    x = fused_mlp(sd, p + "1.fn.", ln(x, p + "1.")) + x
    return x


def upsample(sd, p, feat, base_gaussians, base_pre, heads):
    """GS_Adaptive_Read_Write_CA_adaptive_2dgs.forward: every anchor becomes f surfels."""
    B, N, D = feat.shape
    emb_q = sd[p + "latent_embedding"]                      # [1, f, D]
    f = emb_q.shape[1]
    tok = torch.cat([feat.reshape(B * N, 1, D), emb_q.expand(B * N, -1, -1)], dim=1)
    depth = 1 + max(int(k[len(p + "transformer.layers."):].split(".")[0]) for k in sd if k.startswith(p + "transformer.layers."))
    for i in range(depth):
        tok = srt_layer(sd, f"{p}transformer.layers.{i}.", tok, heads)
    emb = tok[:, 1:].reshape(B, N, f, D)
    q = p + "gaussian_residual_pred."
    res = F.linear(F.layer_norm(emb, (D,), sd[q + "norm.weight"], sd[q + "norm.bias"], 1e-5), sd[q + "fn.weight"],
                   sd[q + "fn.bias"])                        # [B, N, f, 13]
    pos = offset_act(res[..., :3]) + base_gaussians[..., None, :3]
    res = res + base_pre[..., None, :]
    g = activate(pos, res)
    return g.reshape(B, N * f, 13), res.reshape(B, N * f, 13), emb.reshape(B, N * f, D)


def decode(sd, latent, xyz):
    """latent [B, N, Cz] (KL latent tokens), xyz [B, N, 3] (anchor points)  ->  dict of the four Gaussian levels
    ([B, N*{1, 8, 32, 96}, 13]: xyz, opacity, scale(2), quaternion wxyz(4), rgb(3)) and ``latent_from_vit``."""
    sd = {k: v.float() for k, v in sd.items()}
    latent, xyz = latent.float(), xyz.float()
    D = sd["vit_decoder.pos_embed"].shape[-1]
    heads = D // sd["vit_decoder.blocks.0.attn.q_norm.weight"].shape[0]
    q = "superresolution.post_quant_conv."
    c = F.linear(F.gelu(F.linear(latent, sd[q + "fc1.weight"], sd[q + "fc1.bias"]), approximate="tanh"),
                 sd[q + "fc2.weight"], sd[q + "fc2.bias"])
    x = sd["vit_decoder.pos_embed"].expand(latent.shape[0], -1, -1)
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("vit_decoder.blocks."))
    for i in range(depth):
        x = dit2_block(sd, f"vit_decoder.blocks.{i}.", x, c, heads)
    q = "superresolution.conv_sr.gaussian_pred.1."
    pre = F.linear(F.silu(x), sd[q + "weight"], sd[q + "bias"])
    base = activate(offset_act(pre[..., :3]) * SKIP_WEIGHT + xyz, pre)
    out = {"latent_from_vit": x, "gaussians_base": base}
    feat, g, gpre = x, base, pre
    for name, key in (("ada_CA_f4_1", "gaussians_upsampled"), ("ada_CA_f4_2", "gaussians_upsampled_2"),
                      ("ada_CA_f4_3", "gaussians_upsampled_3")):
        g, gpre, feat = upsample(sd, f"superresolution.{name}.", feat, g, gpre, heads)
        out[key] = g
    return out
