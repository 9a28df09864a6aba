"""CPU oracle of the DiT/SiT denoiser forward (plain PyTorch, fp32), a restatement of

  DiT_I23D_PCD_PixelArt_noclip.forward               /root/reference/dit/dit_i23d.py:511-567      (stage 1)
  DiT_I23D_PCD_PixelArt_noclip_clay_stage2.forward   /root/reference/dit/dit_i23d.py:707-750      (stage 2, use_pe_cond)
  DiT_I23D.forward_with_cfg                          /root/reference/dit/dit_i23d.py:159-172
  ImageCondDiTBlockPixelArtRMSNormClayLRM.forward    /root/reference/dit/dit_models_xformers.py:765-787
  MemEffAttention.forward                            /root/reference/vit/vision_transformer.py:215-303
  MemoryEfficientCrossAttention.forward              /root/reference/ldm/modules/attention.py:484-561
  RMSNorm.forward                                    /root/reference/dit/norm.py:29-43
  TimestepEmbedder / T2IFinalLayer / t2i_modulate    /root/reference/dit/dit_models_xformers.py:88-128,62-85,53-54
  XYZPosEmbed + get_embedder                         /root/reference/vit/vit_triplane.py:187-229, utils/nerf_utils.py:16-66
and of the third-party pieces those call (xformers memory_efficient_attention = softmax(QK^T/sqrt(d))V, xformers
FusedMLP = Linear -> +bias, erf-GELU -> Linear -> +bias, timm Mlp = fc2(act(fc1 x))).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).  It works directly on a
reference-format ``state_dict`` (same keys as the reference classes produce).

PARITY PINNED: tests/golden/dit_ref_*.pt hold inputs, state dicts and outputs produced by the REFERENCE'S OWN model code
imported from /root/reference (tests/golden/make_dit_golden.py; third-party xformers/timm pieces replaced by the
stand-ins documented in tests/golden/ref_dit_loader.py); tests/test_cpu_oracle_and_host.py (test_dit_oracle_matches_reference_golden) checks this file against them.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps=1e-5):
    var = x.float().pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * weight


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def xyz_embed(xyz, multires=10):
    out = [xyz]
    for k in range(multires):
        f = 2.0 ** k
        out += [torch.sin(xyz * f), torch.cos(xyz * f)]
    return torch.cat(out, -1)


def _heads(t, h):
    b, l, c = t.shape
    return t.reshape(b, l, h, c // h).permute(0, 2, 1, 3)  # B H L d


def attention(q, k, v):
    d = q.shape[-1]
    att = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    return att @ v


def self_attn(sd, p, x, h):
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    b, l, c3 = qkv.shape
    q, k, v = qkv.reshape(b, l, 3, h, c3 // 3 // h).unbind(2)          # B L H d
    q, k = rmsnorm(q, sd[p + "q_norm.weight"]), rmsnorm(k, sd[p + "k_norm.weight"])
    o = attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))
    o = o.permute(0, 2, 1, 3).reshape(b, l, c3 // 3)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def cross_attn(sd, p, x, ctx, h):
    q = _heads(F.linear(x, sd[p + "to_q.weight"]), h)
    k = _heads(F.linear(ctx, sd[p + "to_k.weight"]), h)
    v = _heads(F.linear(ctx, sd[p + "to_v.weight"]), h)
    q, k = rmsnorm(q, sd[p + "q_norm.weight"]), rmsnorm(k, sd[p + "k_norm.weight"])
    o = attention(q, k, v).permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], -1)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def fused_mlp(sd, p, x):
    hdn = F.gelu(F.linear(x, sd[p + "mlp.0.weight"]) + sd[p + "mlp.1.bias"])          # exact erf GELU
    return F.linear(hdn, sd[p + "mlp.2.weight"]) + sd[p + "mlp.3.bias"]


def block(sd, i, x, t0, ctx, h):
    p = f"blocks.{i}."
    b = x.shape[0]
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = \
        (sd[p + "scale_shift_table"][None] + t0.reshape(b, 6, -1)).chunk(6, dim=1)
    x = x + cross_attn(sd, p + "cross_attn_dino.", rmsnorm(x, sd[p + "prenorm_ca_dino.weight"]), ctx, h)
    x = x + gate_msa * self_attn(sd, p + "attn.", rmsnorm(x, sd[p + "norm1.weight"]) * (1 + scale_msa) + shift_msa, h)
    x = x + gate_mlp * fused_mlp(sd, p + "mlp.", rmsnorm(x, sd[p + "norm2.weight"]) * (1 + scale_mlp) + shift_mlp)
    return x


def config_from_state_dict(sd):
    dm = sd["t_embedder.mlp.2.weight"].shape[0]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    d_head = sd["blocks.0.attn.q_norm.weight"].shape[0]
    return dict(hidden_size=dm, depth=depth, num_heads=dm // d_head, in_channels=sd["x_embedder.fc1.weight"].shape[1],
                out_channels=sd["final_layer.linear.weight"].shape[0],
                context_dim=sd["blocks.0.cross_attn_dino.to_k.weight"].shape[1],
                stage2="xyz_pos_embed.xyz_projection.weight" in sd)


def dit_forward(sd, x, timesteps, context):
    """x [B,L,C], timesteps [B] in [0,1], context {img_crossattn [B,M,ctx], img_vector [B,ctx], (fps-xyz [B,L,3])}."""
    cfg = config_from_state_dict(sd)
    h = cfg["num_heads"]
    x = x.float()
    ctx = context["img_crossattn"].float()
    vec = context["img_vector"].float()
    te = timestep_embedding(timesteps)
    te = F.linear(F.silu(F.linear(te, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])),
                  sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    pv = F.layer_norm(vec, vec.shape[-1:], sd["pooled_vec_embedder.0.weight"], sd["pooled_vec_embedder.0.bias"], 1e-5)
    t = te + F.linear(pv, sd["pooled_vec_embedder.1.weight"], sd["pooled_vec_embedder.1.bias"])
    t0 = F.linear(F.silu(t), sd["adaLN_modulation.1.weight"], sd["adaLN_modulation.1.bias"])
    xe = F.linear(F.gelu(F.linear(x, sd["x_embedder.fc1.weight"], sd["x_embedder.fc1.bias"]), approximate="tanh"),
                  sd["x_embedder.fc2.weight"], sd["x_embedder.fc2.bias"])
    if cfg["stage2"]:
        xe = xe + F.linear(xyz_embed(context["fps-xyz"].float()), sd["xyz_pos_embed.xyz_projection.weight"],
                           sd["xyz_pos_embed.xyz_projection.bias"])
    x = xe
    for i in range(cfg["depth"]):
        x = block(sd, i, x, t0, ctx, h)
    shift, scale = (sd["final_layer.scale_shift_table"][None] + t[:, None]).chunk(2, dim=1)
    x = F.layer_norm(x, x.shape[-1:], None, None, 1e-6) * (1 + scale) + shift
    return F.linear(x, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"]).float()


def forward_with_cfg(sd, x, t, context, cfg_scale):
    eps = dit_forward(sd, x, t, context)
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half = uncond + cfg_scale * (cond - uncond)
    return torch.cat([half, half], dim=0)
