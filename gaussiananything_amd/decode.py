"""Surfel decode: KL latent tokens + anchor points -> four levels of surfel Gaussians, with the
reference's operator surface and state-dict keys on top of the HIP kernels of include/ga_decode.h / ga_dit.h.

Mirror of ``pcd_structured_latent_space_vae_decoder_cascaded`` (/root/reference/vit/vit_triplane.py:1594-1676 on top of
:1266-1592) for the release configuration ``--arch_dit_decoder DiT2-B/2 --in_plane_attention False``:

    post_quant_conv (Mlp z -> D)  ->  DiT2 backbone: x = pos_embed, per-TOKEN adaLN conditioning on the latent
    (dit/dit_decoder.py:19-35, 99-176)  ->  surfel_prediction head + activations = 768 base surfels  ->  three
    GS_Adaptive_Read_Write_CA_adaptive_2dgs upsamplers (x8, x4, x3; an SRT transformer over [anchor feature | f learned
    queries] per anchor, vit_triplane.py:995-1064)  ->  73 728 surfels.

``vit_decode_backbone`` / ``vit_decode_postprocess`` / ``forward_gaussians`` keep the reference's names, arguments and
dict keys; ``decode`` chains them.  Inference only, CUDA(ROCm) tensors only, head_dim 64; no PyTorch fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import decode_ops as dops
from . import dit_ops as ops
from .dit.dit_i23d import _Attn, _FusedMLP, _Mlp

SKIP_WEIGHT = 0.1   # vit_triplane.py:1290


class _DiTBlock2(nn.Module):  # dit_decoder.py:19-35 over dit_models_xformers.py:232-289 (LayerNorms have no parameters)
    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.attn = _Attn(dim, heads)
        self.mlp = _FusedMLP(dim, int(mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))


class _DiT2(nn.Module):
    def __init__(self, dim, depth, heads, tokens):
        super().__init__()
        self.pos_embed = nn.Parameter(torch.zeros(1, tokens, dim))
        self.blocks = nn.ModuleList([_DiTBlock2(dim, heads) for _ in range(depth)])
        self.embed_dim, self.num_heads, self.depth = dim, heads, depth


class _PreNorm(nn.Module):  # nsr/srt/layers.py:82-92
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _SRT(nn.Module):  # nsr/srt/layers.py:146-190
    def __init__(self, dim, depth, heads):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_PreNorm(dim, _Attn(dim, heads)), _PreNorm(dim, _FusedMLP(dim, 4))])
                                     for _ in range(depth)])


class _Upsampler(nn.Module):  # GS_Adaptive_Read_Write_CA_adaptive_2dgs (cross_attention=False)
    def __init__(self, dim, depth, f):
        super().__init__()
        self.f = f
        self.latent_embedding = nn.Parameter(torch.randn(1, f, dim))
        self.transformer = _SRT(dim, depth, dim // 64)
        self.gaussian_residual_pred = _PreNorm(dim, nn.Linear(dim, 13, bias=True))


class _SurfelPrediction(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gaussian_pred = nn.Sequential(nn.SiLU(), nn.Linear(dim, 13, bias=True))


def _bf(t):
    return t.detach().to(torch.bfloat16).contiguous()


def _f32(t):
    return t.detach().float().contiguous()


class SurfelDecoder(nn.Module):
    """``pcd_structured_latent_space_vae_decoder_cascaded`` for inference.  ``output_size`` is the reference's level ->
    render-resolution table (vit_triplane.py:1299-1301, 1605-1611)."""

    output_size = {"gaussians_base": 128, "gaussians_upsampled": 256, "gaussians_upsampled_2": 384,
                   "gaussians_upsampled_3": 512}

    rand_base_render = True   # the cascaded decoder renders one random coarse level + the finest unless asked for all

    def __init__(self, embed_dim=768, depth=12, num_heads=12, tokens=768, ldm_z_channels=10, triplane_decoder=None):
        super().__init__()
        self.gs = triplane_decoder  # a GaussianRenderer2DGS (no parameters); only triplane_decode needs it
        if embed_dim % 64 or embed_dim // num_heads != 64:
            raise ValueError("the MI355X attention kernels are built for head_dim 64 (DiT2-B/2: 12 heads of 64)")
        self.embed_dim, self.ldm_z_channels = embed_dim, ldm_z_channels
        self.vit_decoder = _DiT2(embed_dim, depth, num_heads, tokens)
        self.superresolution = nn.ModuleDict(dict(
            conv_sr=_SurfelPrediction(embed_dim),
            post_quant_conv=_Mlp(ldm_z_channels, ldm_z_channels, embed_dim),
            ada_CA_f4_1=_Upsampler(embed_dim, depth // 6 if depth == 12 else 2, 8),
            ada_CA_f4_2=_Upsampler(embed_dim, 1, 4),
            ada_CA_f4_3=_Upsampler(embed_dim, 1, 3)))
        self._pack = None

    # ---- weights: packed once into what the kernels read (bf16 matrices, fp32 vectors) ---------------------------
    def _packed(self):
        ver = tuple(p._version for p in self.parameters()) + (str(next(self.parameters()).device),)
        if self._pack is not None and self._pack[0] == ver:
            return self._pack[1]

        def attn(a):
            return dict(qkv_w=_bf(a.qkv.weight), qkv_b=_f32(a.qkv.bias), proj_w=_bf(a.proj.weight), proj_b=_f32(a.proj.bias),
                        qn=_f32(a.q_norm.weight), kn=_f32(a.k_norm.weight))

        def mlp(m):
            return dict(fc1_w=_bf(m.mlp[0].weight), fc1_b=_f32(m.mlp[1].bias), fc2_w=_bf(m.mlp[2].weight),
                        fc2_b=_f32(m.mlp[3].bias))

        pk = dict(pos=_f32(self.vit_decoder.pos_embed[0]), blocks=[], ups=[])
        for b in self.vit_decoder.blocks:
            pk["blocks"].append(dict(attn=attn(b.attn), mlp=mlp(b.mlp), ada_w=_bf(b.adaLN_modulation[1].weight),
                                     ada_b=_f32(b.adaLN_modulation[1].bias)))
        pq = self.superresolution["post_quant_conv"]
        pk["pq"] = [_f32(pq.fc1.weight), _f32(pq.fc1.bias), _f32(pq.fc2.weight), _f32(pq.fc2.bias)]
        head = self.superresolution["conv_sr"].gaussian_pred[1]
        pk["head"] = (_f32(head.weight), _f32(head.bias))
        for name in ("ada_CA_f4_1", "ada_CA_f4_2", "ada_CA_f4_3"):
            u = self.superresolution[name]
            layers = [dict(n0w=_f32(l[0].norm.weight), n0b=_f32(l[0].norm.bias), attn=attn(l[0].fn),
                           n1w=_f32(l[1].norm.weight), n1b=_f32(l[1].norm.bias), mlp=mlp(l[1].fn)) for l in u.transformer.layers]
            pk["ups"].append(dict(f=u.f, emb=_f32(u.latent_embedding[0]), layers=layers,
                                  nw=_f32(u.gaussian_residual_pred.norm.weight), nb=_f32(u.gaussian_residual_pred.norm.bias),
                                  w=_f32(u.gaussian_residual_pred.fn.weight), b=_f32(u.gaussian_residual_pred.fn.bias)))
        self._pack = (ver, pk)
        return pk

    # ---- reference surface -----------------------------------------------------------------------------------------
    def vit_decode_backbone(self, latent, img_size=None):
        """vit_triplane.py:1415-1427: {'latent': post_quant_conv(z) (not materialised: only SiLU(.) is consumed),
        'latent_from_vit': DiT2(pos_embed | z)}."""
        if isinstance(latent, dict):
            latent = latent["latent_normalized"]
        ops._need_cuda(latent)
        pk = self._packed()
        B, N, _ = latent.shape
        D, H = self.embed_dim, self.vit_decoder.num_heads
        M = B * N
        sc = dops.tiny_mlp_silu(latent.reshape(M, -1).float().contiguous(), *pk["pq"])           # bf16 silu(c) [M, D]
        # fp32 residual stream, updated in place by the GEMM epilogues: a fresh copy (for B = 1 the expand / reshape /
        # contiguous chain would hand back the parameter's own storage)
        x = pk["pos"].unsqueeze(0).expand(B, -1, -1).reshape(M, D).clone()
        Lp = (N + 63) // 64 * 64
        for blk in pk["blocks"]:
            mod = ops.gemm(sc, blk["ada_w"], blk["ada_b"], ops.EPI_STORE_F32)                     # [M, 6D] per token
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (mod[:, i * D:(i + 1) * D] for i in range(6))
            a = blk["attn"]
            h = dops.layernorm_modulate(x, 1e-6, scale=scale_msa, shift=shift_msa)
            vt = torch.zeros((B * H * 64, Lp), device=x.device, dtype=torch.bfloat16)
            qk = ops.gemm(h, a["qkv_w"], a["qkv_b"], ops.EPI_STORE_BF16, rows_per_batch=N, vt=vt, vt_col0=2 * D,
                          qk_w0=a["qn"], qk_cols0=D, qk_w1=a["kn"], qk_cols1=2 * D)
            q = qk[:, :D].unflatten(0, (B, N)).unflatten(-1, (H, 64))
            k = qk[:, D:2 * D].unflatten(0, (B, N)).unflatten(-1, (H, 64))
            o = ops.attention(q, k, vt).reshape(M, D)
            ops.gemm(o, a["proj_w"], a["proj_b"], ops.EPI_RESIDUAL, out=x, gate=gate_msa, rows_per_batch=1)
            m = blk["mlp"]
            h = dops.layernorm_modulate(x, 1e-6, scale=scale_mlp, shift=shift_mlp)
            hid = ops.gemm(h, m["fc1_w"], m["fc1_b"], ops.EPI_GELU_BF16)
            ops.gemm(hid, m["fc2_w"], m["fc2_b"], ops.EPI_RESIDUAL, out=x, gate=gate_mlp, rows_per_batch=1)
        return {"latent": None, "latent_from_vit": x.reshape(B, N, D)}

    def _upsample(self, u, src, src_f, P, base_g, base_pre):
        """One GS_Adaptive_Read_Write_CA_adaptive_2dgs level over P anchors: returns (gaussians [P*f,13], pre [P*f,13],
        token stream [P*(1+f), D] whose non-leading rows are the next level's anchor features)."""
        D, H, f = self.embed_dim, self.embed_dim // 64, u["f"]
        S, T = 1 + f, P * (1 + f)
        x = dops.assemble_tokens(src, u["emb"], P, f, src_f)
        for l in u["layers"]:
            a = l["attn"]
            h = dops.layernorm_modulate(x, 1e-5, weight=l["n0w"], bias=l["n0b"])
            qkv = ops.gemm(h, a["qkv_w"], a["qkv_b"], ops.EPI_STORE_BF16, qk_w0=a["qn"], qk_cols0=D, qk_w1=a["kn"],
                           qk_cols1=2 * D)
            o = dops.tiny_attention(qkv, P, S, H)
            ops.gemm(o, a["proj_w"], a["proj_b"], ops.EPI_RESIDUAL, out=x)
            m = l["mlp"]
            h = dops.layernorm_modulate(x, 1e-5, weight=l["n1w"], bias=l["n1b"])
            hid = ops.gemm(h, m["fc1_w"], m["fc1_b"], ops.EPI_GELU_BF16)
            ops.gemm(hid, m["fc2_w"], m["fc2_b"], ops.EPI_RESIDUAL, out=x)
        g, pre = dops.surfel_head(x, u["w"], u["b"], base_g, P * f, 1, f=f, ln_weight=u["nw"], ln_bias=u["nb"],
                                  base_pre=base_pre)
        return g, pre, x

    def vit_decode_postprocess(self, latent_from_vit, ret_dict):
        """vit_triplane.py:1467-1501 + 1645-1676: base surfels and the three upsampled levels."""
        feat = latent_from_vit["latent_from_vit"]
        ops._need_cuda(feat)
        pk = self._packed()
        B, N, D = feat.shape
        xyz = ret_dict["query_pcd_xyz"].reshape(B * N, 3).float().contiguous()
        feat2 = feat.reshape(B * N, D).float().contiguous()
        g, pre = dops.surfel_head(feat2, pk["head"][0], pk["head"][1], xyz, B * N, 0, skip_weight=SKIP_WEIGHT)
        ret_dict = dict(ret_dict)
        ret_dict["gaussians_base"] = g.reshape(B, N, 13)
        src, src_f, P = feat2, 0, B * N
        for u, key in zip(pk["ups"], ("gaussians_upsampled", "gaussians_upsampled_2", "gaussians_upsampled_3")):
            g, pre, src = self._upsample(u, src, src_f, P, g, pre)
            P, src_f = P * u["f"], u["f"]
            ret_dict[key] = g.reshape(B, P // B, 13)
        return ret_dict

    def forward_gaussians(self, ret_after_decoder, c=None):
        """vit_triplane.py:1512-1546."""
        ret_after_decoder["gaussians"] = ret_after_decoder["gaussians_upsampled"]
        ret_after_decoder.update({"pos": ret_after_decoder["gaussians"][..., :3],
                                  "gaussians_base_opa": ret_after_decoder["gaussians_base"][..., 3:4]})
        return ret_after_decoder

    def triplane_decode(self, ret_after_gaussian_forward, c, bg_color=None, render_all_scale=False, **kwargs):
        """vit_triplane.py:1550-1591: render every level at its own resolution (128 / 256 / 384 / 512) for the cameras in
        ``c`` ({cam_view, cam_view_proj [B,V,4,4], cam_pos [B,V,3], tanfov}); all V views of a level are one rasterizer
        call per batch item."""
        if self.gs is None:
            from .gs_surfel import GaussianRenderer2DGS
            self.gs = GaussianRenderer2DGS(output_size=512, out_chans=3, rendering_kwargs={})
        keys = list(self.output_size.keys())
        if self.rand_base_render and not render_all_scale:
            import random
            keys = [random.choice(keys[:-1])] + [keys[-1]]
        out = {}
        sets = [ret_after_gaussian_forward[key] for key in keys]
        if torch.is_grad_enabled() and any(g.requires_grad for g in sets):
            rendered = [self.gs.render(g, c["cam_view"], c["cam_view_proj"], c["cam_pos"], tanfov=c["tanfov"], bg_color=bg_color,
                                       output_size=self.output_size[key]) for g, key in zip(sets, keys)]
        else:   # the levels are independent surfel sets: overlapped on two streams, one overflow read-back for all of them
            rendered = self.gs.render_levels(sets, [self.output_size[key] for key in keys], c["cam_view"], c["cam_view_proj"],
                                             c["cam_pos"], c["tanfov"], bg_color=bg_color)
        for key, res in zip(keys, rendered):
            res["image_raw"] = res["image"] * 2 - 1  # [0,1] -> [-1,1]
            res["image_depth"] = res["depth"]
            res["image_mask"] = res["alpha"]
            out[key] = res
        return out

    @torch.no_grad()
    def decode(self, latent_normalized, query_pcd_xyz):
        """AE.decode_after_vae_no_render_gs (nsr/script_util.py:268-275) for {latent_normalized, query_pcd_xyz}."""
        ret = {"latent_normalized": latent_normalized, "query_pcd_xyz": query_pcd_xyz}
        lat = self.vit_decode_backbone(ret, None)
        return self.forward_gaussians(self.vit_decode_postprocess(lat, ret))
