"""Image conditioner of the cascade: ``FrozenDinov2ImageEmbedder`` -- same class name,
constructor arguments, methods and outputs as /root/reference/sgm/modules/encoders/modules.py:791-931 -- with the DINOv2
ViT (``torch.hub`` 'dinov2_vit{arch}14_reg' upstream) run on the MI355X kernels of the denoise half: ``ga_gemm_bf16`` (patch
embedding as a GEMM over unfolded 14x14 patches, qkv with the transposed-V epilogue, GELU, LayerScale as the gate of the
residual epilogue), ``ga_attention_bf16`` (1374 tokens: cls + 4 registers + 37 x 37 patches at 518 px) and
``ga_layernorm_modulate``.  One image costs ~1 TFLOP, once per sample.  The release builds it from
sgm/configs/img23d-clipl-compat-fm-lognorm-480-uniform-clay-dinoonly.yaml:43-52 (``arch: vitl, inp_size: 518, output_cls:
True, ucg_rate: 0.1``; selected at nsr/lsgm/flow_matching_trainer.py:274-276) -- the native 37 x 37 grid, so no
position-embedding interpolation is involved.

``DinoVisionTransformer`` below is a parameter container with the DINOv2 state-dict layout (``cls_token, pos_embed,
register_tokens, mask_token, patch_embed.proj.*, blocks.{i}.{norm1,norm2}.*, blocks.{i}.attn.{qkv,proj}.*,
blocks.{i}.{ls1,ls2}.gamma, blocks.{i}.mlp.{fc1,fc2}.*, norm.*``), so ``embedder.model.load_state_dict(hub_model.state_dict())``
is a strict load; there is no network here, so construction does not download anything and the weights start random.
Other input sizes than the stored 37 x 37 grid resample the position embeddings as DINOv2 does.
The resize of ``preprocess`` is a handful of torch ops on the device (plumbing, once per sample), not a kernel of ours.

Parity is UNPINNED for this row (the arithmetic is third-party code absent from /root/reference): oracle/dinov2.py restates
the published algorithm and the GPU tests compare against it.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import decode_ops as dops
from . import dit_ops as ops

_ARCH = {"vits": (384, 12, 6), "vitb": (768, 12, 12), "vitl": (1024, 24, 16), "vitg": (1536, 40, 24)}


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _LayerScale(nn.Module):
    def __init__(self, dim, init=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1, self.attn, self.ls1 = nn.LayerNorm(dim, eps=1e-6), _Attn(dim), _LayerScale(dim)
        self.norm2, self.mlp, self.ls2 = nn.LayerNorm(dim, eps=1e-6), _Mlp(dim, int(dim * mlp_ratio)), _LayerScale(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


def _bf(t):
    return t.detach().to(torch.bfloat16).contiguous()


def _f32(t):
    return t.detach().float().contiguous()


class DinoVisionTransformer(nn.Module):
    """dinov2/models/vision_transformer.py::DinoVisionTransformer (ViT with registers), forward_features only."""

    def __init__(self, embed_dim=1024, depth=24, num_heads=16, patch_size=14, img_size=518, num_register_tokens=4,
                 mlp_ratio=4.0):
        super().__init__()
        if embed_dim != num_heads * 64:
            raise NotImplementedError("the MI355X attention kernel is built for head_dim 64 (ViT-S/B/L of DINOv2; not ViT-g)")
        self.embed_dim, self.num_heads, self.patch_size, self.num_register_tokens = embed_dim, num_heads, patch_size, num_register_tokens
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))     # state-dict compatibility; inference never masks
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        nn.init.normal_(self.register_tokens, std=1e-6)
        self._pack = None

    def _packed(self):
        ver = tuple(p._version for p in self.parameters()) + (str(self.cls_token.device),)
        if self._pack is not None and self._pack[0] == ver:
            return self._pack[1]
        D, P = self.embed_dim, self.patch_size
        K = 3 * P * P
        Kp = (K + 63) // 64 * 64                        # the GEMM wants K % 64 == 0: zero columns
        wpe = torch.zeros((D, Kp), dtype=torch.bfloat16, device=self.cls_token.device)
        wpe[:, :K] = self.patch_embed.proj.weight.detach().reshape(D, K).to(torch.bfloat16)
        pk = dict(wpe=wpe, bpe=_f32(self.patch_embed.proj.bias), Kp=Kp, blocks=[], nw=_f32(self.norm.weight), nb=_f32(self.norm.bias))
        for b in self.blocks:
            pk["blocks"].append(dict(
                n1w=_f32(b.norm1.weight), n1b=_f32(b.norm1.bias), qkv_w=_bf(b.attn.qkv.weight), qkv_b=_f32(b.attn.qkv.bias),
                proj_w=_bf(b.attn.proj.weight), proj_b=_f32(b.attn.proj.bias), ls1=_f32(b.ls1.gamma).unsqueeze(0),
                n2w=_f32(b.norm2.weight), n2b=_f32(b.norm2.bias), fc1_w=_bf(b.mlp.fc1.weight), fc1_b=_f32(b.mlp.fc1.bias),
                fc2_w=_bf(b.mlp.fc2.weight), fc2_b=_f32(b.mlp.fc2.bias), ls2=_f32(b.ls2.gamma).unsqueeze(0)))
        self._pack = (ver, pk)
        return pk

    def interpolate_pos_encoding(self, h0, w0):
        """dinov2 DinoVisionTransformer.interpolate_pos_encoding as the register models are built by the hub
        (interpolate_antialias=True, interpolate_offset=0.0) [UPSTREAM-RECALLED]: the stored M x M grid of patch position
        embeddings is resampled bicubically to h0 x w0; the class position is kept.  Host-side plumbing on a parameter."""
        pos = self.pos_embed.detach().float()
        N = pos.shape[1] - 1
        if h0 * w0 == N and h0 == w0:
            return pos
        M = int(round(N ** 0.5))
        grid = pos[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(h0, w0), mode="bicubic", antialias=True)
        return torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, h0 * w0, -1)], dim=1)

    @torch.no_grad()
    def forward_features(self, x, masks=None):
        if masks is not None:
            raise NotImplementedError("masked tokens are a training feature of DINOv2")
        if x.device.type != "cuda":
            raise RuntimeError("gaussiananything_amd conditioner only runs on an MI355X (HIP) device; there is no CPU path")
        pk = self._packed()
        B, _, Hh, Ww = x.shape
        D, H, P, R = self.embed_dim, self.num_heads, self.patch_size, self.num_register_tokens
        n = (Hh // P) * (Ww // P)
        if Hh % P or Ww % P:
            raise ValueError(f"image size {Hh}x{Ww} is not a multiple of the patch size {P}")
        # patch embedding: Conv2d(k = s = P) == GEMM over the unfolded patches ([B*n, 3*P*P], channel-major like the conv weight)
        cols = F.unfold(x.float(), kernel_size=P, stride=P).transpose(1, 2).reshape(B * n, 3 * P * P)
        a = torch.zeros((B * n, pk["Kp"]), dtype=torch.bfloat16, device=x.device)
        a[:, :cols.shape[1]] = cols.to(torch.bfloat16)
        tok = ops.gemm(a, pk["wpe"], pk["bpe"], ops.EPI_STORE_F32).reshape(B, n, D)
        pos = self.interpolate_pos_encoding(Hh // P, Ww // P)
        T = 1 + R + n
        xs = torch.empty((B, T, D), dtype=torch.float32, device=x.device)      # fp32 residual stream, updated in place
        xs[:, 0] = self.cls_token.detach().float()[0, 0] + pos[0, 0]
        xs[:, 1:1 + R] = self.register_tokens.detach().float()
        xs[:, 1 + R:] = tok + pos[:, 1:]
        M, Tp = B * T, (T + 63) // 64 * 64
        x2 = xs.view(M, D)
        for blk in pk["blocks"]:
            h = dops.layernorm_modulate(x2, 1e-6, weight=blk["n1w"], bias=blk["n1b"])
            vt = torch.zeros((B * H * 64, Tp), device=x.device, dtype=torch.bfloat16)
            qk = ops.gemm(h, blk["qkv_w"], blk["qkv_b"], ops.EPI_STORE_BF16, rows_per_batch=T, vt=vt, vt_col0=2 * D)
            q = qk[:, :D].unflatten(0, (B, T)).unflatten(-1, (H, 64))
            k = qk[:, D:2 * D].unflatten(0, (B, T)).unflatten(-1, (H, 64))
            o = ops.attention(q, k, vt).reshape(M, D)
            ops.gemm(o, blk["proj_w"], blk["proj_b"], ops.EPI_RESIDUAL, out=x2, gate=blk["ls1"], rows_per_batch=M)   # LayerScale
            h = dops.layernorm_modulate(x2, 1e-6, weight=blk["n2w"], bias=blk["n2b"])
            hid = ops.gemm(h, blk["fc1_w"], blk["fc1_b"], ops.EPI_GELU_BF16)
            ops.gemm(hid, blk["fc2_w"], blk["fc2_b"], ops.EPI_RESIDUAL, out=x2, gate=blk["ls2"], rows_per_batch=M)
        xn = dops.layernorm_modulate(x2, 1e-6, weight=pk["nw"], bias=pk["nb"]).float().view(B, T, D)
        return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:1 + R], "x_norm_patchtokens": xn[:, 1 + R:],
                "x_prenorm": xs, "masks": masks}

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else ret["x_norm_clstoken"]


class FrozenDinov2ImageEmbedder(nn.Module):
    """sgm/modules/encoders/modules.py:791-931 (same constructor arguments; ``version`` / hub download are not used: load
    weights with ``self.model.load_state_dict``)."""

    def __init__(self, arch="vitl", version="dinov2", device="cuda", max_length=77, freeze=True, antialias=True,
                 ucg_rate=0.0, unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False,
                 output_cls=False, init_device=None, inp_size=224, _vit_kwargs=None):
        super().__init__()
        dim, depth, heads = _ARCH[arch]
        kw = dict(embed_dim=dim, depth=depth, num_heads=heads, patch_size=14, img_size=518, num_register_tokens=4)
        kw.update(_vit_kwargs or {})
        self.model = DinoVisionTransformer(**kw).to(torch.device(init_device or "cpu"))
        self.inp_size = inp_size
        if freeze:
            self.freeze()
        self.max_crops = num_image_crops
        self.pad_to_max_len = self.max_crops > 0
        self.repeat_to_max_len = repeat_to_max_len and (not self.pad_to_max_len)
        self.device = device
        self.max_length = max_length
        self.antialias = antialias
        self.register_buffer("mean", torch.tensor((0.485, 0.456, 0.406)), persistent=False)
        self.register_buffer("std", torch.tensor((0.229, 0.224, 0.225)), persistent=False)
        self.ucg_rate = ucg_rate
        self.unsqueeze_dim = unsqueeze_dim
        self.stored_batch = None
        self.output_tokens = output_tokens
        self.output_cls = output_cls

    def preprocess(self, x):
        """modules.py:864-876: kornia.geometry.resize(bicubic, align_corners=True, antialias) -> [0,1] -> ImageNet norm."""
        x = x.float()
        H, W = x.shape[-2:]
        S = self.inp_size
        if self.antialias and (H > S or W > S):     # kornia: Gaussian pre-blur when shrinking
            C = x.shape[1]
            for axis, L in ((2, H), (3, W)):
                sigma = max((L / S - 1.0) / 2.0, 0.001)
                ks = int(max(4.0 * sigma, 3))
                ks += 1 - ks % 2
                t = torch.arange(ks, dtype=x.dtype, device=x.device) - (ks - 1) / 2.0
                g = torch.exp(-(t * t) / (2.0 * sigma * sigma))
                g = (g / g.sum()).view(1, 1, ks, 1) if axis == 2 else (g / g.sum()).view(1, 1, 1, ks)
                pad = (0, 0, ks // 2, ks // 2) if axis == 2 else (ks // 2, ks // 2, 0, 0)
                x = F.conv2d(F.pad(x, pad, mode="reflect"), g.expand(C, 1, *g.shape[2:]), groups=C)
        if (H, W) != (S, S):
            x = F.interpolate(x, size=(S, S), mode="bicubic", align_corners=True)
        x = (x + 1.0) / 2.0
        return (x - self.mean.view(1, 3, 1, 1).to(x)) / self.std.view(1, 3, 1, 1).to(x)

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    def _model_forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def encode_with_vision_transformer(self, img, **kwargs):
        if img.dim() == 5:
            img = img.flatten(0, 1)             # "b n c h w -> (b n) c h w"
        img = self.preprocess(img)
        if not self.output_cls:
            return self._model_forward(img, is_training=True, **kwargs)["x_norm_patchtokens"]
        ret = self._model_forward(img, is_training=True)
        return ret["x_norm_clstoken"], ret["x_norm_patchtokens"]

    def forward(self, image, no_dropout=False, **kwargs):
        tokens = self.encode_with_vision_transformer(image, **kwargs)
        z = None
        if self.output_cls:
            z, tokens = tokens[0].to(image.dtype), tokens[1]
        tokens = tokens.to(image.dtype)
        if self.ucg_rate > 0.0 and not no_dropout and not (self.max_crops > 0):
            keep = torch.bernoulli((1.0 - self.ucg_rate) * torch.ones(tokens.shape[0], device=tokens.device))
            if z is not None:
                z = keep[:, None] * z
            tokens = keep[:, None, None] * tokens
        return (tokens, z) if self.output_cls else tokens

    def encode(self, image):
        return self(image)
