"""CPU oracle of the image conditioner: plain PyTorch fp32 restatement of what
``FrozenDinov2ImageEmbedder`` (/root/reference/sgm/modules/encoders/modules.py:791-931) computes:

  preprocess                         modules.py:864-876   kornia.geometry.resize(bicubic, align_corners=True, antialias) ->
                                                           (x + 1) / 2 -> ImageNet mean / std normalisation
  encode_with_vision_transformer     modules.py:886-908   model(img, is_training=True)['x_norm_patchtokens'] (+ 'x_norm_clstoken')
  forward                            modules.py:910-931   (tokens, cls) when output_cls

TEST INFRASTRUCTURE ONLY.  Works on a DINOv2-format state dict (keys ``cls_token, pos_embed, register_tokens,
patch_embed.proj.*, blocks.{i}.{norm1,norm2}.*, blocks.{i}.attn.{qkv,proj}.*, blocks.{i}.{ls1,ls2}.gamma,
blocks.{i}.mlp.{fc1,fc2}.*, norm.*``).

PARITY: the ENCODER (vit_forward) is PINNED against an independent published implementation of the same architecture --
Hugging Face transformers' Dinov2WithRegistersModel with the same randomly drawn weights mapped to the hub key layout, max
difference 2e-5 of the output range (tests/test_cpu_oracle_and_host.py::test_dinov2_oracle_against_the_transformers_implementation);
the kornia resize in front of it and the torch.hub code / released weights themselves stay UNPINNED.  The arithmetic lives entirely in
third-party code that is absent from /root/reference and cannot be fetched here: the model is ``torch.hub.load('facebookresearch/dinov2', 'dinov2_vitl14_reg')`` (modules.py:816-822, hub
ref unpinned; the comments at :841 and :895 cite commit e1277af2ba9496fbadf7aec6eba56e8d882d1e35) and the resize is kornia
(requirements.txt, unpinned).  What is restated below, from the published DINOv2 code [UPSTREAM-RECALLED]:
  dinov2/models/vision_transformer.py  DinoVisionTransformer.prepare_tokens_with_masks / forward_features:
      x = patch_embed(img) (Conv2d k = s = 14, flattened row-major);  x = cat(cls_token, x) + pos_embed  (no interpolation
      when the patch grid equals the stored one, else bicubic + antialias resampling of the patch grid, offset 0);  x = cat(x[:, :1], register_tokens, x[:, 1:]);  blocks;  x_norm = norm(x);
      x_norm_clstoken = x_norm[:, 0], x_norm_regtokens = x_norm[:, 1:1+R], x_norm_patchtokens = x_norm[:, 1+R:]
  dinov2/layers/block.py               x = x + ls1(attn(norm1(x)));  x = x + ls2(mlp(norm2(x)))   (LayerNorm eps 1e-6)
  dinov2/layers/attention.py           qkv Linear(bias) -> heads -> softmax(q k^T / sqrt(d)) v -> proj Linear
  dinov2/layers/mlp.py / layer_scale   fc2(GELU_erf(fc1 x));  x * gamma
and kornia.geometry.transform.resize: when antialias and the image shrinks, a Gaussian blur with sigma = max((factor-1)/2,
0.001) per axis and kernel size max(int(4 sigma), 3) made odd (reflect border) precedes F.interpolate(bicubic,
align_corners=True).  There is no golden vector to pin any of this against; tests compare the HIP path with THIS file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _gauss_kernel1d(ks, sigma, dtype, device):
    x = torch.arange(ks, dtype=dtype, device=device) - (ks - 1) / 2.0
    k = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def resize_bicubic(x, size, antialias=True):
    """kornia.geometry.resize(x, (size, size), 'bicubic', align_corners=True, antialias=antialias)  [UPSTREAM-RECALLED]"""
    H, W = x.shape[-2:]
    if antialias and (H > size or W > size):
        fy, fx = H / size, W / size
        sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
        ky, kx = int(max(2.0 * 2.0 * sy, 3)), int(max(2.0 * 2.0 * sx, 3))
        ky, kx = ky + (1 - ky % 2), kx + (1 - kx % 2)
        C = x.shape[1]
        gy = _gauss_kernel1d(ky, sy, x.dtype, x.device).view(1, 1, ky, 1).expand(C, 1, ky, 1)
        gx = _gauss_kernel1d(kx, sx, x.dtype, x.device).view(1, 1, 1, kx).expand(C, 1, 1, kx)
        x = F.conv2d(F.pad(x, (0, 0, ky // 2, ky // 2), mode="reflect"), gy, groups=C)
        x = F.conv2d(F.pad(x, (kx // 2, kx // 2, 0, 0), mode="reflect"), gx, groups=C)
    if (H, W) == (size, size):
        return x
    return F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)


def preprocess(x, inp_size, antialias=True):
    """modules.py:864-876: images in [-1, 1] -> resized, ImageNet-normalised."""
    x = resize_bicubic(x.float(), inp_size, antialias)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


def vit_forward(sd, img):
    """DinoVisionTransformer.forward_features on a preprocessed image batch [B,3,S,S] (S = patch * grid)."""
    sd = {k: v.float() for k, v in sd.items()}
    w = sd["patch_embed.proj.weight"]
    D, _, P, _ = w.shape
    heads = D // 64
    x = F.conv2d(img.float(), w, sd["patch_embed.proj.bias"], stride=P).flatten(2).transpose(1, 2)      # [B, n, D]
    B, n, _ = x.shape
    pos = sd["pos_embed"]
    if pos.shape[1] != n + 1:   # interpolate_pos_encoding of the register models (antialias=True, offset 0) [UPSTREAM-RECALLED]
        Mg, g0 = int(round((pos.shape[1] - 1) ** 0.5)), img.shape[-1] // P
        grid = F.interpolate(pos[:, 1:].reshape(1, Mg, Mg, D).permute(0, 3, 1, 2), size=(img.shape[-2] // P, g0), mode="bicubic",
                             antialias=True)
        pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1) + pos
    R = sd["register_tokens"].shape[1]
    x = torch.cat([x[:, :1], sd["register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, -1, 3, heads, 64).permute(2, 0, 3, 1, 4)
        att = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * 64 ** -0.5, dim=-1) @ qkv[2]               # [B, h, T, 64]
        a = F.linear(att.transpose(1, 2).reshape(B, -1, D), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + a * sd[p + "ls1.gamma"]
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        m = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                     sd[p + "mlp.fc2.bias"])
        x = x + m * sd[p + "ls2.gamma"]
    xn = F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:1 + R], "x_norm_patchtokens": xn[:, 1 + R:], "x_prenorm": x}


def embed(sd, image, inp_size, antialias=True):
    """FrozenDinov2ImageEmbedder.forward with output_cls=True: (patch tokens [B, n, D], cls [B, D])."""
    out = vit_forward(sd, preprocess(image, inp_size, antialias))
    return out["x_norm_patchtokens"], out["x_norm_clstoken"]
