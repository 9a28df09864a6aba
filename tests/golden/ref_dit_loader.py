"""Imports the REFERENCE's own DiT code (/root/reference/dit/*.py, vit/vision_transformer.py, ldm/modules/attention.py,
dit/norm.py, transport/*.py) on a CPU-only host so that golden vectors can be generated from it.

Only used by tests/golden/make_dit_golden.py, in the build container (needs /root/reference).  The reference cannot be
imported as-is here: it needs xformers, timm and, through vit/vit_triplane.py, half of the project.
This module installs the SMALLEST possible stand-ins for the third-party pieces, written from their published
semantics, and leaves every line of the reference's own model code untouched:

  xformers.ops.memory_efficient_attention(q, k, v)  -> softmax(q k^T / sqrt(d)) v on [B, M, H, K] (or [B, M, K]) inputs
  xformers FusedMLP(dim, dropout=0, GeLU, mult)     -> Linear(no bias) -> +bias, exact-erf GELU -> Linear(no bias) -> +bias
                                                       (state-dict keys mlp.0.weight, mlp.1.bias, mlp.2.weight, mlp.3.bias)
  timm Mlp / PatchEmbed                             -> fc1 -> act -> fc2 ; Conv2d patchify (unused by the PCD models)
  vit.vit_triplane.XYZPosEmbed                      -> the class source is exec'd verbatim out of the reference file
  torchdiffeq.odeint                                -> fixed-grid euler / heun + dopri5 (see oracle/ode.py), for transport
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, scale=None, op=None):
    assert attn_bias is None and p == 0.0
    if q.dim() == 3:  # [B, M, K]
        q, k, v = q.unsqueeze(2), k.unsqueeze(2), v.unsqueeze(2)
        return _memory_efficient_attention(q, k, v, scale=scale).squeeze(2)
    d = q.shape[-1]
    s = scale if scale is not None else d ** -0.5
    qf, kf, vf = (t.permute(0, 2, 1, 3).float() for t in (q, k, v))  # B H M K
    att = torch.softmax(qf @ kf.transpose(-1, -2) * s, dim=-1)
    return (att @ vf).permute(0, 2, 1, 3).to(q.dtype)


class _FusedDropoutBias(nn.Module):
    def __init__(self, p, bias_shape, activation):
        super().__init__()
        assert p == 0
        self.bias = nn.Parameter(torch.zeros(bias_shape))
        self.activation = activation

    def forward(self, x):
        x = x + self.bias
        return torch.nn.functional.gelu(x) if self.activation == "gelu" else x


class _FusedMLP(nn.Module):
    def __init__(self, dim_model, dropout, activation, hidden_layer_multiplier, bias=True, *a, **k):
        super().__init__()
        dim_mlp = hidden_layer_multiplier * dim_model
        self.mlp = nn.Sequential(
            nn.Linear(dim_model, dim_mlp, bias=False), _FusedDropoutBias(dropout, dim_mlp, activation),
            nn.Linear(dim_mlp, dim_model, bias=False), _FusedDropoutBias(dropout, dim_model, None))

    def forward(self, x):
        return self.mlp(x)


class _Mlp(nn.Module):  # timm.layers.Mlp
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0, **k):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class _PatchEmbed(nn.Module):  # timm PatchEmbed (constructed then deleted by the PCD models)
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True, **k):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def install():
    """Put /root/reference on sys.path with the stand-ins in place; returns the reference's dit.dit_i23d module."""
    if "dit.dit_i23d" in sys.modules:
        return sys.modules["dit.dit_i23d"]
    if REF not in sys.path:
        sys.path.insert(0, REF)

    class _Op:  # attention "op" selectors are only passed through
        pass

    xops = _mod("xformers.ops", memory_efficient_attention=_memory_efficient_attention, unbind=torch.unbind,
                fmha=types.SimpleNamespace(), MemoryEfficientAttentionFlashAttentionOp=_Op,
                MemoryEfficientAttentionCutlassOp=_Op)
    xf = _mod("xformers", ops=xops, __version__="0.0.22")
    act = _mod("xformers.components.activations", build_activation=None,
               Activation=types.SimpleNamespace(GeLU="gelu"))
    fm = _mod("xformers.components.feedforward.fused_mlp", FusedMLP=_FusedMLP)
    ff = _mod("xformers.components.feedforward", fused_mlp=fm)
    _mod("xformers.components", activations=act, feedforward=ff)
    xf.components = sys.modules["xformers.components"]
    tv = _mod("timm.models.vision_transformer", PatchEmbed=_PatchEmbed, Mlp=_Mlp)
    tm = _mod("timm.models", vision_transformer=tv)
    _mod("timm", models=tm)

    # vit.vit_triplane pulls in half of the project: provide a module that only holds XYZPosEmbed, exec'd verbatim
    # from the reference source (class body = vit/vit_triplane.py:187-229)
    nerf_utils = importlib.import_module("utils.nerf_utils")
    src = open(os.path.join(REF, "vit/vit_triplane.py")).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("class XYZPosEmbed"))
    end = next(i for i in range(start + 1, len(src)) if src[i].startswith("class ") or src[i].startswith("def "))
    ns = {"nn": nn, "torch": torch, "get_embedder": nerf_utils.get_embedder}
    exec("\n".join(src[start:end]), ns)
    import vit  # the reference package (vit/__init__.py)
    vt = _mod("vit.vit_triplane", XYZPosEmbed=ns["XYZPosEmbed"])
    vit.vit_triplane = vt

    m = importlib.import_module("dit.dit_i23d")
    # dit_models_xformers only binds fused_mlp / Activation when CUDA is available (dit_models_xformers.py:40-43)
    dmx = sys.modules["dit.dit_models_xformers"]
    dmx.fused_mlp = fm
    dmx.Activation = act.Activation
    return m


def rerandomize_zero_init(model, std=0.02, seed=1234):
    """The reference zero-initialises final_layer.linear, adaLN_modulation and pooled_vec_embedder (outputs would be
    exactly 0 without a checkpoint): re-draw every all-zero weight/bias from N(0, std)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.numel() > 1 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
    return model
