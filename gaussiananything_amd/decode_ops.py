"""ctypes binding of include/ga_decode.h (same library) plus thin torch-tensor wrappers.  No fallback: every function
needs the HIP library and CUDA(ROCm) tensors."""
from __future__ import annotations

import ctypes

import torch

from . import dit_ops as ops

c_p, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class GaTinyMlpArgs(ctypes.Structure):
    _fields_ = [("M", i32), ("Cin", i32), ("Ch", i32), ("D", i32), ("x", c_p), ("w1", c_p), ("b1", c_p), ("w2", c_p),
                ("b2", c_p), ("out", c_p)]


class GaLayerNormArgs(ctypes.Structure):
    _fields_ = [("M", i32), ("D", i32), ("eps", f32), ("x", c_p), ("weight", c_p), ("bias", c_p), ("scale", c_p),
                ("shift", c_p), ("mod_stride", i64), ("out", c_p)]


class GaAssembleArgs(ctypes.Structure):
    _fields_ = [("P", i32), ("f", i32), ("D", i32), ("src_f", i32), ("src", c_p), ("latent_embedding", c_p), ("out", c_p)]


class GaTinyAttentionArgs(ctypes.Structure):
    _fields_ = [("groups", i32), ("S", i32), ("heads", i32), ("qkv", c_p), ("out", c_p)]


class GaSurfelHeadArgs(ctypes.Structure):
    _fields_ = [("rows", i32), ("D", i32), ("mode", i32), ("f", i32), ("x", c_p), ("ln_weight", c_p), ("ln_bias", c_p),
                ("w", c_p), ("b", c_p), ("anchor", c_p), ("base_pre", c_p), ("skip_weight", f32), ("gaussians", c_p),
                ("pre_out", c_p)]


DECODE_EXPORTS = ("ga_tiny_mlp_silu", "ga_layernorm_modulate", "ga_assemble_tokens", "ga_tiny_attention", "ga_surfel_head")
_bound = False


def lib():
    global _bound
    L = ops.lib()
    if not _bound:
        for name in DECODE_EXPORTS:
            if not hasattr(L, name):
                raise RuntimeError(f"the HIP library does not export {name}")
            getattr(L, name).restype = ctypes.c_int
        _bound = True
    return L


def _call(name, args, ref):
    ops.check(getattr(lib(), name)(ctypes.byref(args), ops._stream(ref)), name)


def tiny_mlp_silu(x, w1, b1, w2, b2):
    """x [M, Cin] fp32 -> bf16 [M, D] = silu(fc2(gelu_tanh(fc1 x)))."""
    ops._need_cuda(x, w1, b1, w2, b2)
    M, Cin = x.shape
    out = torch.empty((M, w2.shape[0]), device=x.device, dtype=torch.bfloat16)
    _call("ga_tiny_mlp_silu", GaTinyMlpArgs(M, Cin, w1.shape[0], w2.shape[0], x.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                             w2.data_ptr(), b2.data_ptr(), out.data_ptr()), x)
    return out


def layernorm_modulate(x, eps, weight=None, bias=None, scale=None, shift=None):
    """x [M, D] fp32 -> bf16; affine and per-row modulation optional (scale/shift: views with row stride, unit column stride)."""
    ops._need_cuda(x, weight, bias, scale, shift)
    M, D = x.shape
    out = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
    if scale is not None:
        assert scale.stride(1) == 1 and shift.stride(1) == 1 and scale.stride(0) == shift.stride(0)
    _call("ga_layernorm_modulate", GaLayerNormArgs(M, D, float(eps), x.data_ptr(), ops._ptr(weight), ops._ptr(bias),
                                                   ops._ptr(scale), ops._ptr(shift),
                                                   scale.stride(0) if scale is not None else 0, out.data_ptr()), x)
    return out


def assemble_tokens(src, latent_embedding, P, f, src_f):
    D = src.shape[-1]
    out = torch.empty((P * (1 + f), D), device=src.device, dtype=torch.float32)
    _call("ga_assemble_tokens", GaAssembleArgs(P, f, D, src_f, src.data_ptr(), latent_embedding.data_ptr(), out.data_ptr()), src)
    return out


def tiny_attention(qkv, groups, S, heads):
    out = torch.empty((groups * S, heads * 64), device=qkv.device, dtype=torch.bfloat16)
    _call("ga_tiny_attention", GaTinyAttentionArgs(groups, S, heads, qkv.data_ptr(), out.data_ptr()), qkv)
    return out


def surfel_head(x, w, b, anchor, rows, mode, f=1, ln_weight=None, ln_bias=None, base_pre=None, skip_weight=0.0):
    g = torch.empty((rows, 13), device=x.device, dtype=torch.float32)
    pre = torch.empty((rows, 13), device=x.device, dtype=torch.float32)
    _call("ga_surfel_head", GaSurfelHeadArgs(rows, x.shape[-1], mode, f, x.data_ptr(), ops._ptr(ln_weight), ops._ptr(ln_bias),
                                             w.data_ptr(), b.data_ptr(), anchor.data_ptr(), ops._ptr(base_pre),
                                             float(skip_weight), g.data_ptr(), pre.data_ptr()), x)
    return g, pre
