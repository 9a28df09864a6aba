"""ctypes binding of include/ga_dit.h (same library as the surfel rasterizer) plus thin torch-tensor wrappers of the
per-op entry points.  No fallback: every function needs the HIP library and CUDA(ROCm) tensors."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

c_p = ctypes.c_void_p
i32, i64, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float

EPI_STORE_BF16, EPI_GELU_BF16, EPI_RESIDUAL, EPI_STORE_F32 = 0, 1, 2, 3


class GaGemmArgs(ctypes.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("epilogue", i32), ("A", c_p), ("lda", i64), ("W", c_p),
                ("bias", c_p), ("out", c_p), ("ldo", i64), ("gate", c_p), ("gate_stride", i64), ("rows_per_batch", i32),
                ("vt", c_p), ("vt_col0", i32), ("vt_ld", i64), ("qk_w0", c_p), ("qk_w1", c_p), ("qk_cols0", i32),
                ("qk_cols1", i32), ("emit_x", c_p), ("emit_ss", c_p), ("emit_ld", i64), ("row_ss", c_p),
                ("row_ss_tiles", i32), ("row_ss_dim", i32), ("row_ss_eps", ctypes.c_float), ("w_tiled", i32),
                ("emit_w", c_p), ("emit_scale", c_p), ("emit_scale_stride", i64), ("bias_stride", i64), ("k_rows", i32),
                ("splitk_ws", c_p), ("splitk_ws_bytes", i64)]


class GaAttentionArgs(ctypes.Structure):
    _fields_ = [("batch", i32), ("heads", i32), ("Lq", i32), ("Lk", i32), ("q", c_p), ("k", c_p), ("vt", c_p),
                ("q_stride", i64), ("k_stride", i64), ("vt_ld", i64), ("q_norm_weight", c_p), ("k_norm_weight", c_p),
                ("out", c_p), ("out_stride", i64), ("qp_a", c_p), ("qp_w", c_p), ("qp_lda", i64), ("qp_k", i32), ("qp_w_tiled", i32),
                ("qp_row_ss", c_p), ("qp_row_ss_tiles", i32), ("qp_row_ss_dim", i32), ("qp_row_ss_eps", ctypes.c_float)]


class GaAttentionHdArgs(ctypes.Structure):
    _fields_ = [("batch", i32), ("heads", i32), ("Lq", i32), ("Lk", i32), ("head_dim", i32), ("q", c_p), ("k", c_p), ("v", c_p),
                ("q_stride", i64), ("k_stride", i64), ("v_stride", i64), ("out", c_p), ("out_stride", i64),
                ("vt", c_p), ("vt_ld", i64), ("q_norm_weight", c_p), ("k_norm_weight", c_p)]


class GaRmsNormArgs(ctypes.Structure):
    _fields_ = [("M", i32), ("D", i32), ("rows_per_batch", i32), ("x", c_p), ("weight", c_p), ("scale", c_p),
                ("shift", c_p), ("mod_stride", i64), ("out", c_p), ("row_bias", c_p), ("row_bias_first", i32)]


class GaSmallLinearArgs(ctypes.Structure):
    _fields_ = [("B", i32), ("N", i32), ("K", i32), ("act_in", i32), ("act_out", i32), ("x", c_p), ("W", c_p),
                ("bias", c_p), ("add", c_p), ("y", c_p)]


class GaDitBlockWeights(ctypes.Structure):
    _fields_ = [(n, c_p) for n in (
        "prenorm_ca_w", "ca_q_w", "ca_q_w_prenorm", "ca_kv_w", "ca_q_norm_w", "ca_k_norm_w", "ca_out_w", "ca_out_b", "norm1_w", "qkv_w",
        "qkv_b", "q_norm_w", "k_norm_w", "proj_w", "proj_b", "norm2_w", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
        "scale_shift_table")]


class GaDitModel(ctypes.Structure):
    _fields_ = [("hidden", i32), ("depth", i32), ("heads", i32), ("in_channels", i32), ("out_channels", i32),
                ("context_dim", i32), ("stage2", i32)] + [(n, c_p) for n in (
                    "t_mlp0_w", "t_mlp0_b", "t_mlp2_w", "t_mlp2_b", "pool_ln_w", "pool_ln_b", "pool_w", "pool_b",
                    "adaln_w", "adaln_b", "xe_fc1_w", "xe_fc1_b", "xe_fc2_w", "xe_fc2_b", "xyz_w", "xyz_b",
                    "final_table", "final_w", "final_b")] + [("blocks", ctypes.POINTER(GaDitBlockWeights)),
                                                              ("gemm_weights_tiled", i32)]


class GaDitSamplerStep(ctypes.Structure):
    """include/ga_dit.h: GaDitSamplerStep"""
    _fields_ = [("cfg_scale", ctypes.c_float), ("cfg", i32), ("dt", c_p), ("state", c_p), ("traj", c_p),
                ("traj_stride", ctypes.c_int64), ("counter", c_p), ("velocity", c_p)]


class GaOdeDopri5(ctypes.Structure):
    """include/ga_dit.h: GaOdeDopri5"""
    _fields_ = [("n", ctypes.c_int64), ("batch", i32), ("grid_len", i32), ("y", c_p), ("k", c_p * 7), ("ystage", c_p), ("timesteps", c_p),
                ("ctl", c_p), ("t_grid", c_p), ("out", c_p), ("ctl_words", ctypes.c_int64)]


# indices into GaOdeDopri5.ctl (include/ga_dit.h)
(GA_ODE_T, GA_ODE_DT, GA_ODE_SUMSQ, GA_ODE_ATOL, GA_ODE_RTOL, GA_ODE_DONE, GA_ODE_STEPS, GA_ODE_REJECTED, GA_ODE_ACCEPT, GA_ODE_TA,
 GA_ODE_TB, GA_ODE_DT_USED, GA_ODE_JNEXT, GA_ODE_JBEG, GA_ODE_JCOUNT, GA_ODE_ERROR, GA_ODE_RATIO) = range(17)
GA_ODE_CTL_WORDS = 24          # the scalar head the host reads back
GA_ODE_CTL_ALLOC = 24 + 2048    # + one error-norm partial per workgroup of the error launch (GA_ODE_MAX_PARTIALS)


class GaDitForwardArgs(ctypes.Structure):
    _fields_ = [("batch", i32), ("tokens", i32), ("ctx_tokens", i32), ("x", c_p), ("timesteps", c_p),
                ("img_vector", c_p), ("fps_xyz", c_p), ("ca_k", c_p), ("ca_vt", c_p), ("out", c_p), ("workspace", c_p),
                ("workspace_bytes", ctypes.c_size_t), ("ca_batch", i32), ("step", ctypes.POINTER(GaDitSamplerStep)),
                ("pooled_vec", c_p)]


DIT_EXPORTS = ("ga_gemm_bf16", "ga_attention_bf16", "ga_attention_hd_bf16", "ga_head_rmsnorm_bf16", "ga_rmsnorm_modulate", "ga_small_linear", "ga_dit_workspace_bytes",
               "ga_dit_cache_context", "ga_dit_forward", "ga_dit_pooled_vector", "ga_dit_shift_bias", "ga_dit_sampler_advance", "ga_ode_dopri5_stage", "ga_ode_dopri5_finish",
               "ga_dit_version", "ga_gemm_splitk_workspace_bytes", "ga_gemm_splitk_mode")
_ERR = {-1: "GA_DIT_ERR_NULL_ARG", -2: "GA_DIT_ERR_BAD_SHAPE", -4: "GA_DIT_ERR_LAUNCH"}
_bound = False


def lib():
    global _bound
    L = _lib.lib()
    if not _bound:
        for name in DIT_EXPORTS:
            if not hasattr(L, name):
                raise RuntimeError(f"{_lib.LIB_PATH} does not export {name}")
        L.ga_dit_version.restype = ctypes.c_char_p
        L.ga_dit_workspace_bytes.restype = ctypes.c_size_t
        L.ga_dit_workspace_bytes.argtypes = [ctypes.POINTER(GaDitModel), i32, i32, i32]
        for name in ("ga_gemm_bf16", "ga_attention_bf16", "ga_rmsnorm_modulate", "ga_small_linear"):
            getattr(L, name).restype = ctypes.c_int
        L.ga_dit_cache_context.restype = ctypes.c_int
        L.ga_dit_cache_context.argtypes = [ctypes.POINTER(GaDitModel), i32, i32, c_p, c_p, c_p, c_p]
        L.ga_dit_forward.restype = ctypes.c_int
        L.ga_dit_forward.argtypes = [ctypes.POINTER(GaDitModel), ctypes.POINTER(GaDitForwardArgs), c_p]
        L.ga_head_rmsnorm_bf16.restype = ctypes.c_int
        L.ga_head_rmsnorm_bf16.argtypes = [c_p, i64, i64, i32, i32, c_p, c_p]
        L.ga_dit_pooled_vector.restype = ctypes.c_int
        L.ga_dit_pooled_vector.argtypes = [ctypes.POINTER(GaDitModel), i32, c_p, c_p, c_p, c_p]
        L.ga_dit_sampler_advance.restype = ctypes.c_int
        L.ga_dit_sampler_advance.argtypes = [c_p, c_p, c_p, i32, c_p, i32, c_p, c_p]
        L.ga_dit_shift_bias.restype = ctypes.c_int
        L.ga_dit_shift_bias.argtypes = [c_p, i32, c_p, i32, i32, c_p, i64, i32, c_p, c_p]
        L.ga_ode_dopri5_stage.restype = ctypes.c_int
        L.ga_ode_dopri5_stage.argtypes = [ctypes.POINTER(GaOdeDopri5), i32, c_p]
        L.ga_ode_dopri5_finish.restype = ctypes.c_int
        L.ga_ode_dopri5_finish.argtypes = [ctypes.POINTER(GaOdeDopri5), c_p]
        L.ga_gemm_splitk_workspace_bytes.restype = ctypes.c_size_t
        L.ga_gemm_splitk_workspace_bytes.argtypes = [i32, i32]
        L.ga_gemm_splitk_mode.restype = ctypes.c_int
        L.ga_gemm_splitk_mode.argtypes = [ctypes.c_int]
        _bound = True
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


def _stream(t):
    return c_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("gaussiananything_amd DiT ops only run on an MI355X (HIP) device; there is no CPU path")


def gemm(A, W, bias=None, epilogue=EPI_STORE_BF16, out=None, gate=None, rows_per_batch=1, vt=None, vt_col0=0,
         qk_w0=None, qk_cols0=0, qk_w1=None, qk_cols1=0, emit_x=None, emit_ss=None, row_ss=None, row_ss_dim=0,
         row_ss_eps=1e-5, w_tiled=False, N=None, emit_w=None, emit_scale=None, k_rows=0, splitk_ws=None):
    """A [M,K] bf16, W [N,K] bf16 -> see ga_dit.h.  EPI_RESIDUAL accumulates into ``out`` (fp32 [M,N]).
    ``vt`` [B*heads*64, Lpad] bf16 (zero-initialised): columns >= vt_col0 are stored transposed there (V projection).
    ``qk_w0/qk_w1``: per-head RMSNorm weights for the column groups [0, qk_cols0) / [qk_cols0, qk_cols1).
    ``emit_*`` (EPI_RESIDUAL) / ``row_ss`` (EPI_STORE_BF16, EPI_GELU_BF16): the folded RMSNorm of ga_dit.h; ``emit_w`` [N] and
    ``emit_scale`` [B, N] fp32 make it the modulated one; a 2-d ``bias`` [B, N] is one bias row per batch item (``rows_per_batch``).
    ``k_rows``: EPI_RESIDUAL rows >= k_rows get the epilogue with a zero product.
    ``splitk_ws``: uint8 scratch of ``splitk_workspace(M, N)`` (zero-initialised counters) = GaGemmArgs.splitk_ws."""
    _need_cuda(A, W, bias, out, gate, vt)
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and A.stride(-1) == 1 and W.is_contiguous()
    M, K = A.shape
    N = W.shape[0] if N is None else N       # (a tiled weight comes as the flat image of tile_weight)
    if out is None:
        out = torch.empty((M, N if vt is None else vt_col0), device=A.device,
                          dtype=torch.bfloat16 if epilogue in (EPI_STORE_BF16, EPI_GELU_BF16) else torch.float32)
    M = out.shape[0] if k_rows else M       # (A may hold only the first k_rows rows)
    a = GaGemmArgs(M, N, K, epilogue, A.data_ptr(), A.stride(0), W.data_ptr(), _ptr(bias), out.data_ptr(), out.stride(0),
                   _ptr(gate), gate.stride(0) if gate is not None else 0, rows_per_batch, _ptr(vt), vt_col0,
                   vt.stride(0) if vt is not None else 0, _ptr(qk_w0), _ptr(qk_w1), qk_cols0, max(qk_cols1, qk_cols0),
                   _ptr(emit_x), _ptr(emit_ss), emit_x.stride(0) if emit_x is not None else 0,
                   _ptr(row_ss), row_ss.shape[1] if row_ss is not None else 0, row_ss_dim, row_ss_eps, 1 if w_tiled else 0,
                   _ptr(emit_w), _ptr(emit_scale), emit_scale.stride(0) if emit_scale is not None else 0,
                   bias.stride(0) if (bias is not None and bias.dim() == 2) else 0, k_rows,
                   _ptr(splitk_ws), splitk_ws.numel() if splitk_ws is not None else 0)
    check(lib().ga_gemm_bf16(ctypes.byref(a), _stream(A)), "ga_gemm_bf16")
    return out


def splitk_workspace(M, N, device):
    """zeroed scratch for the deterministic split-K of ``gemm`` (include/ga_dit.h: GaGemmArgs.splitk_ws)"""
    return torch.zeros(int(lib().ga_gemm_splitk_workspace_bytes(M, N)), dtype=torch.uint8, device=device)


def splitk_mode(mode):
    """ga_gemm_splitk_mode: -1 by shape (default), 0 off, 1 / 2 / 3 force a configuration; returns the previous mode"""
    return int(lib().ga_gemm_splitk_mode(int(mode)))


def tile_weight(W):
    """[N, K] weight (N % 8 == 0, K % 64 == 0) -> the tiled image [N/8][K/64][8][64] of GaGemmArgs.w_tiled, flat [N*K]."""
    N, K = W.shape
    assert N % 8 == 0 and K % 64 == 0
    return W.detach().reshape(N // 8, 8, K // 64, 64).permute(0, 2, 1, 3).contiguous().reshape(-1)


def transpose_v(v):
    """v [B,Lk,H,64] bf16 view -> V^T [B*H*64, Lk rounded up to 64] (zero pad), the layout ``attention`` reads."""
    B, Lk, H, d = v.shape
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros((B, H, 64, Lp), device=v.device, dtype=torch.bfloat16)
    vt[..., :Lk] = v.permute(0, 2, 3, 1)
    return vt.reshape(B * H * 64, Lp)


def attention(q, k, vt, q_norm_weight=None, k_norm_weight=None, qp=None):
    """q [B,Lq,H,64], k [B,Lk,H,64] bf16 views (token stride arbitrary, head stride 64), vt = V^T [B*H*64, Lpad]
    -> [B,Lq,H*64] bf16.  ``qp`` = dict(a=[B*Lq, K] bf16 rows, w=[H*64, K] bf16 weight (or its tile_weight image with tiled=True),
    row_ss=[B*Lq, tiles] fp32 / row_ss_dim / row_ss_eps optional, B=, Lq=, H=): the q projection inside the attention workgroups
    (GaAttentionArgs.qp_*); ``q`` is then None."""
    _need_cuda(k, vt)
    if qp is None:
        _need_cuda(q)
        B, Lq, H, d = q.shape
        assert d == 64 and q.stride(3) == 1 and q.stride(2) == 64
        qs = q.stride(1) if Lq > 1 else q.stride(0)
        assert q.stride(0) == Lq * qs
    else:
        B, Lq, H, qs = qp["B"], qp["Lq"], qp["H"], 0
        assert qp["a"].dtype == torch.bfloat16 and qp["w"].dtype == torch.bfloat16 and qp["a"].stride(-1) == 1
    Lk = k.shape[1]
    assert k.shape[3] == 64 and k.stride(2) == 64 and vt.stride(1) == 1
    # the C-ABI addresses row (b, i) at (b * L + i) * stride; a size-1 token axis has no meaningful stride of its own in torch
    ks = k.stride(1) if Lk > 1 else k.stride(0)
    assert k.stride(0) == Lk * ks and vt.shape[0] == B * H * 64
    out = torch.empty((B, Lq, H * 64), device=k.device, dtype=torch.bfloat16)
    a = GaAttentionArgs(B, H, Lq, Lk, q.data_ptr() if qp is None else None, k.data_ptr(), vt.data_ptr(), qs, ks, vt.stride(0),
                        _ptr(q_norm_weight), _ptr(k_norm_weight), out.data_ptr(), H * 64)
    if qp is not None:
        rs = qp.get("row_ss")
        a.qp_a, a.qp_w, a.qp_lda, a.qp_k = qp["a"].data_ptr(), qp["w"].data_ptr(), qp["a"].stride(0), qp["a"].shape[1]
        a.qp_w_tiled = 1 if qp.get("tiled") else 0
        if rs is not None:
            a.qp_row_ss, a.qp_row_ss_tiles, a.qp_row_ss_dim, a.qp_row_ss_eps = rs.data_ptr(), rs.shape[1], qp["row_ss_dim"], qp.get("row_ss_eps", 1e-5)
    check(lib().ga_attention_bf16(ctypes.byref(a), _stream(k)), "ga_attention_bf16")
    return out


def attention_hd(q, k, v=None, vt=None, q_norm_weight=None, k_norm_weight=None):
    """Head dims other than 64: q [B,Lq,H,d], k / v [B,Lk,H,d] bf16 views (token stride arbitrary, head stride d), k already head-normalised
    -> [B,Lq,H*d] bf16.  ``v`` row-major: the round-5 kernel (q normalised by the caller).  ``vt`` = V^T [B*H*d, Lpad] (``v_transposed_hd``):
    the tuned kernel; ``q_norm_weight`` / ``k_norm_weight`` fp32 [d] make it apply q's / k's per-head RMSNorm itself."""
    _need_cuda(q, k, v if vt is None else vt)
    B, Lq, H, d = q.shape
    Lk = k.shape[1]
    for t in (q, k) + ((v,) if vt is None else ()):
        assert t.dtype == torch.bfloat16 and t.stride(3) == 1 and t.stride(2) == d and t.stride(0) == t.shape[1] * t.stride(1)
    out = torch.empty((B, Lq, H * d), device=q.device, dtype=torch.bfloat16)
    if vt is None:
        a = GaAttentionHdArgs(B, H, Lq, Lk, d, q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(1), k.stride(1), v.stride(1), out.data_ptr(), H * d,
                              None, 0, None, None)
    else:
        assert vt.dtype == torch.bfloat16 and vt.stride(1) == 1 and vt.shape[0] == B * H * d
        a = GaAttentionHdArgs(B, H, Lq, Lk, d, q.data_ptr(), k.data_ptr(), None, q.stride(1), k.stride(1), 0, out.data_ptr(), H * d,
                              vt.data_ptr(), vt.stride(0), _ptr(q_norm_weight), _ptr(k_norm_weight))
    check(lib().ga_attention_hd_bf16(ctypes.byref(a), _stream(q)), "ga_attention_hd_bf16")
    return out


def v_transposed_hd(v):
    """v [B,Lk,H,d] bf16 -> V^T [B*H*d, Lpad] with zero pad columns (what GaGemmArgs.vt stores for a model of any width)."""
    B, Lk, H, d = v.shape
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros((B, H, d, Lp), device=v.device, dtype=torch.bfloat16)
    vt[..., :Lk] = v.permute(0, 2, 3, 1)
    return vt.reshape(B * H * d, Lp)


def head_rmsnorm_(x, heads, head_dim, weight):
    """in place on the first heads*head_dim columns of every row of x [rows, stride] bf16"""
    _need_cuda(x, weight)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    check(lib().ga_head_rmsnorm_bf16(x.data_ptr(), x.shape[0], x.stride(0), heads, head_dim, weight.data_ptr(), _stream(x)), "ga_head_rmsnorm_bf16")
    return x


def rmsnorm_modulate(x, weight, scale=None, shift=None, rows_per_batch=1):
    """x [M,D] fp32 -> bf16 [M,D]; scale/shift [B,D] (any row stride)."""
    _need_cuda(x, weight, scale, shift)
    M, D = x.shape
    out = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
    a = GaRmsNormArgs(M, D, rows_per_batch, x.data_ptr(), weight.data_ptr(), _ptr(scale), _ptr(shift),
                      scale.stride(0) if scale is not None else 0, out.data_ptr(), None, 0)
    check(lib().ga_rmsnorm_modulate(ctypes.byref(a), _stream(x)), "ga_rmsnorm_modulate")
    return out


def shift_bias(W, shift, bias=None, w_tiled=False, N=None):
    """W [N, K] bf16 (or its tiled image with ``N``), shift [B, K] fp32 rows -> bias[n] + shift_b . W[n] as fp32 [B, N]: the bias
    rows of a GEMM behind a folded modulated RMSNorm (ga_dit.h)."""
    _need_cuda(W, shift, bias)
    B, K = shift.shape
    N = W.shape[0] if N is None else N
    assert shift.dtype == torch.float32 and shift.stride(1) == 1 and W.dtype == torch.bfloat16
    out = torch.empty((B, N), device=W.device, dtype=torch.float32)
    check(lib().ga_dit_shift_bias(W.data_ptr(), 1 if w_tiled else 0, _ptr(bias), N, K, shift.data_ptr(), shift.stride(0), B,
                                  out.data_ptr(), _stream(W)), "ga_dit_shift_bias")
    return out


def small_linear(x, W, bias=None, add=None, act_in=0, act_out=0):
    _need_cuda(x, W, bias, add)
    B, K = x.shape
    N = W.shape[0]
    y = torch.empty((B, N), device=x.device, dtype=torch.float32)
    a = GaSmallLinearArgs(B, N, K, act_in, act_out, x.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(add), y.data_ptr())
    check(lib().ga_small_linear(ctypes.byref(a), _stream(x)), "ga_small_linear")
    return y
