#!/usr/bin/env python3
"""Exactly the kernels the bench line quotes, each alone, for counter passes (tools/collect_r5.sh):
  attn : the two attention launches of a DiT-L evaluation (self 2x16x768x768, cross 1x16x768x1369), 20 launches each
  gemm : the GEMM launches of a DiT-L block at M = 1536 (qkv, fc1, fc2, proj) and M = 768 (cross-attention q), 20 each,
         weights rotated through 40 copies (cold, as in an evaluation)
usage (GPU box): python tools/dit_kernels_two.py attn|gemm"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
if what == "attn":
    for (B, H, Lq, Lk) in [(2, 16, 768, 768), (1, 16, 768, 1369)]:
        D = H * 64
        q = torch.randn(B, Lq, D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
        qq = q.unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
        vt = ops.transpose_v(v)
        for _ in range(20): ops.attention(qq, k, vt, None, None)
        torch.cuda.synchronize()
    # round 5: the cross-attention launch of the evaluation proper -- the q projection (K = 1024, tiled weight, folded row scale, per-head
    # norm) inside the workgroups (GaAttentionArgs.qp_*): attention_fwd_kernel<4, 3, false> again, 20 launches behind the 20 above
    B, H, Lq, Lk, K = 1, 16, 768, 1369, 1024
    D = H * 64
    A = torch.randn(B * Lq, K, device=dev).bfloat16(); W = (torch.randn(D, K, device=dev) / 32).bfloat16()
    kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    k = kv[..., :D].unflatten(-1, (H, 64)); vt = ops.transpose_v(kv[..., D:].unflatten(-1, (H, 64)))
    rss = torch.rand(B * Lq, 16, device=dev) * 64; wq = torch.ones(64, device=dev)
    qp = dict(a=A, w=ops.tile_weight(W), tiled=True, row_ss=rss, row_ss_dim=K, B=B, Lq=Lq, H=H)
    for _ in range(20): ops.attention(None, k, vt, q_norm_weight=wq, qp=qp)
    torch.cuda.synchronize()
else:
    for (M, N, K, epi) in [(1536, 3072, 1024, 0), (1536, 4096, 1024, 1), (1536, 1024, 4096, 2), (1536, 1024, 1024, 2), (768, 1024, 1024, 0)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(40, N, K, device=dev) * 0.03).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev) if epi in (2, 3) else None
        for i in range(20): ops.gemm(A, W[i % 40], bias, epi, out=out)
        torch.cuda.synchronize()
