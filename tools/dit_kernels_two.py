#!/usr/bin/env python3
"""Exactly the kernels the bench line quotes, each alone, for counter passes (tools/pmc_r3.sh):
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
else:
    for (M, N, K, epi) in [(1536, 3072, 1024, 0), (1536, 4096, 1024, 1), (1536, 1024, 4096, 2), (1536, 1024, 1024, 2), (768, 1024, 1024, 0)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(40, N, K, device=dev) * 0.03).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev) if epi in (2, 3) else None
        for i in range(20): ops.gemm(A, W[i % 40], bias, epi, out=out)
        torch.cuda.synchronize()
