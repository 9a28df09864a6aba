#!/usr/bin/env python3
"""ga_gemm_bf16 next to torch.matmul at M = 768 (the stage-2 evaluation of the cascade: batch 1), cold weights, graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_yardstick import graph_us
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 768
for name, N, K, epi, qk in [("qkv", 3072, 1024, 0, True), ("qkv_plain", 3072, 1024, 0, False), ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 2, False),
                            ("proj", 1024, 1024, 2, False), ("ca_q", 1024, 1024, 0, True)]:
    copies = 40
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(copies, N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    out_t = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(M, N, device=dev) if epi == 2 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    qw = torch.ones(64, device=dev)
    kw = dict(qk_w0=qw, qk_w1=qw, qk_cols0=1024, qk_cols1=min(N, 2048)) if qk else {}
    if qk and N == 1024: kw = dict(qk_w0=qw, qk_cols0=1024)
    t_us = graph_us(lambda i: torch.matmul(A, W[i].t(), out=out_t), copies)
    o_us = graph_us(lambda i: ops.gemm(A, W[i], bias if not qk else None, epi, out=out, **kw), copies)
    fl = 2.0 * M * N * K
    print(f"M={M} {name:9s} N={N} K={K}: ours {o_us:6.2f} us ({fl / o_us / 1e6:5.0f} TF)   torch.matmul {t_us:6.2f} us ({fl / t_us / 1e6:5.0f} TF)", flush=True)
