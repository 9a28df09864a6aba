#!/usr/bin/env python3
"""How much of a DiT GEMM's time is the weights being cold?  ga_gemm_bf16 on the DiT-L shapes with the weights rotated through 40 copies
(cold: more than the Infinity Cache holds, as in an evaluation), 4 copies (Infinity-Cache-warm: what a prefetch by idle CUs of the
previous launch could give at best) and 1 copy (L2-warm).  HIP-graph replay.  usage (GPU box): python tools/warm_vs_cold.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.gemm_yardstick import graph_us  # noqa: E402
from gaussiananything_amd import dit_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
for name, M, N, K, epi in (("qkv", 1536, 3072, 1024, 0), ("fc1", 1536, 4096, 1024, 1), ("fc2", 1536, 1024, 4096, 2), ("proj", 1536, 1024, 1024, 2),
                           ("fc2 b1", 768, 1024, 4096, 2), ("fc1 b1", 768, 4096, 1024, 1)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(40, N, K, device=dev) * 0.03).bfloat16()
    Wt = torch.stack([ops.tile_weight(W[i]) for i in range(40)])
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev) if epi == 2 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for copies in (40, 4, 1):
        best = min(graph_us(lambda i: ops.gemm(A, Wt[i % copies], bias, epi, out=out, w_tiled=True, N=N), 40) for _ in range(3))
        res.append(best)
    print(f"{name:8s} {M}x{N}x{K}: cold {res[0]:6.2f} us | 4 copies {res[1]:6.2f} us | 1 copy {res[2]:6.2f} us", flush=True)
