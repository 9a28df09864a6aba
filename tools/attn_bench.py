#!/usr/bin/env python3
"""ga_attention_bf16 at the DiT shapes (self-attention 768 keys, image cross-attention 1369 keys), as the forward calls
it: q/k already RMS-normalised by the projection GEMM (no norm weights here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

for (B, H, Lq, Lk) in [(2, 16, 768, 768), (2, 16, 768, 1369), (1, 16, 768, 1369), (2, 12, 768, 768), (1, 12, 768, 1369)]:
    D = H * 64
    q = torch.randn(B, Lq, D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    qq = q.unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
    vt = ops.transpose_v(v)
    us = timeit(lambda: ops.attention(qq, k, vt, None, None))
    print(f"attn B={B} H={H} Lq={Lq} Lk={Lk}: {us:7.1f} us  {4*B*H*Lq*Lk*64/us/1e6:7.1f} TF/s", flush=True)
