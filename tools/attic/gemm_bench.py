#!/usr/bin/env python3
"""Times ga_gemm_bf16 / ga_attention_bf16 at the DiT shapes (torch events on the current stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

M = 1536
for (N, K, epi) in [(1024, 1024, 2), (3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 2), (1024, 1024, 0), (2048, 1024, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16() / 32
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev) if epi in (2, 3) else None
    us = timeit(lambda: ops.gemm(A, W, bias, epi, out=out))
    print(f"gemm M={M} N={N} K={K} epi={epi}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
for (B, H, Lq, Lk) in [(2, 16, 768, 768), (2, 16, 768, 1369), (2, 12, 768, 768)]:
    D = H * 64
    q = torch.randn(B, Lq, 3 * D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    w = torch.ones(64, device=dev)
    qq = q[..., :D].unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
    vt = ops.transpose_v(v)
    us = timeit(lambda: ops.attention(qq, k, vt, w, w))
    print(f"attn B={B} H={H} Lq={Lq} Lk={Lk}: {us:7.1f} us  {4*B*H*Lq*Lk*64/us/1e6:7.1f} TF/s")
# ablations
B, H, Lq, Lk = 2, 16, 768, 768
D = H * 64
q = torch.randn(B, Lq, 3 * D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
qq = q[..., :D].unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
vt = ops.transpose_v(v)
print("attn no norms:", timeit(lambda: ops.attention(qq, k, vt, None, None)))
for LK in (64, 128, 256, 512, 768):
    kk = k[:, :LK].contiguous(); vtt = ops.transpose_v(v[:, :LK])
    print("attn Lk", LK, timeit(lambda: ops.attention(qq, kk, vtt, w, w)))
