import os, sys
sys.path.insert(0, "/root/repo")
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, H, Lq, Lk) in [(2, 16, 768, 768), (1, 16, 768, 1369), (1, 16, 768, 768), (2, 12, 768, 768), (8, 16, 768, 768)]:
    D = H * 64
    q = torch.randn(B, Lq, 3 * D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    w = torch.ones(64, device=dev)
    qq = q[..., :D].unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
    vt = ops.transpose_v(v)
    us = timeit(lambda: ops.attention(qq, k, vt, None, None))
    print(f"attn B={B} H={H} Lq={Lq} Lk={Lk}: {us:7.2f} us  {4*B*H*Lq*Lk*64/us/1e6:7.1f} TF/s")
