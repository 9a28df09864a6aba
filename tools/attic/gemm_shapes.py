import os, sys
sys.path.insert(0, "/root/repo")
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (1536, 768):
    for (N, K, epi) in [(3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 2), (1024, 1024, 2), (2048, 1024, 0)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16() / 32
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev) if epi in (2, 3) else None
        us = timeit(lambda: ops.gemm(A, W, bias, epi, out=out))
        print(f"gemm M={M} N={N} K={K} epi={epi}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
