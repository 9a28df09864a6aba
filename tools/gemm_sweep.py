#!/usr/bin/env python3
"""Sweep the GEMM tile configurations (GA_GEMM_CFG = 10 MT + ring slots; a -DGA_TUNING build of the library) over the DiT
shapes.  usage (GPU box): python tools/gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")

def timeit(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M in (1536, 768, 1369, 2738):
    for (N, K, epi) in [(3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 2), (1024, 1024, 2), (2048, 1024, 0), (1024, 1024, 0)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16() / 32
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev) if epi in (2, 3) else None
        res = {}
        os.environ.pop("GA_GEMM_CFG", None)
        res["auto"] = timeit(lambda: ops.gemm(A, W, bias, epi, out=out))
        for mt in (4, 3, 2, 1):
            for nst in (2, 4):
                os.environ["GA_GEMM_CFG"] = str(10 * mt + nst)
                res[f"{mt}/{nst}"] = timeit(lambda: ops.gemm(A, W, bias, epi, out=out))
        os.environ.pop("GA_GEMM_CFG", None)
        best = min((k for k in res if k != "auto"), key=lambda k: res[k])
        print(f"M={M} N={N} K={K} epi={epi}: auto {res['auto']:.1f}  best {best} {res[best]:.1f} | " + " ".join(f"{k}:{v:.1f}" for k, v in res.items() if k != "auto"), flush=True)
