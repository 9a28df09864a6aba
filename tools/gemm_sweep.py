#!/usr/bin/env python3
"""K / N / M sweeps of ga_gemm_bf16: separates the fixed cost of a launch from the per-K-tile cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return a.elapsed_time(b) / n * 1e3  # us

COLD = bool(os.environ.get("GA_COLD"))   # rotate through > 512 MB of distinct weights, as one DiT evaluation does (0.8 GB)

def run(M, N, K, epi):
    A = torch.randn(M, K, device=dev).bfloat16()
    nW = max(1, (640 << 20) // (N * K * 2)) if COLD else 1
    Ws = [torch.randn(N, K, device=dev).bfloat16() / 32 for _ in range(nW)]
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev) if epi in (2, 3) else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    it = [0]
    def fn():
        it[0] = (it[0] + 1) % nW
        ops.gemm(A, Ws[it[0]], bias, epi, out=out)
    us = timeit(fn, 200 if COLD else 100)
    print(f"gemm M={M:5d} N={N:5d} K={K:5d} epi={epi}{' cold' if COLD else ''}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s", flush=True)

if os.environ.get("GA_ONE_SHAPE"):
    n_, k_, e_ = [int(x) for x in os.environ["GA_ONE_SHAPE"].split(",")]
    run(1536, n_, k_, e_)
    sys.exit(0)
for (N, K, epi) in [(3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 2), (1024, 1024, 2), (2048, 1024, 0)]:
    run(1536, N, K, epi)
for (N, K, epi) in [(1024, 1024, 2), (1024, 1024, 0)]:   # one CFG half alone (cross-attention q / out projections)
    run(768, N, K, epi)
if os.environ.get("GA_FULL_SWEEP"):
    for K in (64, 256, 1024, 2048, 4096):
        run(1536, 1024, K, 2)
    run(8192, 8192, 8192, 0)
