#!/usr/bin/env python3
"""Same-node yardstick for the DiT GEMM shapes: torch.matmul in bf16 (hipBLASLt / rocBLAS behind PyTorch) on exactly the
shapes ga_gemm_bf16 runs in a DiT-L evaluation, weights rotated through more copies than the Infinity Cache holds.
Tools only -- the product path never calls a vendor GEMM.  Prints one JSON object.  usage (GPU box): python tools/gemm_yardstick.py"""
import json
import sys

import torch


def time_us(fn, n=100, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def yardstick(shapes=None, copies=40):
    dev = torch.device("cuda:0")
    shapes = shapes or [("fc1", 1536, 4096, 1024), ("qkv", 1536, 3072, 1024), ("fc2", 1536, 1024, 4096),
                        ("proj", 1536, 1024, 1024), ("caq", 768, 1024, 1024), ("fc1x4", 6144, 4096, 1024)]
    res = {}
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(copies, N, K, device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cold = time_us(lambda i: torch.matmul(A, W[i % copies].t(), out=out))
        warm = time_us(lambda i: torch.matmul(A, W[0].t(), out=out))
        fl = 2.0 * M * N * K
        res[name] = {"M": M, "N": N, "K": K, "cold_us": round(cold, 2), "cold_tflops": round(fl / cold / 1e6, 1),
                     "warm_us": round(warm, 2), "warm_tflops": round(fl / warm / 1e6, 1)}
    return res


if __name__ == "__main__":
    json.dump(yardstick(), sys.stdout, indent=1)
    print()
