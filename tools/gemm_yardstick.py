#!/usr/bin/env python3
"""Same-node yardstick for the DiT GEMM shapes: this library's ga_gemm_bf16 next to torch.matmul in bf16 (hipBLASLt / rocBLAS
behind PyTorch) on exactly the shapes of a DiT-L evaluation, weights rotated through more copies than the Infinity Cache holds,
every sequence of launches captured in a HIP graph and replayed (so that the small shapes are not host-bound).
Tools / bench only -- the product path never calls a vendor GEMM.  usage (GPU box): python tools/gemm_yardstick.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = [("qkv", 1536, 3072, 1024, 0), ("fc1", 1536, 4096, 1024, 1), ("fc2", 1536, 1024, 4096, 2), ("proj", 1536, 1024, 1024, 2),
          ("ca_q", 768, 1024, 1024, 0), ("fc1_x4", 6144, 4096, 1024, 1)]


def graph_us(fn, copies, reps=5):
    """mean microseconds per call of fn(i), i = 0 .. copies-1 captured once and replayed `reps` times"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(min(3, copies)):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(copies):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * copies) * 1e3


def yardstick(dev=None, copies=40):
    from gaussiananything_amd import dit_ops as ops
    dev = dev or torch.device("cuda:0")
    res = {}
    for name, M, N, K, epi in SHAPES:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(copies, N, K, device=dev) * 0.03).bfloat16()
        bias = torch.randn(N, device=dev)
        out_t = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out_o = torch.zeros(M, N, device=dev) if epi == 2 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_us = graph_us(lambda i: torch.matmul(A, W[i].t(), out=out_t), copies)
        o_us = graph_us(lambda i: ops.gemm(A, W[i], bias, epi, out=out_o), copies)
        fl = 2.0 * M * N * K
        res[name] = {"M": M, "N": N, "K": K, "ga_gemm_epilogue": ["bias", "bias+gelu", "bias+residual(fp32)"][epi],
                     "ga_gemm_us": round(o_us, 2), "ga_gemm_tflops": round(fl / o_us / 1e6, 1),
                     "torch_matmul_us": round(t_us, 2), "torch_matmul_tflops": round(fl / t_us / 1e6, 1)}
    res["note"] = ("cold weights (40 copies), HIP-graph replay; torch.matmul is the plain product (no bias / activation / residual), "
                   "ga_gemm_bf16 includes its fused epilogue")
    return res


if __name__ == "__main__":
    json.dump(yardstick(), sys.stdout, indent=1)
    print()
