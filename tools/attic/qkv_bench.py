#!/usr/bin/env python3
"""The qkv projection as the DiT forward issues it (per-head q/k RMSNorm + transposed V store in the epilogue): time and V^T check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
M, D = 1536, 1024
A = torch.randn(M, D, device=dev).bfloat16(); W = (torch.randn(3 * D, D, device=dev) / 32).bfloat16(); b = torch.randn(3 * D, device=dev)
qn = torch.ones(64, device=dev); vt = torch.zeros(2 * 16 * 64, 768, device=dev, dtype=torch.bfloat16)
out = torch.empty(M, 2 * D, device=dev, dtype=torch.bfloat16)
f = lambda: ops.gemm(A, W, b, ops.EPI_STORE_BF16, out=out, rows_per_batch=768, vt=vt, vt_col0=2 * D, qk_w0=qn, qk_cols0=D, qk_w1=qn, qk_cols1=2 * D)
for _ in range(10): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): f()
e1.record(); torch.cuda.synchronize()
ref = (A.float() @ W.float().T + b)[:, 2 * D:]
got = vt.reshape(2, 16, 64, 768).permute(0, 3, 1, 2).reshape(M, D).float()
print(f"qkv GEMM with V^T store: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us; V^T rel err {((got - ref).norm() / ref.norm()).item():.2e}")
