#!/usr/bin/env python3
"""Which tile does ga_gemm_bf16 want at a given (M, N, K)?  A -DGA_TUNING build of the library honours GA_GEMM_RING (0 = the
round-1/2 kernel with its own tile choice, 1 = 192x128 / 8 waves, 2 = 96x64, 3 = 64x64) and this script times every choice per
shape with cold weights under HIP-graph replay.  usage (GPU box): GA_LIB=tools/_build/libga_tuning.so python tools/gemm_ring_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
from tools.gemm_yardstick import graph_us
dev = torch.device("cuda:0")
Ms = [int(v) for v in (sys.argv[1:] or ["768", "1536", "3072", "6144", "1369", "1374"])]
for M in Ms:
    for (name, N, K, epi) in [("qkv", 3072, 1024, 0), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 2), ("proj", 1024, 1024, 2),
                              ("qkvB", 2304, 768, 0), ("fc1B", 3072, 768, 1), ("fc2B", 768, 3072, 2), ("projB", 768, 768, 2)]:
        copies = 40
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(copies, N, K, device=dev) * 0.03).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.zeros(M, N, device=dev) if epi == 2 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {}
        for ring in ("auto", 0, 1, 2, 3):
            if ring == "auto": os.environ.pop("GA_GEMM_RING", None)
            else: os.environ["GA_GEMM_RING"] = str(ring)
            res[ring] = graph_us(lambda i: ops.gemm(A, W[i], bias, epi, out=out), copies, reps=3)
        os.environ.pop("GA_GEMM_RING", None)
        best = min((k for k in res if k != "auto"), key=lambda k: res[k])
        print(f"M={M:5d} {name:5s} N={N} K={K}: auto {res['auto']:6.2f} | best ring {best} {res[best]:6.2f} | " +
              " ".join(f"{k}:{v:.2f}" for k, v in res.items() if k != "auto"), flush=True)
