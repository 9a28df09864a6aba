#!/usr/bin/env python3
"""Times the ablated builds of tools/gemm_ablate.sh at the DiT-L GEMM shapes (M = 1536)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
BITS = {1: "MFMAs", 2: "DMA", 4: "barrier", 8: "epilogue", 16: "fragment reads"}
MASKS = [0, 1, 2, 4, 8, 16, 3, 7, 19, 23, 31, 32]
M = 1536
for (N, K, epi) in [(1024, 1024, 2), (1024, 4096, 2), (4096, 1024, 1), (3072, 1024, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16() / 32
    bias = torch.randn(N, device=dev); gate = torch.randn(2, N, device=dev)
    out = torch.zeros(M, N, device=dev) if epi in (2, 3) else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    a = ops.GaGemmArgs(M, N, K, epi, A.data_ptr(), A.stride(0), W.data_ptr(), bias.data_ptr(), out.data_ptr(), out.stride(0),
                       gate.data_ptr() if epi == 2 else None, gate.stride(0) if epi == 2 else 0, 768, None, 0, 0, None, None, 0, 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in MASKS:
        name = "full" if i == 0 else ("no K loop" if i == 32 else "no " + " / ".join(v for k, v in BITS.items() if i & k))
        so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", f"gemm_ablate_{i}.so"))
        for _ in range(10): assert so.ga_gemm_bf16(ctypes.byref(a), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): so.ga_gemm_bf16(ctypes.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} K={K} epi={epi}  {name:44s} {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us", flush=True)
