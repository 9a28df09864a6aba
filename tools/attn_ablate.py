#!/usr/bin/env python3
"""Times the ablated builds of tools/attn_ablate.sh at the DiT self-attention shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
dev = torch.device("cuda:0")
NAMES = ["full", "no PV MFMAs", "no S MFMAs", "no exponentials", "no V^T fragment reads", "no DMA in the loop", "no barrier"]
for (B, H, Lq, Lk) in [(2, 16, 768, 768), (1, 16, 768, 1369)]:
    D = H * 64
    q = torch.randn(B, Lq, D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    qq = q.unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
    vt = ops.transpose_v(v)
    out = torch.empty(B, Lq, H * 64, device=dev, dtype=torch.bfloat16)
    a = ops.GaAttentionArgs(B, H, Lq, Lk, qq.data_ptr(), k.data_ptr(), vt.data_ptr(), qq.stride(1), k.stride(1), vt.stride(0),
                            None, None, out.data_ptr(), H * 64)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, name in enumerate(NAMES):
        so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", f"attn_ablate_{i}.so"))
        for _ in range(10): so.ga_attention_bf16(ctypes.byref(a), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): so.ga_attention_bf16(ctypes.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} Lk={Lk}  {name:24s} {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us", flush=True)
