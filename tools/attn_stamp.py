#!/usr/bin/env python3
"""Cycle stamps inside ga_attention_bf16's tile loop.  Build the instrumented copy first (hipcc, gfx950):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGA_ATTN_STAMP=5 -Iinclude -o tools/_build/attn_stamp.so \
         gaussiananything_amd/csrc/dit_attention.hip
Stamps (s_memtime, 100 MHz-independent shader clock counter) of step GA_ATTN_STAMP, per wave of workgroups x = 0..7:
0 step start | 1 staged (LDS writes + global loads issued) | 2 K fragment reads issued | 3 S MFMAs issued |
4 row max known | 5 exponentials done | 6 PV MFMAs issued | 7 past the barrier"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiananything_amd import dit_ops as ops
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "attn_stamp.so"))
dev = torch.device("cuda:0")
B, H, Lq, Lk = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 16, 768, 768))]
D = H * 64
q = torch.randn(B, Lq, D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
qq = q.unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
vt = ops.transpose_v(v)
out = torch.empty(B, Lq, H * 64, device=dev, dtype=torch.bfloat16)
a = ops.GaAttentionArgs(B, H, Lq, Lk, qq.data_ptr(), k.data_ptr(), vt.data_ptr(), qq.stride(1), k.stride(1), vt.stride(0),
                        None, None, out.data_ptr(), H * 64)
for _ in range(3):
    rc = so.ga_attention_bf16(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); assert rc == 0
torch.cuda.synchronize()
ref = ops.attention(qq, k, vt, None, None)
print("max diff vs product library:", (ref.float() - out.float()).abs().max().item())
buf = (ctypes.c_ulonglong * (8 * 16 * 16))()
assert so.ga_attn_debug_stamps(buf) == 0
st = np.array(buf, dtype=np.int64).reshape(8, 16, 16)
for wg in (0, 3):
    for w in range(16):
        if st[wg, w, 0] == 0: continue
        d = st[wg, w, :8] - st[wg, 0, 0]
        print(f"wg {wg} wave {w:2d}: " + " ".join(f"{x:6d}" for x in d) + "   deltas " + " ".join(f"{x:5d}" for x in np.diff(st[wg, w, :8]))
              + f"   vmcnt wait ends at {st[wg, w, 8] - st[wg, 0, 0]}")
