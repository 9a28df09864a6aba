#!/usr/bin/env python3
"""(query waves, key groups) of ga_attention_bf16 per shape: a -DGA_TUNING build honours GA_ATTN_CFG = 10 NW + KS.
usage (GPU box, tuning library in place): python tools/attn_cfg_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiananything_amd import dit_ops as ops
from tools.gemm_yardstick import graph_us
dev = torch.device("cuda:0")
for (B, H, Lq, Lk) in [(2, 16, 768, 768), (1, 16, 768, 1369), (1, 16, 768, 768), (2, 12, 768, 768), (1, 12, 768, 1369), (8, 16, 768, 768), (4, 16, 768, 1369)]:
    D = H * 64
    q = torch.randn(B, Lq, D, device=dev).bfloat16(); kv = torch.randn(B, Lk, 2 * D, device=dev).bfloat16()
    qq = q.unflatten(-1, (H, 64)); k = kv[..., :D].unflatten(-1, (H, 64)); v = kv[..., D:].unflatten(-1, (H, 64))
    vt = ops.transpose_v(v)
    res = {}
    for cfg in (0, 81, 42, 23, 43, 22, 41, 82, 21):
        if cfg: os.environ["GA_ATTN_CFG"] = str(cfg)
        else: os.environ.pop("GA_ATTN_CFG", None)
        try:
            res[cfg] = graph_us(lambda i: ops.attention(qq, k, vt, None, None), 20, reps=3)
        except Exception as e:
            res[cfg] = float("nan")
    os.environ.pop("GA_ATTN_CFG", None)
    print(f"B={B} H={H} Lq={Lq} Lk={Lk}: " + " ".join(f"{c}:{u:.2f}" for c, u in res.items()), flush=True)
