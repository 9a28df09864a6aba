"""Times ga_attention_hd_bf16 at DiT-PixArt-PCD-CLAY-XL's two shapes (16 heads of 72): the round-5 kernel (v row-major, q normalised by a
launch of its own) against the tuned V^T variant (q's norm inside).  python tools/attn_hd_bench.py [reps]"""
import sys

import torch

from gaussiananything_amd import dit_ops as ops


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    H, d = 16, 72
    for name, B, Lq, Lk in (("self-attention, CFG pair", 2, 768, 768), ("cross-attention, one sample", 1, 768, 1369), ("self-attention, one sample", 1, 768, 768)):
        g = torch.Generator().manual_seed(1)
        q = torch.randn(B, Lq, H, d, generator=g).to(dev).bfloat16()
        k = torch.randn(B, Lk, H, d, generator=g).to(dev).bfloat16()
        v = torch.randn(B, Lk, H, d, generator=g).to(dev).bfloat16()
        wq = torch.ones(d, device=dev)
        vt = ops.v_transposed_hd(v)
        qn = q.clone().view(B * Lq, H * d)
        t_old = timed(lambda: ops.attention_hd(q, k, v), reps)
        t_norm = timed(lambda: ops.head_rmsnorm_(qn, H, d, wq), reps)
        t_new = timed(lambda: ops.attention_hd(q, k, vt=vt, q_norm_weight=wq), reps)
        flops = 4.0 * B * H * Lq * Lk * d
        print(f"{name}: round-5 kernel {t_old:.1f} us (+ q norm launch {t_norm:.1f}), V^T variant with the norm inside {t_new:.1f} us "
              f"({flops / t_new * 1e-6:.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
