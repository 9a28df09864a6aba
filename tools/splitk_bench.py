#!/usr/bin/env python3
"""fc2 of a DiT block (the residual GEMM with the long reduction: 1536 / 768 / 3072 rows x 1024 x 4096, and DiT-B's 1536 x 768 x 3072)
through ga_gemm_bf16 with each split-K configuration of GaGemmArgs.splitk_ws (0 = none: the 96 x 64 / 64 x 64 ring tiles; 1 = 192 x 128
x 4 splits; 2 = 96 x 128 x 2; 3 = 96 x 128 x 4), torch.matmul beside them.  Cold weights (40 copies), HIP-graph replay, same process,
modes interleaved and repeated.  usage (GPU box): python tools/splitk_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.gemm_yardstick import graph_us  # noqa: E402


def main():
    from gaussiananything_amd import dit_ops as ops
    dev = torch.device("cuda:0")
    copies = 40
    for name, M, N, K in (("fc2 L cfg2", 1536, 1024, 4096), ("fc2 L b1", 768, 1024, 4096), ("fc2 L cfg4", 3072, 1024, 4096), ("fc2 B cfg2", 1536, 768, 3072),
                          ("proj L cfg2", 1536, 1024, 1024)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(copies, N, K, device=dev) * 0.03).bfloat16()
        Wt = torch.stack([ops.tile_weight(W[i]) for i in range(copies)])
        bias = torch.randn(N, device=dev)
        gate = torch.randn(M // 768, N, device=dev)
        out = torch.zeros(M, N, device=dev)
        ex = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ss = torch.empty(M, N // 64, device=dev)
        ws = ops.splitk_workspace(M, N, dev)
        out_t = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {m: [] for m in (0, 1, 2, 3)}
        tm = []
        for rep in range(3):
            tm.append(graph_us(lambda i: torch.matmul(A, W[i].t(), out=out_t), copies))
            for mode in (0, 1, 2, 3):
                prev = ops.splitk_mode(mode)
                try:   # the call as ga_dit_forward makes it: tiled weights, gate, emit for the next block's folded pre-norm
                    res[mode].append(graph_us(lambda i: ops.gemm(A, Wt[i], bias, ops.EPI_RESIDUAL, out=out, gate=gate, rows_per_batch=768, emit_x=ex,
                                                                 emit_ss=ss, w_tiled=True, N=N, splitk_ws=ws if mode else None), copies))
                finally:
                    ops.splitk_mode(prev)
        fl = 2.0 * M * N * K
        print(f"{name:12s} {M}x{N}x{K}: torch.matmul {min(tm):6.2f} us | " + " | ".join(
            f"mode {m}: {min(v):6.2f} us ({fl / min(v) / 1e6:5.0f} TF/s) runs {' '.join(f'{x:.1f}' for x in v)}" for m, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
