"""One line: ms per evaluation of DiT-PixArt-PCD-CLAY-XL (bench.bench_dit, replayed Euler graph).  python tools/xl_eval.py [nfe]"""
import sys

import torch

import bench

if __name__ == "__main__":
    nfe = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    r = bench.bench_dit(torch.device("cuda:0"), "DiT-PixArt-PCD-CLAY-XL", nfe, 2)
    print({k: r[k] for k in ("arch", "ms_per_nfe", "achieved_tflops", "frac_of_mfma_peak")}, flush=True)
