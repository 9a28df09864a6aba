#!/usr/bin/env python3
"""Runs a few forward_with_cfg evaluations of one DiT arch (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "DiT-PixArt-PCD-CLAY-L"
print(bench.bench_dit(dev, arch, int(sys.argv[2]) if len(sys.argv) > 2 else 10, 2))
