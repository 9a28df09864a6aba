#!/bin/bash
# Timing-only ablations of ga_gemm_bf16 (results are wrong by construction): what each phase costs IN SITU.
# bit mask of removed phases: 1 MFMAs | 2 DMA in the K loop (stale tiles) | 4 barrier | 8 epilogue | 16 fragment reads; 32 = no K loop
cd "$(dirname "$0")/.." && mkdir -p tools/_build
for a in 0 1 2 4 8 16 3 7 19 23 31 32; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DGA_GEMM_ABLATE=$a -Iinclude \
      -o tools/_build/gemm_ablate_$a.so gaussiananything_amd/csrc/dit_gemm.hip &
done
wait
