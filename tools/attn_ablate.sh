#!/bin/bash
# Timing-only ablations of ga_attention_bf16 (results are wrong by construction): which phase of the tile loop costs what
# IN SITU.  Build here (hipcc cross-compiles), run tools/attn_ablate.py on the GPU box.
# 0 full | 1 no PV MFMAs | 2 no S MFMAs | 3 no exponentials | 4 no V^T fragment reads | 5 no DMA in the loop | 6 no barrier
cd "$(dirname "$0")/.." && mkdir -p tools/_build
for a in 0 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DGA_ATTN_ABLATE=$a -Iinclude \
      -o tools/_build/attn_ablate_$a.so gaussiananything_amd/csrc/dit_attention.hip &
done
wait
