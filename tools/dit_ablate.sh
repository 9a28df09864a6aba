#!/bin/bash
# Wall-time shares of the launch classes of one DiT evaluation: a library build whose forward can drop them (GA_DIT_SKIP).
cd "$(dirname "$0")/.." && mkdir -p tools/_build
C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGA_DIT_ABLATE -Iinclude"
S=gaussiananything_amd/csrc
/opt/rocm/bin/hipcc $C -shared -o tools/_build/libga_dit_ablate.so $S/surfel_preprocess.hip $S/surfel_bin.hip $S/surfel_blend.hip \
    $S/surfel_api.hip $S/surfel_post.hip $S/surfel_backward.hip $S/tsdf.hip $S/dit_gemm.hip $S/dit_attention.hip $S/dit_ops.hip $S/decode_ops.hip
