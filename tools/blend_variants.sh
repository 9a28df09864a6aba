#!/bin/bash
# Build variants of the blend kernel (macros in surfel_blend.hip) into tools/_build/libga_<name>.so.
# usage: tools/blend_variants.sh name1 "-DFLAGS1" name2 "-DFLAGS2" ...
set -e
cd "$(dirname "$0")/.."
SRC=gaussiananything_amd/csrc; OBJ=gaussiananything_amd/lib/obj; OUT=tools/_build
mkdir -p $OUT
make -s -C $SRC
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $flags -c $SRC/surfel_blend.hip -o $OUT/blend_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $flags -c $SRC/surfel_bin.hip -o $OUT/bin_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $flags -c $SRC/surfel_preprocess.hip -o $OUT/pre_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $flags -c $SRC/surfel_api.hip -o $OUT/api_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libga_$name.so $OUT/pre_$name.o $OUT/bin_$name.o $OUT/blend_$name.o $OUT/api_$name.o $OBJ/surfel_post.o $OBJ/surfel_backward.o $OBJ/tsdf.o $OBJ/dit_gemm.o $OBJ/dit_attention.o $OBJ/dit_ops.o $OBJ/ode_dopri5.o $OBJ/decode_ops.o
  echo "built $name ($flags): $(/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT/blend_$name.o 2>/dev/null | grep -m1 -E 'vgpr_count' || true)"
done
