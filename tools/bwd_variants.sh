#!/bin/bash
# Build variants of the backward kernels (macros in surfel_backward.hip) into tools/_build/libga_<name>.so.
# usage: tools/bwd_variants.sh name1 "-DFLAGS1" name2 "-DFLAGS2" ...
set -e
cd "$(dirname "$0")/.."
SRC=gaussiananything_amd/csrc; OBJ=gaussiananything_amd/lib/obj; OUT=tools/_build
mkdir -p $OUT
make -s -C $SRC
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $flags -c $SRC/surfel_backward.hip -o $OUT/bwd_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libga_$name.so $OBJ/surfel_preprocess.o $OBJ/surfel_bin.o $OBJ/surfel_blend.o $OBJ/surfel_api.o $OBJ/surfel_post.o $OUT/bwd_$name.o $OBJ/tsdf.o $OBJ/dit_gemm.o $OBJ/dit_attention.o $OBJ/dit_ops.o $OBJ/decode_ops.o
  echo "built $name ($flags)"
done
