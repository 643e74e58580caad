#!/bin/bash
# Build a tuning variant of the library for A/B timing on the GPU box:
#   tools/build_variant.sh <tag> <file.hip> [-DNAME=VALUE ...]
# recompiles ONE translation unit with the given defines and links it with the product's other
# objects into <pkg>/lib/variants/libsdpa_hip_<tag>.so (select it with $SDPA_HIP_LIB in tools/).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
TAG=$1; TU=$2; shift 2
make -s -C $PKG/csrc
mkdir -p $PKG/lib/variants $PKG/build/variants
OBJ=$PKG/build/variants/${TU%.hip}_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-inline-asm "$@" -c $PKG/csrc/$TU -o $OBJ
OTHERS=$(ls $PKG/build/*.o | grep -v "/${TU%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $PKG/lib/variants/libsdpa_hip_$TAG.so $OBJ $OTHERS -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo "built lib/variants/libsdpa_hip_$TAG.so"
