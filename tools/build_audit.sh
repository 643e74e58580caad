#!/bin/bash
# The DMA / fragment-load bounds AUDIT build (never shipped): all three kernel translation units with
# -DSDPA_DMA_ASSERT, linked with the product's other objects into <pkg>/lib/variants/libsdpa_hip_audit.so.
#   SDPA_HIP_LIB=<that file> python -m pytest tests -m gpu     -> tests/conftest.py reports the audit counters
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
make -s -C $PKG/csrc
mkdir -p $PKG/lib/variants $PKG/build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-inline-asm -DSDPA_DMA_ASSERT"
OBJS=""
for tu in sdpa_fwd_f32 sdpa_fwd_f32_dksplit sdpa_fwd_bf16; do
  /opt/rocm/bin/hipcc $FLAGS -c $PKG/csrc/$tu.hip -o $PKG/build/variants/${tu}_audit.o &
  OBJS="$OBJS $PKG/build/variants/${tu}_audit.o"
done
wait
OTHERS=$(ls $PKG/build/*.o | grep -v "/sdpa_fwd_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $PKG/lib/variants/libsdpa_hip_audit.so $OBJS $OTHERS -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo "built lib/variants/libsdpa_hip_audit.so"
