#!/bin/bash
# round 5, call 8: bf16 tandem kernel with the four waves' LDS-DMA pieces STAGGERED over the MFMA gaps (-DSDPA_TANDEM_STAGGER=1;
# VSHIFT = which gap the Vt pieces take relative to the K pieces) against the shipped loop, interleaved, same box; then the
# tandem tests (bit-identity with the wide kernel) on the staggered library
O=gpurun_out/r05_08; mkdir -p $O
export TMPDIR=/tmp
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
for rep in 1 2 3; do
  for tag in base tstag tstag0 tstag2 tstag3; do
    lib=$PWD/$PKG/lib/variants/libsdpa_hip_$tag.so; [ $tag = base ] && lib=$PWD/$PKG/lib/libsdpa_hip.so
    SDPA_HIP_LIB=$lib timeout 200 python tools/gpu_bf16_bench.py 512 2>/dev/null | head -1 | sed "s/^/$tag /" >> $O/stagger_ab.log
  done
done
cat $O/stagger_ab.log | cut -c1-120
SDPA_HIP_LIB=$PWD/$PKG/lib/variants/libsdpa_hip_tstag.so timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x > $O/pytest_bf16_tstag.log 2>&1; echo "bf16 tests on tstag rc=$? $(tail -1 $O/pytest_bf16_tstag.log)"
