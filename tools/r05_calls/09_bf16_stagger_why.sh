#!/bin/bash
# round 5, call 9: why the staggered loop of call 8 was 17 % SLOWER -- four copies of the SAME loop (tcopy: the copies' cost
# alone), and the same code with the waves 16 / 32 cycles apart behind every barrier (tskew1 / tskew2: the phases' cost alone)
O=gpurun_out/r05_09; mkdir -p $O
export TMPDIR=/tmp
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
for rep in 1 2 3; do
  for tag in base tcopy tskew1 tskew2 tstag; do
    lib=$PWD/$PKG/lib/variants/libsdpa_hip_$tag.so; [ $tag = base ] && lib=$PWD/$PKG/lib/libsdpa_hip.so
    SDPA_HIP_LIB=$lib timeout 200 python tools/gpu_bf16_bench.py 512 2>/dev/null | head -1 | sed "s/^/$tag /" >> $O/stagger_why_ab.log
  done
done
cat $O/stagger_why_ab.log | cut -c1-120
