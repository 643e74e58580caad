#!/bin/bash
# round 5, call 31: what do the persistent form's waits and arguments cost the KERNEL?  fused_bf16_tandem_stream_kernel<512> on resident images with every
# ready word raised beforehand (tools build, $SDPA_TUNE bit 12) against the classic launch, config 5's shape, interleaved
O=gpurun_out/r05_31; mkdir -p $O
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
lib=$PWD/$PKG/lib/variants/libsdpa_hip_bfabl.so
for rep in 1 2 3; do
  for tune in 0 4096; do
    SDPA_HIP_LIB=$lib SDPA_TUNE=$tune timeout 200 python tools/gpu_bf16_bench.py 512 2>/dev/null | head -1 | sed "s/^/tune=$tune /" >> $O/ab.log
  done
done
cut -c1-130 $O/ab.log
