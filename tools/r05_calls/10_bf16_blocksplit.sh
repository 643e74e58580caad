#!/bin/bash
# round 5, call 10: the tandem loop WITHOUT the accumulator tile that hipcc keeps in architectural VGPRs across the back edge
# (16 v_accvgpr_write + 16 v_accvgpr_read + a full MFMA drain per two steps): a never-taken scalar branch behind each barrier
# (-DSDPA_TANDEM_SKEW=-1) changes the live-range split -- same work, same order; against the shipped loop, interleaved
O=gpurun_out/r05_10; mkdir -p $O
export TMPDIR=/tmp
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
for rep in 1 2 3 4; do
  for tag in base tsplit; do
    lib=$PWD/$PKG/lib/variants/libsdpa_hip_$tag.so; [ $tag = base ] && lib=$PWD/$PKG/lib/libsdpa_hip.so
    SDPA_HIP_LIB=$lib timeout 200 python tools/gpu_bf16_bench.py 512 2>/dev/null | head -1 | sed "s/^/$tag /" >> $O/blocksplit_ab.log
  done
done
cat $O/blocksplit_ab.log | cut -c1-120
SDPA_HIP_LIB=$PWD/$PKG/lib/variants/libsdpa_hip_tsplit.so timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x > $O/pytest_bf16_tsplit.log 2>&1; echo "bf16 tests on tsplit rc=$? $(tail -1 $O/pytest_bf16_tsplit.log)"
