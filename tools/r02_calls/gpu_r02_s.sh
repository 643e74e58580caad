#!/bin/bash
# round 2, GPU call S: which softmax-slice change slows the duo kernel? (same-box A/B, d = 128 / 64 / 256)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
for rep in 1 2; do
for tag in oldbf16 product sl1 sl2 sl3; do
  echo "== $tag (rep $rep)" >> $O/ab.log
  if [ $tag = product ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  timeout 300 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep shape | grep -v 8192 | cut -c1-90 >> $O/ab.log
done
done
cat $O/ab.log
