#!/bin/bash
# round 2, GPU call F: duo kernel LDS ring depth A/B (variant libraries), parity of the default build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -12 > $O/pytest_bf16.log
for rep in 1 2; do
for tag in default nkb3 nkb4 vd3; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/bf16_bench.log
  timeout 300 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep shape >> $O/bf16_bench.log
done; done
unset SDPA_HIP_LIB
tail -6 $O/pytest_bf16.log | cut -c1-200; cat $O/bf16_bench.log | cut -c1-150
