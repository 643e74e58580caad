#!/bin/bash
# round 2, GPU call J: Q-image prescale (single rounding) build -- whole -m gpu suite, then d=256
# register-pressure variants of the duo kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 | cut -c1-220 > $O/pytest_gpu.log
for rep in 1 2; do
for tag in default kd2 vd1 nkb5; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/bf16.log
  timeout 200 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep shape | head -3 | cut -c1-110 >> $O/bf16.log
done; done
unset SDPA_HIP_LIB
cat $O/pytest_gpu.log; cat $O/bf16.log
