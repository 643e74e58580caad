#!/bin/bash
# round 2, GPU call U: row sums of the bf16 fixed-reference kernels from the rounded weights
# (v_dot2_f32_bf16, two per instruction) vs v_add_f32: same-box A/B + the bf16 suite on the variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02u
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
for rep in 1 2; do
for tag in product dot2; do
  echo "== $tag (rep $rep)" >> $O/ab.log
  if [ $tag = product ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-90 >> $O/ab.log
done
done
cat $O/ab.log
export SDPA_HIP_LIB=$V/libsdpa_hip_dot2.so
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_baseline_configs.py -q 2>&1 | tail -15 | cut -c1-300 > $O/pytest_dot2.log
cat $O/pytest_dot2.log
