#!/bin/bash
# round 2, GPU call R: same-box A/B of the bf16 kernels: HEAD's kernels (oldbf16) vs the trimmed
# softmax VALU (product build) vs the same + scalar end-of-shard fix (sclamp, timing only); then
# the bf16 suite on the product build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
for rep in 1 2; do
for tag in oldbf16 product sclamp; do
  echo "== $tag (rep $rep)" >> $O/ab.log
  if [ $tag = product ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-110 >> $O/ab.log
done
done
unset SDPA_HIP_LIB
cat $O/ab.log
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_host_pipeline.py tests/test_gpu_baseline_configs.py -q -x 2>&1 | tail -15 | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
