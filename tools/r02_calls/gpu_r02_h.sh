#!/bin/bash
# round 2, GPU call H: where the duo kernel's time goes (timing-only ablation variants), and parity
# of the default build after moving the per-tile closing work to the end of the kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -4 > $O/pytest_bf16.log
for rep in 1 2; do
for tag in default abl1 abl2 abl4 abl3 abl7 abl15; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/abl.log
  timeout 120 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep shape | head -3 | cut -c1-110 >> $O/abl.log
done; done
unset SDPA_HIP_LIB
cat $O/pytest_bf16.log; cat $O/abl.log
