#!/bin/bash
# round 2, GPU call G: duo ring depths -- correctness (duo tests) and timing per variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g2
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
for tag in default nkb3 nkb4 nkb5 nkb6 vd3; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/variants.log
  timeout 300 python -m pytest tests/test_gpu_bf16.py -q -k "duo or shapes" 2>&1 | grep -E "FAILED|passed|failed" | cut -c1-110 >> $O/variants.log
  timeout 120 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep shape | head -3 | cut -c1-120 >> $O/variants.log
done
unset SDPA_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -4 >> $O/variants.log
cat $O/variants.log
