#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02k2
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
for tag in default pinq; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/dbg.log
  timeout 200 python tools/gpu_dbg2.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" >> $O/dbg.log
  timeout 200 python tools/gpu_dbg.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | head -4 >> $O/dbg.log
done
unset SDPA_HIP_LIB
cat $O/dbg.log
