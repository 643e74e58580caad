#!/bin/bash
# round 2, GPU call L: build with the asm-MFMA lead wait states -- whole -m gpu suite, bf16 timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 | cut -c1-220 > $O/pytest_gpu.log
timeout 200 python tools/gpu_dbg.py 2>&1 | grep "^(" > $O/dbg.log
for rep in 1 2; do
  timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-110 >> $O/bf16.log
done
cat $O/pytest_gpu.log; cat $O/dbg.log | cut -c1-200; cat $O/bf16.log
