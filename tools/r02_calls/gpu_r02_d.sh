#!/bin/bash
# round 2, GPU call D: the bf16 duo kernel (two query blocks per wave) -- parity, then A/B timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -25 > $O/pytest_bf16.log
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py -q 2>&1 | tail -15 > $O/pytest_host.log
for rep in 1 2; do
SDPA_BF16_DUO=1 timeout 300 python tools/gpu_bf16_bench.py 128 64 >> $O/bf16_bench.log 2>&1
SDPA_BF16_DUO=0 timeout 300 python tools/gpu_bf16_bench.py 128 64 >> $O/bf16_bench.log 2>&1
done
tail -25 $O/pytest_bf16.log; tail -8 $O/pytest_host.log; grep shape $O/bf16_bench.log
