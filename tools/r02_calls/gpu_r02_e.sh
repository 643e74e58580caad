#!/bin/bash
# round 2, GPU call E: duo kernel (all instantiations) parity + A/B timing over head dims
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -30 > $O/pytest_bf16.log
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py -q 2>&1 | tail -8 > $O/pytest_host.log
for rep in 1 2; do
SDPA_BF16_DUO=1 timeout 300 python tools/gpu_bf16_bench.py 256 128 64 >> $O/bf16_bench.log 2>&1
SDPA_BF16_DUO=0 timeout 300 python tools/gpu_bf16_bench.py 256 128 64 >> $O/bf16_bench.log 2>&1
done
tail -30 $O/pytest_bf16.log | cut -c1-200; tail -5 $O/pytest_host.log; grep shape $O/bf16_bench.log
