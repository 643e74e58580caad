#!/bin/bash
# long differential fuzz on the GPU box: SDPA_FUZZ_CASES cases per test of tests/test_gpu_fuzz.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fuzz
mkdir -p $O
cd $R
export TMPDIR=/tmp
SDPA_FUZZ_CASES=${1:-600} timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -s 2>&1 | grep -E "worst|passed|failed|FAILED|^E |Error" | cut -c1-400 | tee $O/fuzz.log
