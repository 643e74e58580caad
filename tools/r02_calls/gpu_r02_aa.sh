#!/bin/bash
# round 2, GPU call AA: fp32 operand images padded to 64/128/256 columns -> pipelined kernel for every
# head dim in (32, 256]: parity suites, fuzz, rates at odd head dims
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02aa
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
timeout 600 python tools/gpu_f32_dims.py 64 80 96 100 128 160 192 200 256 > $O/f32_dims.log 2>&1
grep tflops $O/f32_dims.log
SDPA_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -s 2>&1 | grep -E "worst|passed|failed|FAILED|^E " | cut -c1-300 > $O/fuzz.log
cat $O/fuzz.log
