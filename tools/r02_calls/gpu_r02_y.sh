#!/bin/bash
# round 2, GPU call Y: fp32 pipelined kernel at dense 256-wide dims (one wave per SIMD): parity, rates
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02y
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
timeout 300 python tools/gpu_f32_dims.py 128 256 384 512 > $O/f32_dims.log 2>&1
cat $O/f32_dims.log
