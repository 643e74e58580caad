#!/bin/bash
# round 2, GPU call Z: dk-split kernel with one query block per workgroup (512 < dk <= 1024): parity, rates
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_abi.py -q -x 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
timeout 600 python tools/gpu_f32_dims.py 384 512 768 1024 > $O/f32_dims.log 2>&1
cat $O/f32_dims.log
