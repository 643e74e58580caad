#!/bin/bash
# round 2, GPU call N: re-check the bf16 suite after the score-range change (+-80), smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_host_pipeline.py -q 2>&1 | tail -5 | cut -c1-200 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest.log 2>&1
cat $O/pytest.log
