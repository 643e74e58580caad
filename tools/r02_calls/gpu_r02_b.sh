#!/bin/bash
# round 2, second GPU call: host pipeline with the decoupled copy stream and head/tail row pieces
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_baseline_configs.py -q -x 2>&1 | tail -40 > $O/pytest_host.log
echo "host pipeline tests rc=$?" >> $O/pytest_host.log
timeout 300 python tools/gpu_hostlevel.py headline --sweep --pinned > $O/hostlevel_headline_pinned.log 2>&1
timeout 300 python tools/gpu_hostlevel.py headline config2 config1 config4 config3 config5:bf16 > $O/hostlevel_pageable.log 2>&1
timeout 300 python tools/gpu_hostlevel.py config2 --sweep --pinned > $O/hostlevel_config2_pinned.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -20 > $O/pytest_parity.log
tail -5 $O/pytest_host.log; tail -5 $O/pytest_parity.log; cat $O/hostlevel_headline_pinned.log | cut -c1-420; cat $O/hostlevel_pageable.log | cut -c1-420
