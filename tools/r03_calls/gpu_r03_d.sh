#!/bin/bash
# round 3, call d: host-side converts ($SDPA_HOST_CVT=1) -- bitwise tests, then the boundary A/B over the
# BASELINE shapes and thread counts.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -x -q -k "host_side_convert or schedules_agree" 2>&1 | grep -v Warning | tail -15 > $O/pytest_hostcvt.log
nproc > $O/host.txt; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread" >> $O/host.txt
timeout 900 python tools/gpu_hostlevel.py config2 headline config5:bf16 config5 config1 --hostcvt > $O/hostcvt_ab_pageable.log 2>$O/err.log
timeout 900 python tools/gpu_hostlevel.py config2 headline config5:bf16 --hostcvt --pinned > $O/hostcvt_ab_pinned.log 2>>$O/err.log
tail -6 $O/pytest_hostcvt.log; cat $O/host.txt; cut -c1-330 $O/hostcvt_ab_pageable.log; cut -c1-330 $O/hostcvt_ab_pinned.log; tail -3 $O/err.log
