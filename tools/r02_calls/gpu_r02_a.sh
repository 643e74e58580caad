#!/bin/bash
# round 2, first GPU call: the new host pipeline (tests + timing sweeps), the kernel-rate grid the
# chunk/batch sizes are chosen from, the full -m gpu suite, one bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
export TMPDIR=/tmp
( rocm-smi --showproductname | head -20; nproc; free -g | head -2 ) > $O/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -q -x 2>&1 | tail -40 > $O/pytest_host.log
echo "host pipeline tests rc=$?" >> $O/pytest_host.log
timeout 300 python tools/gpu_kernel_grid.py 128 > $O/kernel_grid_f32.log 2>&1
timeout 300 python tools/gpu_hostlevel.py headline --sweep > $O/hostlevel_headline_pageable.log 2>&1
timeout 300 python tools/gpu_hostlevel.py headline --sweep --pinned > $O/hostlevel_headline_pinned.log 2>&1
timeout 300 python tools/gpu_hostlevel.py config2 config1 config4 config3 --pinned > $O/hostlevel_others_pinned.log 2>&1
timeout 300 python tools/gpu_hostlevel.py config2 config4 config5:bf16 > $O/hostlevel_others_pageable.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest_all.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -5 $O/pytest_host.log; tail -15 $O/pytest_all.log; cat $O/hostlevel_headline_pinned.log | head -20; cat $O/bench_n1.json | cut -c1-1500
