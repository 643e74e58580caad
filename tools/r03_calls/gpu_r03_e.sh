#!/bin/bash
# round 3, call e: full GPU suite on the build with the unified dk-split kernel, the AVX-512 host converter
# and the qf-slot fix; host-convert A/B again; fp32 head-dim series (dk-split shapes unchanged?).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -15 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
timeout 900 python tools/gpu_hostlevel.py config2 headline config5:bf16 --hostcvt > $O/hostcvt_ab_pageable.log 2>$O/err.log
timeout 900 python tools/gpu_hostlevel.py config2 config5:bf16 --hostcvt --pinned > $O/hostcvt_ab_pinned.log 2>>$O/err.log
timeout 600 python tools/gpu_f32_dims.py > $O/f32_head_dims.log 2>>$O/err.log
tail -6 $O/pytest_gpu.log; cut -c1-330 $O/hostcvt_ab_pageable.log; cut -c1-330 $O/hostcvt_ab_pinned.log; cat $O/f32_head_dims.log | cut -c1-200; tail -3 $O/err.log
