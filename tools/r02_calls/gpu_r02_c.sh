#!/bin/bash
# round 2, third GPU call: prefetch + CLI tests, one-shot CLI timing, the round's rocprofv3 profile,
# the rank-share dry run, the C host with 8 loopback ranks at config 3's shape.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -q -x 2>&1 | tail -15 > $O/pytest.log
echo "tests rc=$?" >> $O/pytest.log
timeout 600 bash tools/gpu_cli_timing.sh > $O/cli_timing.log 2>&1
timeout 300 python tools/gpu_hostlevel.py headline config5:bf16 config2 > $O/hostlevel.log 2>&1
SDPA_VIRTUAL_GPUS=8 timeout 300 python tools/gpu_hostlevel.py config3 headline > $O/hostlevel_virtual8.log 2>&1
timeout 900 bash tools/gpu_profile.sh r02 > $O/profile.log 2>&1
timeout 900 bash tools/gpu_emulate_ranks.sh > $O/emulate_ranks.log 2>&1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -4 $O/pytest.log; cat $O/cli_timing.log; cut -c1-400 $O/hostlevel.log $O/hostlevel_virtual8.log; tail -30 $O/profile.log; cat $O/emulate_ranks.log; cut -c1-1200 $O/bench_n1.json
