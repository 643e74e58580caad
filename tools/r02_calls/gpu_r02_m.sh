#!/bin/bash
# round 2, GPU call M: the round's evidence -- full -m gpu log, rocprofv3 profiles (fp32 headline,
# bf16 at d = 64/128/256/512), bench lines, boundary timing of every BASELINE shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 | cut -c1-200 > $O/pytest_gpu.log
timeout 900 bash tools/gpu_profile.sh r02 > $O/profile_headline.log 2>&1
BENCH_ARGS="--workload headline --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d128 > $O/profile_d128.log 2>&1
BENCH_ARGS="--workload d256 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d256 > $O/profile_d256.log 2>&1
BENCH_ARGS="--workload d64 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d64 > $O/profile_d64.log 2>&1
BENCH_ARGS="--workload config5 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_config5_bf16 > $O/profile_config5.log 2>&1
cp $R/gpurun_out/prof_r02/traffic.json $R/profiles/traffic_latest.json 2>/dev/null
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload headline --precision bf16 --no-cpu-baseline > $O/bench_d128_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload d256 --precision bf16 --no-cpu-baseline > $O/bench_d256_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench_n1.err
timeout 300 python tools/gpu_hostlevel.py headline config2 config1 config4 config3 config5:bf16 > $O/hostlevel_pageable.log 2>&1
tail -5 $O/pytest_gpu.log; cut -c1-1800 $O/bench_n1.json; echo; cut -c1-700 $O/bench_config5_bf16.json $O/bench_d128_bf16.json $O/bench_d256_bf16.json; cut -c1-400 $O/hostlevel_pageable.log
