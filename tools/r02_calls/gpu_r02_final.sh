#!/bin/bash
# round 2, final evidence: full -m gpu log, rocprofv3 profiles (fp32 headline, fp32 d=256, bf16 at
# d = 64/128/256/512), bench lines, boundary timing of every BASELINE shape, kernel-only head-dim series
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02final
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-200 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
timeout 900 bash tools/gpu_profile.sh r02 > $O/profile_headline.log 2>&1
cp $R/gpurun_out/prof_r02/traffic.json $R/profiles/traffic_latest.json 2>/dev/null
BENCH_ARGS="--workload d256" timeout 600 bash tools/gpu_profile.sh r02_f32_d256 > $O/profile_f32_d256.log 2>&1
BENCH_ARGS="--workload headline --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d128 > $O/profile_d128.log 2>&1
BENCH_ARGS="--workload d256 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d256 > $O/profile_d256.log 2>&1
BENCH_ARGS="--workload d64 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_bf16_d64 > $O/profile_d64.log 2>&1
BENCH_ARGS="--workload config5 --precision bf16" timeout 600 bash tools/gpu_profile.sh r02_config5_bf16 > $O/profile_config5.log 2>&1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload headline --precision bf16 --no-cpu-baseline > $O/bench_d128_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload d256 --precision bf16 --no-cpu-baseline > $O/bench_d256_bf16.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload d256 --no-cpu-baseline > $O/bench_d256_f32.json 2>> $O/bench_n1.err
timeout 300 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench_n1.err
timeout 300 python tools/gpu_hostlevel.py headline config2 config1 config4 config3 config5:bf16 > $O/hostlevel_pageable.log 2>&1
for rep in 1 2; do timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-200 >> $O/bf16_head_dims.log; done
# keep only the summaries of the profile directories (the merge-back is capped)
for t in r02 r02_f32_d256 r02_bf16_d128 r02_bf16_d256 r02_bf16_d64 r02_config5_bf16; do
  mkdir -p $O/prof/$t
  cp $R/gpurun_out/prof_$t/summary.txt $O/prof/$t/ 2>/dev/null
  cp $R/gpurun_out/prof_$t/traffic.json $O/prof/$t/ 2>/dev/null
  find $R/gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/$t/kernel_stats.csv \; 2>/dev/null
done
cat $O/pytest_gpu.log; cut -c1-1500 $O/bench_n1.json; echo; cut -c1-600 $O/bench_config5_bf16.json $O/bench_d128_bf16.json $O/bench_d256_bf16.json $O/bench_d256_f32.json; cut -c1-330 $O/hostlevel_pageable.log; cat $O/bf16_head_dims.log | cut -c1-100
