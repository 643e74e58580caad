#!/bin/bash
# round 3, final evidence on one box: full -m gpu suite + smoke(), rocprofv3 profiles of the fp32 workloads
# (headline, d = 256, config 2; the bf16 config-5 profile of call f still carries the current source stamp),
# bench lines, boundary timings.  Outputs land in gpurun_out/r03final/ and are copied to profiles/r03/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03final
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -z "$1" ]; then
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-200 > $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu_final.log 2>&1
timeout 900 bash tools/gpu_profile.sh r03 > $O/profile_headline.log 2>&1
BENCH_ARGS="--workload d256" timeout 900 bash tools/gpu_profile.sh r03_f32_d256 > $O/profile_f32_d256.log 2>&1
BENCH_ARGS="--workload config2" PROF_STEPS=40 timeout 900 bash tools/gpu_profile.sh r03_config2 > $O/profile_config2.log 2>&1
python tools/merge_traffic.py gpurun_out/prof_r03/traffic.json gpurun_out/prof_r03_f32_d256/traffic.json gpurun_out/prof_r03_config2/traffic.json > $O/merge_traffic.log 2>&1
cp profiles/traffic_latest.json $O/traffic_latest.json
for t in r03 r03_f32_d256 r03_config2; do
  mkdir -p $O/prof/$t
  cp $R/gpurun_out/prof_$t/summary.txt $R/gpurun_out/prof_$t/traffic.json $O/prof/$t/ 2>/dev/null
  find $R/gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/$t/kernel_stats.csv \; 2>/dev/null
  rm -rf $R/gpurun_out/prof_$t
done
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err
timeout 300 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
timeout 300 python bench.py --workload d256 --no-cpu-baseline > $O/bench_d256_f32.json 2>> $O/bench.err
timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-boundary --steps 5 > $O/bench_config3_one_gpu.json 2>> $O/bench.err
timeout 300 python bench.py --emulate-ranks 8 --no-cpu-baseline > $O/bench_one_rank_share_of_8.json 2>> $O/bench.err
timeout 600 python tools/gpu_hostlevel.py headline config2 config1 config4 config3 config5:bf16 > $O/hostlevel_all_configs.log 2>> $O/bench.err
cat $O/pytest_gpu_final.log; for t in r03 r03_f32_d256 r03_config2; do grep -A4 "== dominant kernel" $O/prof/$t/summary.txt | cut -c1-200; done; cat $O/merge_traffic.log; for f in n1 config2 d256_f32 config5_bf16 config3_one_gpu one_rank_share_of_8; do python -c "import json,sys; j=json.load(open('$O/bench_$f.json')); r=j['roofline']; print('$f', round(j['ms_per_step'],4), round(r['kernel_ms_avg'],4), round(r['frac'],4), r['traffic'], r['hbm_gbps'], r['mfma_util'], j.get('parity_max_err'))" 2>&1 | cut -c1-300; done; cut -c1-260 $O/hostlevel_all_configs.log; tail -3 $O/bench.err
fi

# ---- addendum (after the tandem bf16 kernel became the dv > 256 default): full suite again, the bf16 config-5
# profile with the new source stamp, its bench line and boundary, the bf16 head-dim series, a short fuzz
if [ "$1" = "bf16" ]; then
  O=$R/gpurun_out/r03final_bf16
  mkdir -p $O
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-200 > $O/pytest_gpu_final.log
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu_final.log 2>&1
  BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r03_config5_bf16 > $O/profile_config5_bf16.log 2>&1
  python tools/merge_traffic.py gpurun_out/prof_r03_config5_bf16/traffic.json > $O/merge_traffic.log 2>&1
  cp profiles/traffic_latest.json $O/traffic_latest.json
  mkdir -p $O/prof/r03_config5_bf16
  cp $R/gpurun_out/prof_r03_config5_bf16/summary.txt $R/gpurun_out/prof_r03_config5_bf16/traffic.json $O/prof/r03_config5_bf16/ 2>/dev/null
  find $R/gpurun_out/prof_r03_config5_bf16/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/r03_config5_bf16/kernel_stats.csv \; 2>/dev/null
  rm -rf $R/gpurun_out/prof_r03_config5_bf16
  timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/bench.err
  timeout 300 python tools/gpu_hostlevel.py config5:bf16 > $O/hostlevel_config5_bf16.log 2>> $O/bench.err
  timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-200 > $O/bf16_head_dims.log
  SDPA_FUZZ_CASES=150 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -s -k "bf16 or loopback" 2>&1 | grep -E "worst|passed|failed|FAILED|^E " | cut -c1-300 > $O/fuzz_bf16.log
  cat $O/pytest_gpu_final.log; grep -A12 "== dominant kernel" $O/prof/r03_config5_bf16/summary.txt | cut -c1-220; python -c "import json; j=json.load(open('$O/bench_config5_bf16.json')); print(round(j['ms_per_step'],4), j['roofline'], j['parity_max_err'])" | cut -c1-500; cut -c1-300 $O/hostlevel_config5_bf16.log; cat $O/bf16_head_dims.log $O/fuzz_bf16.log; tail -2 $O/bench.err
fi

# ---- addendum 2 (after the fp32 dk-split kernel became software-pipelined; the fp32 source stamp now covers
# sdpa_fwd_f32_dksplit.hip and sdpa_f32_device.h too): full suite, every fp32 profile again, bench lines
if [ "$1" = "f32b" ]; then
  O=$R/gpurun_out/r03final_f32b
  mkdir -p $O
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-200 > $O/pytest_gpu_final.log
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu_final.log 2>&1
  timeout 600 bash tools/gpu_profile.sh r03 > $O/profile_headline.log 2>&1
  BENCH_ARGS="--workload d256" timeout 600 bash tools/gpu_profile.sh r03_f32_d256 > $O/profile_f32_d256.log 2>&1
  BENCH_ARGS="--workload config2" PROF_STEPS=40 timeout 600 bash tools/gpu_profile.sh r03_config2 > $O/profile_config2.log 2>&1
  BENCH_ARGS="--workload config3" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r03_config3 > $O/profile_config3.log 2>&1
  BENCH_ARGS="--workload config4" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r03_config4 > $O/profile_config4.log 2>&1
  BENCH_ARGS="--workload config5" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r03_config5_f32 > $O/profile_config5_f32.log 2>&1
  TAGS="r03 r03_f32_d256 r03_config2 r03_config3 r03_config4 r03_config5_f32"
  python tools/merge_traffic.py $(for t in $TAGS; do echo gpurun_out/prof_$t/traffic.json; done) > $O/merge_traffic.log 2>&1
  cp profiles/traffic_latest.json $O/traffic_latest.json
  for t in $TAGS; do
    mkdir -p $O/prof/$t
    cp $R/gpurun_out/prof_$t/summary.txt $R/gpurun_out/prof_$t/traffic.json $O/prof/$t/ 2>/dev/null
    find $R/gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/$t/kernel_stats.csv \; 2>/dev/null
    rm -rf $R/gpurun_out/prof_$t
  done
  timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err
  timeout 300 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
  timeout 300 python bench.py --workload d256 --no-cpu-baseline > $O/bench_d256_f32.json 2>> $O/bench.err
  timeout 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5_f32.json 2>> $O/bench.err
  timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-boundary --steps 5 > $O/bench_config3_one_gpu.json 2>> $O/bench.err
  timeout 300 python tools/gpu_hostlevel.py config5 > $O/hostlevel_config5_f32.log 2>> $O/bench.err
  timeout 300 python tools/gpu_f32_dims.py 128 256 320 384 512 640 768 1024 2>&1 | grep tflops > $O/f32_head_dims.log
  cat $O/pytest_gpu_final.log; for t in $TAGS; do grep -A4 "== dominant kernel" $O/prof/$t/summary.txt | cut -c1-200; done; cat $O/merge_traffic.log
  for f in n1 config2 d256_f32 config5_f32 config3_one_gpu; do python -c "import json,sys; j=json.load(open('$O/bench_$f.json')); r=j['roofline']; print('$f', round(j['ms_per_step'],4), round(r['kernel_ms_avg'],4), round(r['frac'],4), r['traffic'], r['hbm_gbps'], r['mfma_util'], j.get('parity_max_err'))" 2>&1 | cut -c1-300; done
  cut -c1-300 $O/hostlevel_config5_f32.log; cat $O/f32_head_dims.log; tail -3 $O/bench.err
fi
