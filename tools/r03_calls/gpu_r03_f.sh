#!/bin/bash
# round 3, call f: (1) the DMA bounds audit build over the fp32 parity tests and config 3's shape, and the
# shipped build's config-3 device-level test in 10 fresh processes under AMD_LOG_LEVEL=1 (the
# unexplained abort); (2) rocprofv3 profiles that reproduce the bench lines (fp32 headline, fp32 d = 256,
# config 2, bf16 config 5); (3) bench lines and boundary timings of the BASELINE shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
mkdir -p $O
cd $R
export TMPDIR=/tmp
# ---- (1)
SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_dmaassert.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -m gpu -q -k "not bf16" 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-200 > $O/dma_bounds_audit.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  AMD_LOG_LEVEL=1 timeout 300 python -X faulthandler -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -k "config3_shape_device_level" > $O/abort_hunt_$i.log 2>&1
  echo "fresh process $i rc=$? $(grep -E 'passed|failed' $O/abort_hunt_$i.log | tail -1 | cut -c1-80)" >> $O/abort_hunt.log
done
rm -f $O/abort_hunt_*.log
# ---- (2)
timeout 900 bash tools/gpu_profile.sh r03 > $O/profile_headline.log 2>&1
BENCH_ARGS="--workload d256" timeout 900 bash tools/gpu_profile.sh r03_f32_d256 > $O/profile_f32_d256.log 2>&1
BENCH_ARGS="--workload config2" PROF_STEPS=40 timeout 900 bash tools/gpu_profile.sh r03_config2 > $O/profile_config2.log 2>&1
BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r03_config5_bf16 > $O/profile_config5_bf16.log 2>&1
python tools/merge_traffic.py gpurun_out/prof_r03/traffic.json gpurun_out/prof_r03_f32_d256/traffic.json gpurun_out/prof_r03_config2/traffic.json gpurun_out/prof_r03_config5_bf16/traffic.json > $O/merge_traffic.log 2>&1
cp profiles/traffic_latest.json $O/traffic_latest.json
for t in r03 r03_f32_d256 r03_config2 r03_config5_bf16; do
  mkdir -p $O/prof/$t
  cp $R/gpurun_out/prof_$t/summary.txt $R/gpurun_out/prof_$t/traffic.json $O/prof/$t/ 2>/dev/null
  find $R/gpurun_out/prof_$t/trace -name "*kernel_stats.csv" -exec cp {} $O/prof/$t/kernel_stats.csv \; 2>/dev/null
  rm -rf $R/gpurun_out/prof_$t
done
# ---- (3)  (the traffic file of this very build is in place now: the lines carry roofline.traffic / hbm_gbps / mfma_util)
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err
timeout 300 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
timeout 300 python bench.py --workload d256 --no-cpu-baseline > $O/bench_d256_f32.json 2>> $O/bench.err
timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-boundary --steps 5 > $O/bench_config3_one_gpu.json 2>> $O/bench.err
timeout 600 python tools/gpu_hostlevel.py headline config2 config1 config4 config3 config5:bf16 > $O/hostlevel_all_configs.log 2>> $O/bench.err
cat $O/dma_bounds_audit.log $O/abort_hunt.log; for t in r03 r03_f32_d256 r03_config2 r03_config5_bf16; do grep -A12 "== dominant kernel" $O/prof/$t/summary.txt | cut -c1-220; done; cat $O/merge_traffic.log; cut -c1-1200 $O/bench_n1.json; echo; for f in config2 d256_f32 config5_bf16 config3_one_gpu; do python -c "import json,sys; j=json.load(open('$O/bench_$f.json')); print('$f', round(j['ms_per_step'],4), j['roofline'])" 2>&1 | cut -c1-400; done; cut -c1-300 $O/hostlevel_all_configs.log; tail -3 $O/bench.err
