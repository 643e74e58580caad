#!/bin/bash
# round 3: the tail-quantisation-aware K/V split choice against round 2's, then the profiles again (the fp32 and
# bf16 kernel sources changed by host code only, but the stamps are source hashes), then the full suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03splits
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
mkdir -p $O
cd $R
export TMPDIR=/tmp
(echo "## round-2 library (lib/variants/libsdpa_hip_r02.so)"; SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_r02.so timeout 300 python tools/gpu_odd_rows.py; SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_r02.so timeout 300 python tools/gpu_odd_rows.py --bf16; echo "## shipped"; timeout 300 python tools/gpu_odd_rows.py; timeout 300 python tools/gpu_odd_rows.py --bf16) > $O/odd_row_counts_split_choice_ab.log 2>$O/err.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-200 > $O/pytest_gpu_final.log
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
timeout 600 python bench.py > $O/bench_n1.json 2>> $O/err.log
timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2>> $O/err.log
cat $O/odd_row_counts_split_choice_ab.log; cat $O/pytest_gpu_final.log; for t in r03 r03_f32_d256 r03_config2 r03_config5_bf16; do grep -A4 "== dominant kernel" $O/prof/$t/summary.txt | tail -2 | cut -c1-160; done; cat $O/merge_traffic.log; for f in n1 config5_bf16; do python -c "import json; j=json.load(open('$O/bench_$f.json')); r=j['roofline']; print('$f', round(j['ms_per_step'],4), round(r['kernel_ms_avg'],4), round(r['frac'],4), r['traffic'], r['mfma_util'])"; done; tail -2 $O/err.log
