#!/bin/bash
# round 4, call 3: (a) do the merge tail's kernels run under the next batch's fused kernel with slots reserved by grid
# size; (b) the whole GPU suite, abort tracer armed, core dumps on; (c) the dk-split / bf16 / config tests on the
# bounds-AUDIT build; (d) rocprofv3 profile of the headline for profiles/traffic_latest.json
O=gpurun_out/r04_03; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in 0 8; do
  tag=reserve$mode
  (cd /tmp && SDPA_FORCE_COLLECTIVES=1 SDPA_COMM_CUS=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$tag -o t -- python $R/tools/gpu_hostlevel.py config4 > $R/$O/trace_$tag.log 2>&1)
  python tools/summarize_overlap.py $O/trace_$tag > $O/config4_one_rank_forced_collectives_overlap_$tag.txt 2>&1
  tail -1 $O/config4_one_rank_forced_collectives_overlap_$tag.txt; grep total_ms $O/trace_$tag.log | tail -1 | cut -c1-200
  rm -rf $O/trace_$tag
done
# (b)
ulimit -c unlimited; cat /proc/sys/kernel/core_pattern > $O/core_pattern.txt 2>&1
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -E 'passed|failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "SIGABRT\|Fatal\|Segmentation\|Abort\|fault\|File \".*tests\|error" $O/suite_run_$i.log | head -30 | cut -c1-300; ls -la core* /tmp/core* 2>/dev/null | head; fi
done
# (c)
SDPA_HIP_LIB=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants/libsdpa_hip_audit.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -q > $O/audit_build_suite.log 2>&1; echo "audit rc=$?"; grep -a "DMA bounds audit\|passed\|failed" $O/audit_build_suite.log | tail -3
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
# (d)
bash tools/gpu_profile.sh r04_headline > $O/profile_headline.log 2>&1; tail -16 $O/profile_headline.log
cp gpurun_out/prof_r04_headline/summary.txt $O/headline_f32_rocprofv3_summary.txt 2>/dev/null; cp gpurun_out/prof_r04_headline/traffic.json $O/headline_traffic.json 2>/dev/null
find gpurun_out/prof_r04_headline -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/headline_f32_kernel_stats.csv
rm -rf gpurun_out/prof_r04_headline/trace gpurun_out/prof_r04_headline/pmc_*
