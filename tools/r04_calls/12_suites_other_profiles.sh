#!/bin/bash
# round 4, call 12: (a) the whole GPU suite three times, abort tracer armed, registration off (the default);
# (b) the other workloads' profiles re-taken on the round's sources (config 2 / 3 / 4, d = 256, config 5's dims in fp32)
O=gpurun_out/r04_12; mkdir -p $O
export TMPDIR=/tmp
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
BENCH_ARGS="--workload config2" PROF_STEPS=40 timeout 600 bash tools/gpu_profile.sh r04_config2 2>&1 | tail -12
BENCH_ARGS="--workload d256" timeout 600 bash tools/gpu_profile.sh r04_f32_d256 2>&1 | tail -12
BENCH_ARGS="--workload config3" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r04_config3 2>&1 | tail -12
BENCH_ARGS="--workload config4" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r04_config4 2>&1 | tail -12
BENCH_ARGS="--workload config5" PROF_STEPS=5 timeout 600 bash tools/gpu_profile.sh r04_config5_f32 2>&1 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh gpurun_out
