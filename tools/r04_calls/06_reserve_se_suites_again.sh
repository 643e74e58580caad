#!/bin/bash
# round 4, call 6 (= call 5 again: its outputs were lost with the container): (a) how many workgroup slots must stay free before another stream's kernel actually runs beside the
# fused kernel (call 3: with 8 CUs' worth free the merge kernel still waited for the fused kernel's end) -- 16 / 32 / 64;
# (b) the whole GPU suite twice on the library with whole-page registrations; (c) the bench line
O=gpurun_out/r04_06; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in 16 32 64; do
  tag=reserve$mode
  (cd /tmp && SDPA_FORCE_COLLECTIVES=1 SDPA_COMM_CUS=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$tag -o t -- python $R/tools/gpu_hostlevel.py config4 > $R/$O/trace_$tag.log 2>&1)
  python tools/summarize_overlap.py $O/trace_$tag > $O/config4_one_rank_forced_collectives_overlap_$tag.txt 2>&1
  tail -1 $O/config4_one_rank_forced_collectives_overlap_$tag.txt; grep total_ms $O/trace_$tag.log | tail -1 | cut -c1-200
  grep merge_gathered $O/config4_one_rank_forced_collectives_overlap_$tag.txt | head -4 | cut -c1-110
  rm -rf $O/trace_$tag
done
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|Error\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('$O/bench_n1.json'))
print(j['ms_per_step'], j['roofline']['frac'], j['boundary']['pinned_caller_arrays'], j['boundary']['ms'])"
python tools/gpu_hostlevel.py config2 headline --pinned 2>&1 | tail -3 | cut -c1-400
# (d) can the fault be provoked on purpose?  pageable sources, asynchronous copies (tools/gpu_pageable_async_stress.py)
timeout 600 python tools/gpu_pageable_async_stress.py 20 > $O/pageable_async_stress.log 2>&1; cat $O/pageable_async_stress.log | cut -c1-400
