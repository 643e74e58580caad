#!/bin/bash
# round 5, call 13: tools/probes/d2h_probe.hip -- which engine carries the result rows home (the traces show __amd_rocclr_copyBuffer
# shader launches that end with the fused kernel beside them), can a copy engine be had, what does a zero-copy store kernel reach
O=gpurun_out/r05_13; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=$R/tools/probes/d2h_probe
timeout 120 $P 16777216 0 > $O/probe_default.log 2>&1; echo "default rc=$?"; cat $O/probe_default.log | cut -c1-200
timeout 120 $P 16777216 8 > $O/probe_reserve8.log 2>&1; grep "zero-copy\|pieces\|high-priority" $O/probe_reserve8.log | grep "beside" | cut -c1-200
for kv in GPU_FORCE_BLIT_COPY_SIZE=0 HSA_FORCE_SDMA_SIZE=1 HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=1 AMD_SERIALIZE_COPY=0 GPU_BLIT_ENGINE_TYPE=2; do
  env $kv timeout 120 $P 16777216 0 > $O/probe_$kv.log 2>&1; echo "== $kv rc=$?"; grep "portable" $O/probe_$kv.log | cut -c1-200
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/trace -o t -- $P 16777216 0 > $R/$O/trace.log 2>&1)
find $O/trace -name "*stats*.csv" | head; for f in $(find $O/trace -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do echo "-- $f"; head -12 $f | cut -c1-200; done
AMD_LOG_LEVEL=4 timeout 120 $P 4194304 0 2>&1 | grep -i "HSA Copy\|copy_engine\|blit\|sdma" | sort | uniq -c | sort -rn | head -30 | cut -c1-300 > $O/runtime_log_copy_lines.txt; head -30 $O/runtime_log_copy_lines.txt
rm -rf $O/trace
