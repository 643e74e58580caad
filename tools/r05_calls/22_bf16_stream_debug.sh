#!/bin/bash
# round 5, call 22: why does ready word 1 never arrive at (8192, 16384, 512, 512) in bf16 (4 splits, 2 groups)?  the runtime's own log of the copies
O=gpurun_out/r05_22; mkdir -p $O
AMD_LOG_LEVEL=4 SDPA_STREAM_TIMEOUT_MS=300 timeout 120 python tools/gpu_bf16_stream_debug.py 8192 16384 512 512 > $O/out.log 2> $O/full.log
cat $O/out.log | cut -c1-400
sed -n '/==== CALL BEGINS/,$p' $O/full.log | grep -n "hipMemcpy2DAsync\|hipMemcpyAsync (\|HSA Copy\|Rect\|rect\|Blit\|blit\|ShaderName\|hipLaunchKernel\|hipModuleLaunch\|sdpa:" | cut -c1-260 | head -120 > $O/copies.log
wc -l $O/copies.log; cat $O/copies.log
rm -f $O/full.log
