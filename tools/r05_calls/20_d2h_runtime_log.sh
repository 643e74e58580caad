#!/bin/bash
# round 5, call 20: the runtime's own log of one boundary call's device->host copies (why are they shader launches in THIS process?)
O=gpurun_out/r05_20; mkdir -p $O
AMD_LOG_LEVEL=4 timeout 200 python tools/gpu_hostlevel.py config2 > $O/out.log 2> $O/full.log
grep -n "hipMemcpyAsync\|HSA Copy\|copyBuffer\|Blit\|blit\|Query copy engine\|hipHostMalloc\|hipMemcpy2D" $O/full.log | tail -80 | cut -c1-330 > $O/copy_lines.log
wc -l $O/full.log; tail -60 $O/copy_lines.log
rm -f $O/full.log
