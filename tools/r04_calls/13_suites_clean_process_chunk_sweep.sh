#!/bin/bash
# round 4, call 13: (a) the whole GPU suite three times with NO in-process registration left (the opt-in path runs in its own
# process: tests/test_gpu_register_optin.py), abort tracer armed; (b) boundary sweep of the chunk / piece knobs on the new
# default path (pageable arrays through the staging): does a larger last chunk pay now?
O=gpurun_out/r04_13; mkdir -p $O
export TMPDIR=/tmp
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
timeout 600 python tools/gpu_hostlevel.py headline config3 --sweep > $O/host_sweep_pageable.log 2>&1
grep '^{' $O/host_sweep_pageable.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'chunks', j['kv_chunks'], 'launches', j['fused_launches'])"
