#!/bin/bash
# round 4, call 19: (a) two more whole suites on another box, abort tracer armed; (b) one rank's 1/8 and 1/2 share of the metric
# shape on one GPU, whole chip against 16 CUs' worth of slots left free (what bench.py --gpus N launches on): the compute side
# of the N > 1 sizing
O=gpurun_out/r04_19; mkdir -p $O
export TMPDIR=/tmp
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
for n in 8 2; do for r in 0 16; do
  timeout 300 python bench.py --emulate-ranks $n --reserve-cus $r --no-cpu-baseline --no-boundary --no-scaling-record --min-gpu-seconds 0 --steps 40 2>/dev/null | tail -1 > $O/rank_share_${n}_reserve$r.json
  python -c "
import json; j=json.load(open('$O/rank_share_${n}_reserve$r.json'))
print('1/$n share, reserve $r:', 'ms_per_step', round(j['ms_per_step'],4), 'kernel_ms_avg', round(j['roofline']['kernel_ms_avg'],4), 'frac', round(j['roofline']['frac'],4))"
done; done
