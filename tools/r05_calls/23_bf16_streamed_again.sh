#!/bin/bash
# round 5, call 23: the bf16 streamed tests with every code object loaded at rank creation; the fp32 streamed tests (nothing broke);
# config 5 in bf16 streamed at 32 / 48 / 64 pool threads (the transposing V converter is CPU-heavier than the rows)
O=gpurun_out/r05_23; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -k "streamed" > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log | cut -c1-150)"
grep -an "^FAILED\|^E  \|sdpa:" $O/tests.log | head -20 | cut -c1-300
for th in 32 48 64 32 64; do
  SDPA_HOST_CVT_THREADS=$th SDPA_HOST_CVT_TRACE=1 timeout 200 python tools/gpu_hostlevel.py config5:bf16 2> $O/err_$th.log | sed "s/^/threads=$th /" >> $O/threads.log
  grep "hostcvt trace" $O/err_$th.log | tail -1 | cut -c20-330
done
python - <<'P'
import json
for l in open('gpurun_out/r05_23/threads.log'):
    a, js = l.split(' ', 1); j = json.loads(js)
    print(a, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'launch', j['kernel_ms'], 'streamed', j['streamed'], 'pool', j['host_convert_threads'])
P
