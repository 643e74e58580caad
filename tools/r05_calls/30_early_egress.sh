#!/bin/bash
# round 5, call 30 (the early-egress code this call measured was NOT kept -- profiles/r05/bf16_stream_kernel_resident_ab.log; the knob below no longer exists): early egress of the streamed bf16 launch (rows of a Q row piece leave when its workgroups have retired): its test, the streamed tests,
# and config 5 in bf16 at the boundary with $SDPA_STREAM_EARLY_EGRESS=1 / 0, interleaved
O=gpurun_out/r05_30; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -s -k "streamed" > $O/tests.log 2>&1; echo "tests rc=$? $(grep -aE ' passed| failed' $O/tests.log | tail -1 | cut -c1-150)"
grep -an "^FAILED\|^E  \|sdpa:\|early_egress_pieces per call" $O/tests.log | head -20 | cut -c1-300
for rep in 1 2 3; do for e in 1 0; do
  SDPA_STREAM_EARLY_EGRESS=$e timeout 200 python tools/gpu_hostlevel.py config5:bf16 2>/dev/null | sed "s/^/early=$e /" >> $O/ab.log
done; done
python - <<'P'
import json
for l in open('gpurun_out/r05_30/ab.log'):
    a, js = l.split(' ', 1); j = json.loads(js)
    print(a, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'launch', j['kernel_ms'], 'streamed', j['streamed'])
P
