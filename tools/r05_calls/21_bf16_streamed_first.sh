#!/bin/bash
# round 5, call 21: the bf16 form of the streamed first batch -- its two tests, then config 5 in bf16 at the boundary, streamed vs one
# launch per chunk, interleaved, with the converter pool's trace
O=gpurun_out/r05_21; mkdir -p $O
export TMPDIR=/tmp
SDPA_STREAM_TIMEOUT_MS=1500 timeout 600 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -x -k "streamed_bf16" > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log | cut -c1-150)"
grep -an "^FAILED\|^E  \|Error\|error" $O/tests.log | head -20 | cut -c1-300
SDPA_HOST_CVT_TRACE=1 SDPA_STREAM_TIMEOUT_MS=1500 timeout 300 python tools/gpu_hostlevel.py config5:bf16 --streamed > $O/ab.log 2> $O/ab.err; echo "ab rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r05_21/ab.log'):
    j = json.loads(l); print(j['shape'], j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'launches', j['fused_launches'], 'streamed', j['streamed'], j['last_kernel'][:40])
P
grep "hostcvt trace" $O/ab.err | tail -3 | cut -c1-400; grep -v "hostcvt trace" $O/ab.err | tail -5 | cut -c1-300
