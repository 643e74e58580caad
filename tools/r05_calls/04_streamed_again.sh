#!/bin/bash
# round 5, call 4: the streamed launch with ready words raised by 64 KiB copies (the copy engine's) instead of 4-byte ones
# (a shader's: call 3).  Every later step is gated on the parity tests passing; the timeouts are short.
O=gpurun_out/r05_04; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/probes/stream_flag_probe2 > $O/stream_flag_probe2.log 2>&1; echo "probe2 rc=$?"; cut -c1-200 $O/stream_flag_probe2.log | tail -6
SDPA_STREAM_TIMEOUT_MS=1000 timeout 600 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_hosts_agree.py -m gpu -x -q -k "streamed or agree" > $O/streamed_tests.log 2>&1; rc=$?
echo "streamed tests rc=$rc $(grep -aE ' passed| failed' $O/streamed_tests.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "^FAILED\|Error\|assert\|sdpa:" $O/streamed_tests.log | head -20 | cut -c1-300; exit 1; fi
for sh in headline config2 config3 config4; do SDPA_STREAM_TIMEOUT_MS=1000 timeout 200 python tools/gpu_hostlevel.py $sh --streamed >> $O/streamed_ab.log 2>> $O/streamed_ab.err || break; done
SDPA_STREAM_TIMEOUT_MS=1000 timeout 200 python tools/gpu_hostlevel.py headline config2 --streamed --pinned >> $O/streamed_ab.log 2>> $O/streamed_ab.err
python - <<'P'
import json
for l in open('gpurun_out/r05_04/streamed_ab.log'):
    j=json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'launches', j['fused_launches'], 'streamed', j['streamed'])
P
tail -3 $O/streamed_ab.err | cut -c1-300
SDPA_STREAM_TIMEOUT_MS=1000 timeout 1200 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?
echo "suite rc=$rc $(grep -aE ' passed| failed' $O/suite.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|assert\|Error\|sdpa:" $O/suite.log | head -30 | cut -c1-300; fi
echo "amd_mem_obj lines in the suite log: $(grep -ac amd_mem_obj $O/suite.log)"
