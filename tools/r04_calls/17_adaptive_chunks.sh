#!/bin/bash
# round 4, call 17: largest chunk chosen by feed time vs kernel time (65536 / 16384 / 8192 keys), stage-ahead 1: parity of the
# host pipeline and the BASELINE configs, then the boundary of every shape, default path and page-locked callers, twice
O=gpurun_out/r04_17; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_baseline_configs.py tests/test_gpu_hosts_agree.py tests/test_gpu_parity.py -q > $O/pytest_host.log 2>&1; echo "pytest rc=$? $(grep -aE ' passed| failed' $O/pytest_host.log | tail -1 | cut -c1-100)"
grep -a "^FAILED" $O/pytest_host.log | head -5 | cut -c1-200
for rep in 1 2; do
timeout 600 python tools/gpu_hostlevel.py headline config2 config3 config4 config5:bf16 config5 >> $O/hostlevel_default.log 2>&1
timeout 600 python tools/gpu_hostlevel.py headline config2 config3 config4 config5:bf16 --pinned >> $O/hostlevel_pinned.log 2>&1
done
cat $O/hostlevel_default.log $O/hostlevel_pinned.log | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kvstage', j['kv_stage_ms'], 'kernel', j['kernel_ms'], 'chunks', j['kv_chunks'], 'cvt', j['host_convert_threads'], 'widen', j['host_widen'])"
