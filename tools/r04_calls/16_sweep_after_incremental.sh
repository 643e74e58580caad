#!/bin/bash
# round 4, call 16: the chunk / piece knob sweep again, now that chunks are staged incrementally (default path, pageable arrays)
O=gpurun_out/r04_16; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/gpu_hostlevel.py headline config3 config5:bf16 --sweep > $O/host_sweep_pageable.log 2>&1
grep '^{' $O/host_sweep_pageable.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'chunks', j['kv_chunks'], 'launches', j['fused_launches'])"
