#!/bin/bash
# round 5, call 17: the converter pool inside the boundary call, traced ($SDPA_HOST_CVT_TRACE=1: wake-up latency, busy time, rate per busy
# thread) at config 5 in bf16 and the metric shape; item size ($SDPA_HOST_CVT_ITEM_KB) x thread count
O=gpurun_out/r05_17; mkdir -p $O
export TMPDIR=/tmp
for sh in config5:bf16 headline; do
  SDPA_HOST_CVT_TRACE=1 timeout 200 python tools/gpu_hostlevel.py $sh > $O/trace_$sh.log 2> $O/trace_$sh.err
  echo "== $sh"; grep "hostcvt trace" $O/trace_$sh.err | tail -4 | cut -c1-420
done
for kb in 64 256 1024; do
  for th in 16 32 64; do
    SDPA_HOST_CVT_ITEM_KB=$kb SDPA_HOST_CVT_THREADS=$th SDPA_HOST_CVT_TRACE=1 timeout 200 python tools/gpu_hostlevel.py config5:bf16 2> $O/err_${kb}_$th.log | sed "s/^/item_kb=$kb threads=$th /" >> $O/items.log
    grep "hostcvt trace" $O/err_${kb}_$th.log | tail -1 | cut -c20-420
  done
done
python - <<'P'
import json
for l in open('gpurun_out/r05_17/items.log'):
    a, b, js = l.split(' ', 2); j = json.loads(js)
    print(a, b, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'])
P
