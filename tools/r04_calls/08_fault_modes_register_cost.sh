#!/bin/bash
# round 4, call 8: call 7's reproducer faulted in 5-11 s in every mode that REGISTERS caller arrays (refused, probed,
# register).  (a) Does it stay clean when nothing is registered (no_register, staged)?  register once more as the control.
# (b) what the boundary costs without registration, per placement of the converts
O=gpurun_out/r04_08; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/gpu_register_stress.py 90 no_register staged register > $O/register_stress_modes.log 2>&1; cut -c1-600 $O/register_stress_modes.log
for pin in "" "--pinned"; do
  timeout 400 python tools/gpu_hostlevel.py headline config2 config3 --register $pin >> $O/host_register_ab.log 2>&1
done
grep '^{' $O/host_register_ab.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'reg', j['register_ms'], 'kvstage', j['kv_stage_ms'], 'cvt_threads', j['host_convert_threads'], 'widen', j['host_widen'])"
