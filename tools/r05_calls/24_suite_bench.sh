#!/bin/bash
# round 5, call 24: the whole GPU suite on the bf16-streamed default, then the bench line (no CPU baseline: 60 s of mpiexec are not GPU work)
O=gpurun_out/r05_24; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; rc=$?
echo "suite rc=$rc $(grep -aE ' passed| failed' $O/suite.log | tail -1 | cut -c1-120)"
grep -an "^FAILED\|^ERROR\|Memory access fault\|SIGABRT\|Fatal" $O/suite.log | head -20 | cut -c1-300
grep -ac "amd_mem_obj" $O/suite.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -2 $O/bench_n1.err | cut -c1-300
python - <<'P'
import json
j=json.load(open('gpurun_out/r05_24/bench_n1.json'))
r=j['roofline']
print('headline', j['ms_per_step'], 'lat', j.get('latency_ms'), r['kernel'], r['kernel_ms_avg'], round(r['frac'],4), 'traffic', r['traffic'], 'boundary', {k:j['boundary'].get(k) for k in ('ms','head_ms','tail_ms','fused_kernel_ms','fused_launches','streamed')})
s=j['scaling_config3']; print('config3', {k:s.get(k) for k in ('ms_per_step','kernel_ms_avg','kernel_frac_of_peak','boundary_ms','error')})
for k,v in (j.get('configs') or {}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','kernel_ms_avg','frac','boundary_ms','parity_max_err','error')}, (v.get('boundary') or {}).get('streamed'))
P
