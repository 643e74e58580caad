#!/bin/bash
# round 5, call 1: (a) same-box A/B of the fp32 kernel: shipped (classic form's text = round 3's again) vs the round-3
# and round-4 libraries; (b) can a running kernel be told a copy has landed (persistent chunk-streaming design);
# (c) the GPU suite on the new sources; (d) the bench line with the `configs` record
O=gpurun_out/r05_01; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/gpu_lib_ab.py > $O/lib_ab.log 2> $O/lib_ab.err; echo "lib_ab rc=$?"; tail -6 $O/lib_ab.log
timeout 120 tools/probes/stream_flag_probe > $O/stream_flag_probe.log 2>&1; echo "probe rc=$?"; cat $O/stream_flag_probe.log | cut -c1-260
timeout 1200 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; rc=$?
echo "suite rc=$rc $(grep -aE ' passed| failed' $O/suite.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|assert\|Error" $O/suite.log | head -30 | cut -c1-300; fi
grep -c "amd_mem_obj" $O/suite.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -3 $O/bench_n1.err | cut -c1-300
python - <<'P'
import json
j=json.load(open('gpurun_out/r05_01/bench_n1.json'))
r=j['roofline']
print('headline', j['ms_per_step'], 'lat', j.get('latency_ms'), r['kernel'], r['kernel_ms_avg'], r['frac'], 'boundary', j['boundary'].get('ms'))
s=j['scaling_config3']; print('config3', {k:s.get(k) for k in ('ms_per_step','latency_ms','kernel','kernel_ms_avg','kernel_frac_of_peak','boundary_ms','error')})
for k,v in (j.get('configs') or {}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','kernel','kernel_ms_avg','frac','boundary_ms','parity_max_err','error')})
P
