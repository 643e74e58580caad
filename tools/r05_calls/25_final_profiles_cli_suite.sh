#!/bin/bash
# round 5, call 25: on the round's final sources -- rocprofv3 profile + PMC passes of the headline bench command and of config 5 in bf16;
# the one-shot CLI cold (config 5 in bf16 now streams); the whole GPU suite, smoke, and the bench line as the driver runs it
O=gpurun_out/r05_25; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/gpu_profile.sh r05_final_headline 2>&1 | tail -12 | cut -c1-200
BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r05_final_config5_bf16 2>&1 | tail -12 | cut -c1-200
timeout 600 bash tools/gpu_cli_cold.sh 6 > $O/cli_cold.log 2>&1; grep -c total_us $O/cli_cold.log; grep "config5 cold" $O/cli_cold.log | cut -c1-250
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$? $(grep -aE ' passed| failed' $O/suite.log | tail -1 | cut -c1-120)"
grep -an "^FAILED\|^ERROR\|Memory access fault\|SIGABRT\|Fatal" $O/suite.log | head -10 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
/usr/bin/time -v timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; grep "Elapsed (wall" $O/bench_n1.err
python - <<'P'
import json
j=json.load(open('gpurun_out/r05_25/bench_n1.json'))
r=j['roofline']
print('headline', j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_avg'], round(r['frac'],4), 'traffic', r['traffic'], r.get('mfma_util'), 'boundary', {k:j['boundary'].get(k) for k in ('ms','head_ms','tail_ms','fused_kernel_ms','streamed')})
for k,v in (j.get('configs') or {}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','kernel_ms_avg','frac','boundary_ms','parity_max_err','error')}, (v.get('boundary') or {}).get('streamed'))
print('cpu_baseline', j['cpu_baseline'])
P
