#!/bin/bash
# round 5, call 32: the round's final sources -- PMC / kernel-trace profile of config 5 in bf16 (its traffic stamp), the whole GPU suite twice, smoke,
# the bench line as the driver runs it (with the CPU baseline) and config 5's own bench line
O=gpurun_out/r05_32; mkdir -p $O
export TMPDIR=/tmp
BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r05_final2_config5_bf16 2>&1 | tail -11 | cut -c1-200
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu > $O/suite1.log 2>&1; echo "suite 1 rc=$? $(grep -aE ' passed| failed' $O/suite1.log | tail -1 | cut -c1-120)"
grep -an "^FAILED\|^ERROR\|Memory access fault\|SIGABRT\|Fatal" $O/suite1.log | head -10 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
S=$(date +%s); timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"
timeout 600 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2> $O/bench_config5_bf16.err; echo "bench bf16 rc=$?"
python - <<'P'
import json
j=json.load(open('gpurun_out/r05_32/bench_n1.json'))
r=j['roofline']
print('headline', j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_avg'], round(r['frac'],4), 'traffic', r['traffic'], r.get('mfma_util'), 'boundary', {k:j['boundary'].get(k) for k in ('ms','head_ms','tail_ms','fused_kernel_ms','streamed')})
for k,v in (j.get('configs') or {}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','kernel_ms_avg','frac','boundary_ms','parity_max_err','error')}, (v.get('boundary') or {}).get('streamed'))
print('cpu_baseline', json.dumps(j['cpu_baseline'])[:300])
b=json.load(open('gpurun_out/r05_32/bench_config5_bf16.json'))
print('config5 bf16 line', b['ms_per_step'], b['roofline']['kernel'], b['roofline']['kernel_ms_avg'], round(b['roofline']['frac'],4), 'traffic', b['roofline']['traffic'], 'boundary', (b.get('boundary') or {}).get('ms'))
P
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu > $O/suite2.log 2>&1; echo "suite 2 rc=$? $(grep -aE ' passed| failed' $O/suite2.log | tail -1 | cut -c1-120)"
grep -an "^FAILED\|^ERROR\|Memory access fault\|SIGABRT\|Fatal" $O/suite2.log | head -10 | cut -c1-300
