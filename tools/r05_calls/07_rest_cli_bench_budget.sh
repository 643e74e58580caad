#!/bin/bash
# round 5, call 7: the tests call 6 did not reach (it stopped at the register opt-in test, whose bitwise twin of the
# device-convert variant is the chunked schedule since the streamed default), config-2 A/B, the one-shot CLI cold, the bench
# line, then the bf16 per-step cycle budget (tools/gpu_bf16_budget.sh)
O=gpurun_out/r05_07; mkdir -p $O
export TMPDIR=/tmp
SDPA_STREAM_TIMEOUT_MS=1500 timeout 900 python -m pytest tests/test_gpu_register_optin.py tests/test_gpu_torch_collectives.py tests/test_gpu_host_pipeline.py -m gpu -q > $O/rest.log 2>&1; rc=$?
echo "rest rc=$rc $(grep -aE ' passed| failed' $O/rest.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|assert\|Error\|sdpa:" $O/rest.log | head -30 | cut -c1-400; fi
for sh in config2 headline; do timeout 200 python tools/gpu_hostlevel.py $sh --streamed >> $O/streamed_ab.log 2>> $O/streamed_ab.err; done
timeout 200 python tools/gpu_hostlevel.py config2 --streamed --pinned >> $O/streamed_ab.log 2>> $O/streamed_ab.err
timeout 200 python tools/gpu_hostlevel.py config5:bf16 >> $O/streamed_ab.log 2>> $O/streamed_ab.err
python - <<'P'
import json
for l in open('gpurun_out/r05_07/streamed_ab.log'):
    j=json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'launches', j['fused_launches'], 'streamed', j['streamed'], 'chunks', j['kv_chunks'])
P
timeout 900 bash tools/gpu_cli_cold.sh 8 > $O/cli_cold.log 2>&1; grep -c total_us $O/cli_cold.log; cut -c1-230 $O/cli_cold.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -3 $O/bench_n1.err | cut -c1-300
python - <<'P'
import json
j=json.load(open('gpurun_out/r05_07/bench_n1.json'))
r=j['roofline']
print('headline', j['ms_per_step'], 'lat', j.get('latency_ms'), r['kernel'], r['kernel_ms_avg'], r['frac'], 'boundary', {k:j['boundary'].get(k) for k in ('ms','head_ms','tail_ms','fused_kernel_ms','fused_launches','streamed')}, 'pinned', j['boundary']['pinned_caller_arrays']['ms'])
s=j['scaling_config3']; print('config3', {k:s.get(k) for k in ('ms_per_step','kernel_ms_avg','kernel_frac_of_peak','boundary_ms','error')})
for k,v in (j.get('configs') or {}).items(): print(k, {x:v.get(x) for x in ('ms_per_step','kernel_ms_avg','frac','boundary_ms','parity_max_err','error')}, (v.get('boundary') or {}).get('streamed'))
P
timeout 1500 bash tools/gpu_bf16_budget.sh > $O/bf16_budget.log 2>&1; cp gpurun_out/bf16_budget/budget.log $O/bf16_budget_timing.log; tail -30 $O/bf16_budget.log | cut -c1-400
