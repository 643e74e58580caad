#!/bin/bash
# round 4, call 18: the round's final sources: whole GPU suite (abort tracer armed), smoke, the two bench lines
O=gpurun_out/r04_18; mkdir -p $O
export TMPDIR=/tmp
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $O/suite_run_1.log 2>&1; rc=$?
echo "suite run 1 rc=$rc $(grep -aE ' passed| failed' $O/suite_run_1.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|File \".*tests\|assert" $O/suite_run_1.log | head -30 | cut -c1-300; fi
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2> $O/bench_config5_bf16.err; echo "bench bf16 rc=$?"
python -c "
import json
for f in ('bench_n1', 'bench_config5_bf16'):
    j=json.load(open('$O/%s.json' % f))
    print(f, j['ms_per_step'], j['roofline']['frac'], 'boundary', j['boundary']['ms'], j['boundary'].get('pinned_caller_arrays', {}).get('ms'), 'cpu', json.dumps(j.get('cpu_baseline'))[:300])"
