#!/bin/bash
# round 4, call 11: (a) the whole GPU suite, abort tracer armed (call 10's one failure was a stale expectation of the
# hosts-agree test: 248 compute units for a one-batch call); (b) rocprofv3 profile + PMC passes of the headline bench
# command and of config 5 in bf16 on the round's final kernels; (c) the bench lines themselves
O=gpurun_out/r04_11; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $O/suite_run_1.log 2>&1; rc=$?
echo "suite run 1 rc=$rc $(grep -aE ' passed| failed' $O/suite_run_1.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|File \".*tests\|assert" $O/suite_run_1.log | head -30 | cut -c1-300; fi
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
timeout 900 bash tools/gpu_profile.sh r04_headline 2>&1 | tail -18
BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r04_config5_bf16 2>&1 | tail -18
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --workload config5 --precision bf16 --no-cpu-baseline > $O/bench_config5_bf16.json 2> $O/bench_config5_bf16.err; echo "bench bf16 rc=$?"
python -c "
import json
for f in ('bench_n1', 'bench_config5_bf16'):
    j=json.load(open('$O/%s.json' % f))
    print(f, j['ms_per_step'], j['roofline']['frac'], json.dumps(j['boundary'])[:700])"
du -sh gpurun_out
