#!/bin/bash
# round 4, call 15: chunks staged two ahead of the launch being enqueued (host converts): parity of the host pipeline, then the
# boundary of every BASELINE shape on the default path, then the timeline again
O=gpurun_out/r04_15; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_baseline_configs.py tests/test_gpu_hosts_agree.py tests/test_gpu_register_optin.py -q > $O/pytest_host.log 2>&1; echo "pytest rc=$? $(grep -aE ' passed| failed' $O/pytest_host.log | tail -1 | cut -c1-100)"
grep -a "^FAILED" $O/pytest_host.log | head -5 | cut -c1-200
timeout 600 python tools/gpu_hostlevel.py headline config2 config3 config4 config5:bf16 config5 > $O/hostlevel_default.log 2>&1
timeout 600 python tools/gpu_hostlevel.py headline config2 config3 config4 config5:bf16 --pinned > $O/hostlevel_pinned.log 2>&1
cat $O/hostlevel_default.log $O/hostlevel_pinned.log | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kvstage', j['kv_stage_ms'], 'kernel', j['kernel_ms'], 'cvt', j['host_convert_threads'], 'widen', j['host_widen'])"
for shape in headline config3; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_$shape -o t -- python $R/tools/gpu_hostlevel.py $shape > $R/$O/trace_$shape.log 2>&1)
  python tools/summarize_timeline.py $O/trace_$shape > $O/timeline_$shape.txt 2>&1
  grep "compute idle" $O/timeline_$shape.txt | sort -t'e' -k3 | tail -3 | cut -c1-160; tail -1 $O/timeline_$shape.txt
  rm -rf $O/trace_$shape
done
