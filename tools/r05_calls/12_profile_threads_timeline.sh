#!/bin/bash
# round 5, call 12: (a) rocprofv3 kernel-trace stats + PMC passes of the headline bench command on the round's kernel sources
# (tools/gpu_profile.sh -> profiles/r05/headline_f32_*, profiles/traffic_latest.json); (b) converter-pool thread count sweep at the
# boundary for config 5 in bf16 (feed bound) and config 2; (c) kernel + copy timeline of config 5's bf16 boundary call
O=gpurun_out/r05_12; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 bash tools/gpu_profile.sh r05_headline 2>&1 | tail -16
timeout 300 python tools/gpu_hostlevel.py config5:bf16 config2 --hostcvt > $O/threads.log 2> $O/threads.err
python - <<'P'
import json
for l in open('gpurun_out/r05_12/threads.log'):
    j = json.loads(l); print(j['shape'], j['knobs'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'launches', j['fused_launches'], 'threads', j['host_convert_threads'])
P
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_c5 -o t -- python $R/tools/gpu_hostlevel.py config5:bf16 > $R/$O/trace_c5.log 2>&1)
python tools/summarize_timeline.py $O/trace_c5 > $O/timeline_config5_bf16.txt 2>&1
grep total_ms $O/trace_c5.log | tail -1 | cut -c1-400; tail -3 $O/timeline_config5_bf16.txt
rm -rf $O/trace_c5
