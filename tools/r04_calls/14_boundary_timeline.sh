#!/bin/bash
# round 4, call 14: where does the boundary call lose the time its kernels do not account for?  kernel + memory-copy trace of
# tools/gpu_hostlevel.py (6 calls per shape; the last one is read) on the default path (pageable arrays, staged) and from
# page-locked caller arrays; config 4's boundary on the new default
O=gpurun_out/r04_14; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for shape in headline config3; do
  for pin in "" "--pinned"; do
    tag=${shape}${pin:+_pinned}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_$tag -o t -- python $R/tools/gpu_hostlevel.py $shape $pin > $R/$O/trace_$tag.log 2>&1)
    python tools/summarize_timeline.py $O/trace_$tag > $O/timeline_$tag.txt 2>&1
    grep total_ms $O/trace_$tag.log | tail -1 | cut -c1-330; tail -1 $O/timeline_$tag.txt
    rm -rf $O/trace_$tag
  done
done
timeout 300 python tools/gpu_hostlevel.py config4 config5:bf16 2>&1 | grep '^{' | cut -c1-400
