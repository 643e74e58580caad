#!/bin/bash
# one rocprofv3 kernel trace of tools/gpu_short_kernel_probe.py; the per-mode kernel durations and
# gaps are cut out of the trace by tools/summarize_short_probe.py
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/short_probe; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/tools/gpu_short_kernel_probe.py 8192x8192 32768x8192 > $OUT/log.txt 2>&1
echo "rc=$?" >> $OUT/log.txt
grep -E "^\{|rc=" $OUT/log.txt
python $R/tools/summarize_short_probe.py $OUT | tee $OUT/summary.txt
# keep the merge-back small
find $OUT -name "*kernel_trace.csv" -size +20M -delete
