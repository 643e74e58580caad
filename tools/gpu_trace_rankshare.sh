#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/trace_rankshare; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SDPA_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-ranks 8 > $OUT/log.txt 2>&1
python - <<'PY'
import csv,glob,os
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/trace_rankshare'
for p in glob.glob(out+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        print("%-80s calls=%5s avg_us=%8.1f total_ms=%8.3f" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
