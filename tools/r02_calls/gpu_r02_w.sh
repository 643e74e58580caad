#!/bin/bash
# round 2, GPU call W: bench.py's boundary section (fresh result array per call) with and without
# progressive pinning
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02w
mkdir -p $O
cd $R
export TMPDIR=/tmp
for pp in 1 0 1 0; do
  SDPA_PROGRESSIVE_PIN=$pp timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d['boundary']
print('SDPA_PROGRESSIVE_PIN=$pp', {k: round(b[k], 3) for k in ('ms', 'head_ms', 'tail_ms', 'register_ms', 'pipeline_ms')}, 'pinned', round(b['pinned_caller_arrays']['ms'], 3))
" >> $O/boundary.log
done
cat $O/boundary.log
