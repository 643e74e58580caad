#!/bin/bash
# round 2, GPU call X: whole -m gpu suite; bench.py boundary section with progressive pinning on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02x
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | cut -c1-300 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for pp in 1 0 1 0; do
  SDPA_PROGRESSIVE_PIN=$pp timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d['boundary']
print('SDPA_PROGRESSIVE_PIN=$pp', {k: round(b[k], 3) for k in ('ms', 'head_ms', 'tail_ms', 'register_ms', 'pipeline_ms')}, 'caller arrays from sdpa_host_alloc:', round(b['pinned_caller_arrays']['ms'], 3))
" >> $O/boundary.log
done
cat $O/boundary.log
