#!/bin/bash
# round 2, last call: the full GPU suite (with tests/test_gpu_torch_collectives.py) and the bench's
# N > 1 code path forced onto a one-rank RCCL communicator (all_gather_into_tensor merge)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02last; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 150 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|^E " | cut -c1-300 | tee $O/pytest_gpu.log
SDPA_BENCH_FORCE_DIST=1 timeout 60 python bench.py --no-cpu-baseline --merge gather > $O/bench_forced_dist_gather.json 2> $O/err.log
cut -c1-900 $O/bench_forced_dist_gather.json; tail -3 $O/err.log | cut -c1-300
