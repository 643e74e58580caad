#!/bin/bash
# round 2, GPU call O: final build -- whole -m gpu suite, smoke, bench (default and with the RCCL
# choreography forced on a one-rank communicator), fp32 head dims incl. dk = 384
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-200 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $O/pytest_gpu.log
timeout 300 python tools/gpu_f32_dims.py 128 256 384 512 2>&1 | grep '"d"' > $O/f32_dims.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
SDPA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_forced_dist_gather.json 2>> $O/bench_n1.err
SDPA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --merge allreduce > $O/bench_forced_dist_allreduce.json 2>> $O/bench_n1.err
cat $O/pytest_gpu.log $O/f32_dims.log; cut -c1-600 $O/bench_n1.json; echo; cut -c1-900 $O/bench_forced_dist_gather.json | cut -c600-900; tail -3 $O/bench_n1.err
