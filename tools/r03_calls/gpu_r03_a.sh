#!/bin/bash
# round 3, call a: first look at the in-kernel split merge, the fp32 range redo and bench.py's
# self-launched dry-run worlds.  Usage: gpurun -- 'bash tools/r03_calls/gpu_r03_a.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40 > $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q 2>&1 | tail -40 > $O/pytest_multirank.log
for form in kernel pass; do
  if [ $form = pass ]; then export SDPA_SPLIT_MERGE=pass; else unset SDPA_SPLIT_MERGE; fi
  for rep in 1 2; do
    timeout 300 python bench.py --workload config2 --no-cpu-baseline --no-boundary --steps 50 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('config2 $form', j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'])" >> $O/ab.log
    timeout 300 python bench.py --emulate-ranks 8 --no-cpu-baseline --no-boundary --steps 50 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('share8 $form', j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'])" >> $O/ab.log
  done
  timeout 300 python bench.py --no-cpu-baseline --no-boundary 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline $form', j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'])" >> $O/ab.log
done
unset SDPA_SPLIT_MERGE
timeout 600 python tools/gpu_kernel_grid.py > $O/kernel_grid_f32.log 2>&1
cat $O/pytest_parity.log | tail -15; cat $O/pytest_multirank.log | tail -15; cat $O/ab.log; tail -5 $O/bench.err; grep -E "32768.*8192|8192.*8192" $O/kernel_grid_f32.log | cut -c1-200
