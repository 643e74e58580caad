#!/bin/bash
# A/B the $SDPA_TUNE switches on the headline kernel (interleaved rounds, one process each).
mkdir -p gpurun_out
: > gpurun_out/tune.log
for round in 1 2; do
for t in ${TUNES:-0 1 2 3}; do
  echo -n "tune=$t round=$round " >> gpurun_out/tune.log
  SDPA_TUNE=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['achieved'], d['roofline']['kernel_ms_avg'], d['ms_per_step'])" >> gpurun_out/tune.log
done; done
cat gpurun_out/tune.log
