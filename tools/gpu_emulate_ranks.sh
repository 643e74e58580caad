#!/bin/bash
# Per-rank GPU-side step time of the K/V-sharded job at N ranks (one rank emulated on one GPU, RCCL
# calls on a one-rank communicator), for several Q batch sizes.
for N in 2 4 8; do for B in 32768 16384 8192; do
  echo -n "N=$N B=$B "
  SDPA_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-ranks $N --q-batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f kernel_ms_avg %.3f launches/step %d kernel TF %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['launches']//d['steps'], d['roofline']['achieved']))"
done; done
