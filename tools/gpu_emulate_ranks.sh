#!/bin/bash
# Per-rank GPU-side step time of the K/V-sharded job at N ranks (ONE rank's share emulated on one
# GPU, the RCCL calls on a one-rank communicator), for several Q batch sizes, at the metric shape
# and at configs[2]'s shape (n=262144: n_local = 32768 at N=8).  A tuning aid, not a scaling result.
for W in headline config3; do for N in 2 4 8; do for B in 32768 16384 8192; do
  echo -n "$W N=$N B=$B "
  SDPA_BENCH_FORCE_DIST=1 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --emulate-ranks $N --q-batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f kernel_ms_avg %.3f launches/step %d kernel TF %.1f parity_err %.1e' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['launches']//d['steps'], d['roofline']['achieved'], d['parity_max_err']))"
done; done; done
