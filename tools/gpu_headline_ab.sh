#!/bin/bash
# A/B of the headline fp32 kernel: the built lib vs lib/variants/*.so, interleaved, kernel ms from bench.py.
cd "$(dirname "$0")/.."
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
cp $PKG/lib/libsdpa_hip.so /tmp/new.so
run() { python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['achieved'],2), round(d['ms_per_step'],4))"; }
for it in 1 2 3; do
  cp /tmp/new.so $PKG/lib/libsdpa_hip.so; run new
  for so in $PKG/lib/variants/*.so; do cp $so $PKG/lib/libsdpa_hip.so; run "$(basename $so .so)"; done
done
cp /tmp/new.so $PKG/lib/libsdpa_hip.so
