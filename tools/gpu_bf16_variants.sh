#!/bin/bash
# A/B of bf16 d=512 kernel variants: each lib under PKG/lib/variants/ is copied over the
# product lib in this scratch copy, then config 5 is benched.  The stock lib runs first and last.
set -u
cd "$(dirname "$0")/.."
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
OUT=gpurun_out/bf16_variants; mkdir -p $OUT
cp $PKG/lib/libsdpa_hip.so /tmp/stock.so
run() { # name
  timeout 300 python bench.py --workload config5 --precision bf16 --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline 2>$OUT/$1.err | tail -1 > $OUT/$1.json
  python - "$1" "$OUT/$1.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'kernel_TF', d.get('roofline',{}).get('achieved'))
except Exception as e: print(sys.argv[1],'FAILED',e)
PY
}
run stock
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -2
for tune in 1 2 8 9 11; do
  SDPA_TUNE=$((tune*256)) run "stock_abl$tune"
done
for so in $PKG/lib/variants/*.so; do
  cp $so $PKG/lib/libsdpa_hip.so
  run "$(basename $so .so)"
done
cp /tmp/stock.so $PKG/lib/libsdpa_hip.so
run stock_again
