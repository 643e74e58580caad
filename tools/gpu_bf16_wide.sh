#!/bin/bash
# bf16 wide-kernel bring-up: parity tests first, then config 5 timing with ablations.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/bf16_wide; mkdir -p $OUT
[ -n "${SKIP_TESTS:-}" ] || timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -15
run() { # name, env...
  timeout 300 python bench.py --workload config5 --precision bf16 --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline 2>$OUT/$1.err | tail -1 > $OUT/$1.json
  python - "$1" "$OUT/$1.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'kernel_TF', d.get('roofline',{}).get('achieved'))
except Exception as e: print(sys.argv[1],'FAILED',e)
PY
}
run stock
for tune in ${TUNES:-1 2 8 9 11}; do
  SDPA_TUNE=$((tune*256)) run "abl$tune"
done
