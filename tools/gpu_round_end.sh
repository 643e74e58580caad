#!/bin/bash
# End-of-round confirmation on one box: full GPU suite, smoke, the two bench lines, config-5 profile.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/round_end; mkdir -p $OUT
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 2>$OUT/bench_headline.err | tail -1 > $OUT/bench_headline.json; cat $OUT/bench_headline.json | cut -c1-600
timeout 600 python bench.py --workload config5 --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>$OUT/bench_config5_bf16.err | tail -1 > $OUT/bench_config5_bf16.json; cat $OUT/bench_config5_bf16.json | cut -c1-600
[ -n "${SKIP_PROFILE:-}" ] || BENCH_ARGS="--workload config5 --precision bf16" timeout 900 bash tools/gpu_profile.sh r01_config5_bf16 2>&1 | tail -40
