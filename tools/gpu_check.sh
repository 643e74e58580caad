#!/bin/bash
# GPU-box check: parity tests, then the bench line.  Usage: bash tools/gpu_check.sh [tag] [pytest-args]
TAG=${1:-run}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS} > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
tail -3 gpurun_out/bench_$TAG.log | cut -c1-1500
