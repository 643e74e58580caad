#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu > gpurun_out/pytest_bf16.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_bf16.log
tail -25 gpurun_out/pytest_bf16.log
