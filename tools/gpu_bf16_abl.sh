#!/bin/bash
for t in ${TUNES:-0 256 512 2048 2304 2816}; do echo -n "tune=$t "; SDPA_TUNE=$t python tools/gpu_bf16_bench.py 2>/dev/null | head -1; done
