#!/bin/bash
for t in 0 16 32 64 96 112; do echo -n "tune=$t "; SDPA_TUNE=$t python tools/gpu_bf16_bench.py 2>/dev/null | head -1; done
