#!/bin/bash
# round 5, call 33: the one test call 32 failed (its expectation, not the library), after the fix; with the pool's trace to see the node
timeout 300 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -k "numa" 2>&1 | tail -3
SDPA_HOST_CVT_PIN=1 SDPA_HOST_CVT_TRACE=1 timeout 200 python tools/gpu_hostlevel.py config2 2>&1 | grep -a "hostcvt trace" | tail -2 | cut -c1-400
