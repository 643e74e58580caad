#!/bin/bash
# round 4, call 1: stream-K A/B at the device level, then the whole GPU suite with the abort tracer armed, then the bench line
O=gpurun_out/r04_01; mkdir -p $O
rocminfo 2>/dev/null | grep -m1 -i "gfx950" > $O/env.txt; /opt/rocm/bin/hipcc --version | grep -i "hip version" >> $O/env.txt
python -c "import importlib;p=importlib.import_module('mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd');print(p.load().sdpa_version().decode())" >> $O/env.txt 2>&1
timeout 600 python tools/gpu_streamk_ab.py > $O/streamk_ab.log 2> $O/streamk_ab.err; echo "rc=$?" >> $O/streamk_ab.log
tail -3 $O/streamk_ab.err
SDPA_ABORT_TRACE=1 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-900 $O/bench.json
