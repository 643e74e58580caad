#!/bin/bash
# round 5, call 35: the library as rebuilt after the Makefile's dependency fix (same sources, new build stamp): smoke + the streamed tests
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -k "streamed" 2>&1 | tail -2
python -c "
import importlib
pkg = importlib.import_module('mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd')
print(pkg.load().sdpa_version())"
