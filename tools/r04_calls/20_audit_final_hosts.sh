#!/bin/bash
# round 4, call 20: (a) the bounds-audit build of the FINAL kernel sources (the immediate-offset K pieces of the tandem kernel are
# new LDS-DMA sites) over the parity, bf16, BASELINE-config and fuzz tests; (b) the host-side tests on the final host (reservation
# only where it pays)
O=gpurun_out/r04_20; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
SDPA_HIP_LIB=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants/libsdpa_hip_audit.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -q > $O/audit_build_suite.log 2>&1; echo "audit rc=$?"; grep -a "DMA bounds audit\|passed\|failed" $O/audit_build_suite.log | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_hosts_agree.py tests/test_gpu_register_optin.py tests/test_gpu_bench_multirank.py -q > $O/pytest_host.log 2>&1; echo "host rc=$? $(grep -aE ' passed| failed' $O/pytest_host.log | tail -1 | cut -c1-100)"
grep -a "^FAILED" $O/pytest_host.log | head -5 | cut -c1-200
