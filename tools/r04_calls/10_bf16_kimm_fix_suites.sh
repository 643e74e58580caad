#!/bin/bash
# round 4, call 10: call 9's immediate-offset K pieces were wrong (one bf16 shape at 0.049 > 0.01): the instruction offset of an
# LDS-DMA moves the LDS address too.  (a) bf16 parity with M0 compensated (shipped), and on the no-lead-wait-states build;
# (b) the whole suite twice, abort tracer armed; (c) timing base / kimm0 / nolead
O=gpurun_out/r04_10; mkdir -p $O
R=$GRAFT_REPO_ROOT
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16.log 2>&1; echo "(a) bf16 parity rc=$? $(grep -aE ' passed| failed' $O/pytest_bf16.log | tail -1 | cut -c1-100)"
grep -a "^FAILED" $O/pytest_bf16.log | head -5 | cut -c1-200
SDPA_HIP_LIB=$PKG/lib/variants/libsdpa_hip_nolead.so timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16_nolead.log 2>&1; echo "(a) nolead bf16 parity rc=$? $(grep -aE ' passed| failed' $O/pytest_bf16_nolead.log | tail -1 | cut -c1-100)"
grep -a "^FAILED" $O/pytest_bf16_nolead.log | head -5 | cut -c1-200
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|Error\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
for v in base kimm0 nolead base kimm0 nolead; do
  echo -n "$v: " >> $O/bf16_tandem_kimm_ab.log
  L=$PKG/lib/variants/libsdpa_hip_$v.so; [ $v = base ] && L=$PKG/lib/libsdpa_hip.so
  SDPA_HIP_LIB=$L timeout 200 python tools/gpu_bf16_bench.py 512 2>&1 | grep '^{' | head -1 >> $O/bf16_tandem_kimm_ab.log
done
cut -c1-150 $O/bf16_tandem_kimm_ab.log
