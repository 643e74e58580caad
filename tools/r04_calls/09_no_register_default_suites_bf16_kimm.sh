#!/bin/bash
# round 4, call 9: registration of caller arrays is now opt-in (call 8: 0 faults without it, a fault within 8 s with it).
# (a) the reproducer on the new default (with and without page-locked caller arrays), the registered mode as the control;
# (b) the whole GPU suite twice, abort tracer armed; (c) bf16 tandem: immediate-offset K pieces (base) against
# per-piece scalar address arithmetic (kimm0), the round-3 kernel (r03) and base without the lead wait states (nolead);
# (d) bf16 parity on the nolead build (are the lead wait states needed anywhere?); (e) the bench line
O=gpurun_out/r04_09; mkdir -p $O
R=$GRAFT_REPO_ROOT
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
export TMPDIR=/tmp
timeout 600 python tools/gpu_register_stress.py 60 default default_pinned_callers > $O/register_stress_default.log 2>&1
timeout 200 python tools/gpu_register_stress.py 25 register >> $O/register_stress_default.log 2>&1; cut -c1-500 $O/register_stress_default.log
export AMD_LOG_LEVEL=1 SDPA_ABORT_TRACE=1
for i in 1 2; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/suite_run_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc $(grep -aE ' passed| failed' $O/suite_run_$i.log | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|Error\|File \".*tests\|assert" $O/suite_run_$i.log | head -30 | cut -c1-300; fi
done
unset AMD_LOG_LEVEL SDPA_ABORT_TRACE
for v in base kimm0 r03 nolead base kimm0 r03 nolead; do
  echo -n "$v: " >> $O/bf16_tandem_kimm_ab.log
  SDPA_HIP_LIB=$PKG/lib/variants/libsdpa_hip_$v.so timeout 200 python tools/gpu_bf16_bench.py 512 2>&1 | grep '^{' | head -1 >> $O/bf16_tandem_kimm_ab.log
done
cut -c1-150 $O/bf16_tandem_kimm_ab.log
SDPA_HIP_LIB=$PKG/lib/variants/libsdpa_hip_nolead.so timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16_nolead.log 2>&1; echo "(d) nolead bf16 parity rc=$? $(grep -aE ' passed| failed' $O/pytest_bf16_nolead.log | tail -1 | cut -c1-100)"
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('$O/bench_n1.json'))
print(j['ms_per_step'], j['roofline']['frac'], json.dumps(j['boundary'])[:600])"
