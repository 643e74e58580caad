#!/bin/bash
# round 3, call g: same-box A/B of the shipped library against the ROUND-2 library (commit 10ba295, built into
# lib/variants/libsdpa_hip_r02.so): did the round's kernel changes move the headline kernel, or is it the box?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
mkdir -p $O
cd $R
export TMPDIR=/tmp
for it in 1 2 3; do
  for lib in shipped r02; do
    if [ $lib = r02 ]; then export SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_r02.so; else unset SDPA_HIP_LIB; fi
    for w in headline config2; do
      timeout 300 python bench.py --workload $w --no-cpu-baseline --no-boundary --steps 40 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w $lib', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_avg'],4), round(j['roofline']['frac'],4))" >> $O/shipped_vs_round2_library_ab.log
    done
  done
done
unset SDPA_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -x -q -k "convert" 2>&1 | tail -4 > $O/pytest_convert.log
cat $O/shipped_vs_round2_library_ab.log; cat $O/pytest_convert.log; tail -3 $O/bench.err
