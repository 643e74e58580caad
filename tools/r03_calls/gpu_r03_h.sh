#!/bin/bash
# round 3, call h: (1) same-box A/B of the shipped library against the ROUND-2 library (commit 10ba295 built into
# lib/variants/libsdpa_hip_r02.so); (2) where the host converts' ~100 GB/s come from (NUMA probe, no GPU work).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h
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
timeout 600 python tools/gpu_hostcvt_probe.py > $O/hostcvt_numa_probe.log 2>$O/probe.err
cat $O/shipped_vs_round2_library_ab.log; cut -c1-420 $O/hostcvt_numa_probe.log; tail -3 $O/probe.err $O/bench.err
