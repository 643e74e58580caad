#!/bin/bash
# round 3, call i: which source structure of the fp32 kernel's second pass keeps the round-2 rate?  Same box,
# interleaved: the shipped build (walk as a lambda called twice), vd (+ in-kernel merge templated out of the
# default instantiation), vb (walk duplicated textually by a macro), vc (vb without the second pass), r02.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
mkdir -p $O
cd $R
export TMPDIR=/tmp
for it in 1 2 3; do
  for lib in shipped vd vb vc r02; do
    if [ $lib = shipped ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_$lib.so; fi
    for w in headline config2 d256; do
      timeout 300 python bench.py --workload $w --no-cpu-baseline --no-boundary --steps 30 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w $lib', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_avg'],4), round(j['roofline']['frac'],4))" >> $O/second_pass_structure_ab.log
    done
  done
done
unset SDPA_HIP_LIB
sort -k1,2 -s $O/second_pass_structure_ab.log; tail -3 $O/bench.err
