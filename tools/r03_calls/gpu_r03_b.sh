#!/bin/bash
# round 3, call b: the P > 1 host schedule (enqueue threads, comm streams, reduce-scatter egress) on loopback
# ranks -- tests, host-clock enqueue figures, a kernel trace of config 4 on 2 ranks -- and an A/B of the
# fp32 kernel's range check (straight-line second pass) against a build without it.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -s 2>&1 | grep -v Warning | tail -30 > $O/pytest_host.log
timeout 600 python tools/gpu_multirank_host.py config3 headline config4 > $O/multirank_host_enqueue.log 2>$O/multirank.err
mkdir -p $O/trace
(cd /tmp && SDPA_VIRTUAL_GPUS=2 timeout 600 rocprofv3 --kernel-trace -d $O/trace -o cfg4 --output-format csv -- python $R/tools/gpu_hostlevel.py config4 > $O/trace_run.log 2>&1)
python tools/summarize_overlap.py $O/trace > $O/config4_2ranks_overlap.txt 2>&1
rm -rf $O/trace
# A/B: headline + config2, shipped lib vs variant without the range check
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
for it in 1 2 3; do
  for lib in shipped noredo; do
    if [ $lib = noredo ]; then export SDPA_HIP_LIB=$R/$PKG/lib/variants/libsdpa_hip_noredo.so; else unset SDPA_HIP_LIB; fi
    for w in headline config2; do
      timeout 300 python bench.py --workload $w --no-cpu-baseline --no-boundary --steps 40 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w $lib', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_avg'],4), round(j['roofline']['frac'],4))" >> $O/range_check_cost_ab.log
    done
  done
done
unset SDPA_HIP_LIB
tail -12 $O/pytest_host.log; cat $O/multirank_host_enqueue.log | cut -c1-420; tail -3 $O/multirank.err; cat $O/config4_2ranks_overlap.txt | tail -16; cat $O/range_check_cost_ab.log; tail -3 $O/bench.err
