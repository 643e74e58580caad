#!/bin/bash
# round 3, call c: P > 1 host schedule after the enqueue-pool fix -- tests, host-clock enqueue figures,
# kernel traces of config 4 on 2 loopback ranks without / with CUs reserved for the comm streams
# ($SDPA_COMM_CUS), and what a CU-masked compute stream costs the headline kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -s 2>&1 | grep -v Warning | tail -30 > $O/pytest_host.log
timeout 900 python tools/gpu_multirank_host.py config3 headline config4 > $O/multirank_host_enqueue.log 2>$O/multirank.err
for cus in 0 8; do
  mkdir -p $O/trace$cus
  (cd /tmp && SDPA_COMM_CUS=$cus SDPA_VIRTUAL_GPUS=2 timeout 600 rocprofv3 --kernel-trace -d $O/trace$cus -o cfg4 --output-format csv -- python $R/tools/gpu_hostlevel.py config4 > $O/trace_run_$cus.log 2>&1)
  python tools/summarize_overlap.py $O/trace$cus > $O/config4_2ranks_overlap_reserve$cus.txt 2>&1
  rm -rf $O/trace$cus
done
for it in 1 2; do
  for cus in 0 8 16; do
    timeout 300 python bench.py --reserve-cus $cus --no-cpu-baseline --no-boundary --steps 30 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline reserve_cus=$cus', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_avg'],4), round(j['roofline']['frac'],4))" >> $O/masked_stream_cost.log
  done
done
tail -8 $O/pytest_host.log; cut -c1-330 $O/multirank_host_enqueue.log; tail -3 $O/multirank.err; for cus in 0 8; do tail -14 $O/config4_2ranks_overlap_reserve$cus.txt; grep total_ms $O/trace_run_$cus.log | cut -c1-300; done; cat $O/masked_stream_cost.log; tail -3 $O/bench.err
