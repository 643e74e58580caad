#!/bin/bash
# round 4, call 2: reservation by grid size vs CU mask; do collectives overlap the next batch's fused kernel now;
# the new bench line at N = 1 and as a dry-run world of 2 (scaling_config3 / phases / c_host records)
O=gpurun_out/r04_02; mkdir -p $O
timeout 600 python tools/gpu_streamk_ab.py quick > $O/streamk_reserve_ab.log 2> $O/streamk_reserve_ab.err; echo "rc=$?" >> $O/streamk_reserve_ab.log
grep -c shape $O/streamk_reserve_ab.log; tail -2 $O/streamk_reserve_ab.err
# overlap: config 4 (4 Q batches) through the C host, ONE rank with its merge collectives forced, compute stream with
# 0 / 8 CUs' worth of slots reserved (grid size) and 8 reserved by CU mask
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "0 0" "8 0" "8 1"; do
  set -- $mode
  tag=reserve$1_mask$2
  SDPA_FORCE_COLLECTIVES=1 SDPA_COMM_CUS=$1 SDPA_RESERVE_BY_MASK=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$tag -o t -- python $R/tools/gpu_hostlevel.py config4 > $R/$O/trace_$tag.log 2>&1
  python $R/tools/summarize_overlap.py $R/$O/trace_$tag > $R/$O/config4_forced_collectives_overlap_$tag.txt 2>&1
  tail -1 $R/$O/config4_forced_collectives_overlap_$tag.txt; grep total_ms $R/$O/trace_$tag.log | tail -1 | cut -c1-300
  rm -rf $R/$O/trace_$tag
done
cd $R
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench n1 rc=$?"
python -c "
import json; j=json.load(open('$O/bench_n1.json'))
print({k: j[k] for k in ('value','ms_per_step','gpu_busy_extra','scaling_config3')}, j['roofline']['frac'])"
SDPA_BENCH_BACKEND=gloo SDPA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dry2.json 2> $O/bench_dry2.err; echo "bench dry2 rc=$?"
python -c "
import json; j=json.load(open('$O/bench_dry2.json'))
print({k: j[k] for k in ('metric','ms_per_step','phases','c_host')}); print(j['scaling_config3'])"
tail -5 $O/bench_dry2.err
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_host_pipeline.py -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_subset.log
