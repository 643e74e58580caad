#!/bin/bash
# rocprofv3 runs for the round's profiles/: kernel-trace stats of the bench command, then
# PMC passes (each in its own run, never combined with tracing domains).
# usage: [BENCH_ARGS="--workload config5 --precision bf16"] bash tools/gpu_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-boundary: every fused launch in the process is then a timed-step launch, so rocprofv3's
# per-kernel AVERAGE is the full-launch duration bench.py reports
# (bench.py runs its clock pre-warm -- ~60 ms of untimed steps -- before the W warmup steps, so the K timed
#  launches are the LAST K dispatches of the fused kernel in the trace and sit on the clock's plateau:
#  summarize_prof.py quotes their average next to the all-dispatch average)
STEPS=${PROF_STEPS:-10}
# (--no-scaling-record --min-gpu-seconds 0: nothing but the W + K steps launches the fused kernel)
BENCH="python $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-boundary --no-scaling-record --no-configs --min-gpu-seconds 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
BENCH2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --no-scaling-record --no-configs --min-gpu-seconds 0 ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_lds -o bench -- $BENCH2 > $OUT/pmc_lds.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
# keep the merge-back small: drop the big per-dispatch traces except the stats
du -sh $OUT >> $OUT/files.txt
PROF_STEPS=$STEPS PROF_PMC_STEPS=3 BENCH_ARGS="${BENCH_ARGS:-}" python $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A14 "== dominant kernel" $OUT/summary.txt
