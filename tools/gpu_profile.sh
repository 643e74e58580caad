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
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-boundary ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
BENCH2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_lds -o bench -- $BENCH2 > $OUT/pmc_lds.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
# keep the merge-back small: drop the big per-dispatch traces except the stats
du -sh $OUT >> $OUT/files.txt
python $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
