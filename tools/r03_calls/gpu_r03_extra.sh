#!/bin/bash
# round 3, extras: the bench's N = 8 code path as a dry run (8 ranks share one GPU over gloo: launch logic, 8 seeded
# shards of 8192 keys, merge + reduce choreography, parity over the re-drawn shards -- NOT a result), one-shot CLI
# runs (the reference's own usage: fresh process, file in, verdict out), boundary with page-locked caller arrays.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03extra
mkdir -p $O
cd $R
export TMPDIR=/tmp
for n in 8 4 2; do
  SDPA_BENCH_BACKEND=gloo SDPA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $n --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline > $O/bench_dry_run_world$n.json 2>> $O/dry.err
done
timeout 900 bash tools/gpu_cli_timing.sh > $O/cli_one_shot_timing.log 2>&1
timeout 600 python tools/gpu_hostlevel.py headline config2 config4 config5:bf16 --pinned > $O/hostlevel_pinned.log 2>> $O/dry.err
for n in 8 4 2; do python -c "import json; j=json.load(open('$O/bench_dry_run_world$n.json')); print(j['metric'][:60], j['n_gpus'], j['rccl'], j['parity_max_err'], j['parity_tol'], j['config']['kv_rows_per_gpu'], j['config']['parallelism'])"; done
cat $O/cli_one_shot_timing.log | cut -c1-260 | head -30; cut -c1-260 $O/hostlevel_pinned.log; tail -3 $O/dry.err
