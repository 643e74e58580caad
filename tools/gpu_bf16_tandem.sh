#!/bin/bash
# bf16 dv > 256: the tandem kernel against the wide kernel -- bitwise tests, then interleaved kernel timing at
# BASELINE config 5 (bench.py brackets the launch with HIP events).   gpurun -- 'bash tools/gpu_bf16_tandem.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tandem
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "tandem" 2>&1 | tail -8 > $O/pytest_tandem.log
for it in 1 2 3; do
  for k in 0 1; do
    SDPA_BF16_TANDEM=$k timeout 300 python bench.py --workload config5 --precision bf16 --no-cpu-baseline --no-boundary --steps 30 2>>$O/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('config5 bf16 tandem=$k', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_avg'],4), round(j['roofline']['frac'],4), j['parity_max_err'])" >> $O/tandem_vs_wide_ab.log
  done
done
cat $O/pytest_tandem.log; cat $O/tandem_vs_wide_ab.log; tail -3 $O/bench.err
