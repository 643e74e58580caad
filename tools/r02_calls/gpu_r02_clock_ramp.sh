#!/bin/bash
# round 2, last call: (1) A/B of bench.py's clock pre-warm (--prewarm-ms 0 vs the default) on the short
# steps -- config 2 and one rank's 1/8 and 1/4 share of the metric shape -- and on the metric shape
# itself; (2) bf16 head-dim series warmed by time; (3) the default bench line and the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02ramp; mkdir -p $O; cd $R; export TMPDIR=/tmp
line() { python - "$@" <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-34s prewarm_steps %3s  ms_per_step %.4f  kernel_ms_avg %.4f  kernel %.1f TFLOP/s (%.1f %%)  parity %.1e"
              % (tag, d.get("clock_prewarm_steps"), d["ms_per_step"], d["roofline"]["kernel_ms_avg"],
                 d["roofline"]["achieved"], 100 * d["roofline"]["frac"], d["parity_max_err"]))
PY
}
for pw in 0 60; do
  timeout 120 python bench.py --workload config2 --no-cpu-baseline --no-boundary --prewarm-ms $pw > $O/config2_pw$pw.json 2>$O/err.log; line "config2 prewarm-ms=$pw" $O/config2_pw$pw.json
  timeout 120 python bench.py --emulate-ranks 8 --no-cpu-baseline --prewarm-ms $pw > $O/share8_pw$pw.json 2>>$O/err.log; line "1/8 share prewarm-ms=$pw" $O/share8_pw$pw.json
  timeout 120 python bench.py --emulate-ranks 4 --no-cpu-baseline --prewarm-ms $pw > $O/share4_pw$pw.json 2>>$O/err.log; line "1/4 share prewarm-ms=$pw" $O/share4_pw$pw.json
  [ $pw = 0 ] && { timeout 120 python bench.py --no-cpu-baseline --no-boundary --prewarm-ms $pw > $O/headline_pw$pw.json 2>>$O/err.log; line "headline prewarm-ms=$pw" $O/headline_pw$pw.json; }
done 2>&1 | tee $O/prewarm_ab.log
timeout 120 python bench.py --workload config3 --emulate-ranks 8 --no-cpu-baseline > $O/config3_share8.json 2>>$O/err.log; line "config3 1/8 share" $O/config3_share8.json | tee -a $O/prewarm_ab.log
for w in 0 60; do echo "WARM_MS=$w"; WARM_MS=$w timeout 200 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | cut -c1-120; done | tee $O/bf16_head_dims_warm.log
timeout 600 python bench.py > $O/bench_n1.json 2>>$O/err.log; cut -c1-1400 $O/bench_n1.json
timeout 200 python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>>$O/err.log
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_gpu.log
