#!/bin/bash
# One-shot CLI, COLD: what a user of the reference sees is `Elapsed time` of ONE fresh process
# (attention.c:179-186).  Per shape: N fresh processes with the default sdpa_prepare() (buffers, code objects, ~25 ms of
# clock warm-up -- all outside the timer, like MPI_Init), then a few with SDPA_PREPARE_WARM_MS=0 (round 4's prepare), and
# the warm figure (5th of 5 calls in one process, tools/gpu_hostlevel.py) beside them.
#    bash tools/gpu_cli_cold.sh [runs]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-10}
python - <<'PY'
import numpy as np, struct
for name,(m,n,d) in {"headline":(32768,65536,128),"config2":(8192,8192,128),"config5":(32768,65536,512)}.items():
    rng=np.random.default_rng(1)
    with open("/tmp/%s.bin"%name,"wb") as f:
        f.write(struct.pack("<4i",m,n,d,d))
        for shape in ((m,d),(n,d),(n,d)):
            f.write(rng.uniform(-1,1,shape).tobytes())
        f.write(np.zeros((m,d)).tobytes())          # (no answer: the verdict is "Wrong!", the timing lines are what is read)
PY
CLI=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/bin/attention-hip
one() {   # label file env...
  local label=$1 f=$2; shift 2
  env SDPA_VERBOSE=1 "$@" $CLI /tmp/$f.bin > /tmp/cli.out 2> /tmp/cli.err
  local tot=$(grep -o "total [0-9.]* us" /tmp/cli.err | grep -o "[0-9.]*")
  local rest=$(grep "total .* us" /tmp/cli.err | sed 's/.*total [0-9.]* us | //')
  echo "$label total_us=$tot | $rest | $(grep 'last fused launch' /tmp/cli.err | sed 's/.*last fused launch //')"
}
for spec in "headline:" "config2:" "config5:SDPA_PRECISION=bf16"; do
  f=${spec%%:*}; e=${spec#*:}
  for i in $(seq 1 $N); do one "$f cold #$i (prepare warms the clock)" $f ${e:-SDPA_NOP=1}; done
  for i in 1 2 3 4; do one "$f cold #$i SDPA_PREPARE_WARM_MS=0 (round 4's prepare)" $f SDPA_PREPARE_WARM_MS=0 ${e:-SDPA_NOP=1}; done
  for i in 1 2 3; do one "$f cold #$i SDPA_STREAMED=0" $f SDPA_STREAMED=0 ${e:-SDPA_NOP=1}; done
done
