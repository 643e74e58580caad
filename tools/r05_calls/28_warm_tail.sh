#!/bin/bash
# round 5, calls 28 / 29 (experiments in sdpa_prepare that were NOT kept -- profiles/r05/cli_cold_prepare_experiments.log; the knobs below no longer exist) (29: + the staging areas touched in prepare, $SDPA_PREPARE_TOUCH): a short second clock burst BEHIND prepare's small call ($SDPA_PREPARE_WARM_TAIL_MS) -- the one-shot CLI cold at the metric shape
O=gpurun_out/r05_29; mkdir -p $O
R=$GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, struct
m,n,d=32768,65536,128
rng=np.random.default_rng(1)
with open("/tmp/headline.bin","wb") as f:
    f.write(struct.pack("<4i",m,n,d,d))
    for shape in ((m,d),(n,d),(n,d)):
        f.write(rng.uniform(-1,1,shape).tobytes())
    f.write(np.zeros((m,d)).tobytes())
PY
CLI=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/bin/attention-hip
one() {
  local label=$1 f=$2; shift 2
  env SDPA_VERBOSE=1 "$@" $CLI /tmp/$f.bin > /tmp/cli.out 2> /tmp/cli.err
  local tot=$(grep -o "total [0-9.]* us" /tmp/cli.err | grep -o "[0-9.]*")
  local rest=$(grep "total .* us" /tmp/cli.err | sed 's/.*total [0-9.]* us | //')
  echo "$label total_us=$tot | $rest"
}
for i in 1 2 3; do
  for t in 0 1; do one "headline cold staging touched=$t tail burst 8 ms #$i" headline SDPA_PREPARE_TOUCH=$t SDPA_PREPARE_WARM_TAIL_MS=8; done; one "headline cold staging touched, no tail burst #$i" headline
done 2>&1 | tee $O/headline.log | cut -c1-200
one "headline cold touched, tail 8, STREAMED=0" headline SDPA_PREPARE_WARM_TAIL_MS=8 SDPA_STREAMED=0 | tee -a $O/headline.log | cut -c1-200

