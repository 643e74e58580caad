#!/bin/bash
# round 5, call 27 (an experiment in sdpa_prepare that was NOT kept -- profiles/r05/cli_cold_prepare_experiments.log; $SDPA_PREPARE_WARM_DATA no longer exists): does the one timed call of a fresh process run at the warm rate when prepare's clock warm-up ran on pseudo-random operands
# ($SDPA_PREPARE_WARM_DATA=1) instead of zeros?  the one-shot CLI cold at the metric shape and config 5 (bf16), interleaved
O=gpurun_out/r05_27; mkdir -p $O
R=$GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, struct
for name,(m,n,d) in {"headline":(32768,65536,128),"config5":(32768,65536,512)}.items():
    rng=np.random.default_rng(1)
    with open("/tmp/%s.bin"%name,"wb") as f:
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
for i in 1 2 3 4; do
  one "headline cold warm-up on zeros #$i" headline SDPA_PREPARE_WARM_DATA=0
  one "headline cold warm-up on noise #$i" headline SDPA_PREPARE_WARM_DATA=1
done 2>&1 | tee $O/headline.log | cut -c1-200
one "headline cold warm-up on noise 300 ms" headline SDPA_PREPARE_WARM_DATA=1 SDPA_PREPARE_WARM_MS=300 | tee -a $O/headline.log | cut -c1-200
one "headline cold warm-up on noise, STREAMED=0" headline SDPA_PREPARE_WARM_DATA=1 SDPA_STREAMED=0 | tee -a $O/headline.log | cut -c1-200
for i in 1 2 3; do
  one "config5 bf16 cold warm-up on zeros #$i" config5 SDPA_PRECISION=bf16 SDPA_PREPARE_WARM_DATA=0
  one "config5 bf16 cold warm-up on noise #$i" config5 SDPA_PRECISION=bf16 SDPA_PREPARE_WARM_DATA=1
done 2>&1 | tee $O/config5.log | cut -c1-200
