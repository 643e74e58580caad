#!/bin/bash
# round 5, call 26: the converter pool on the source arrays' NUMA node -- the one-shot CLI cold at config 5 (bf16) and the metric shape with
# $SDPA_HOST_CVT_PIN=1 (new default) / 0, interleaved; a longer clock warm-up in prepare; the placement test; the warm boundary A/B
O=gpurun_out/r05_26; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
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
one() {   # label file env...
  local label=$1 f=$2; shift 2
  env SDPA_VERBOSE=1 "$@" $CLI /tmp/$f.bin > /tmp/cli.out 2> /tmp/cli.err
  local tot=$(grep -o "total [0-9.]* us" /tmp/cli.err | grep -o "[0-9.]*")
  local rest=$(grep "total .* us" /tmp/cli.err | sed 's/.*total [0-9.]* us | //')
  echo "$label total_us=$tot | $rest | $(grep 'last fused launch' /tmp/cli.err | sed 's/.*first batch //')"
}
for i in 1 2 3 4 5; do
  one "config5 bf16 cold pin=1 #$i" config5 SDPA_PRECISION=bf16 SDPA_HOST_CVT_PIN=1
  one "config5 bf16 cold pin=0 #$i" config5 SDPA_PRECISION=bf16 SDPA_HOST_CVT_PIN=0
done 2>&1 | tee $O/cli_config5.log | cut -c1-260
for i in 1 2 3 4; do
  one "headline cold pin=1 #$i" headline SDPA_HOST_CVT_PIN=1
  one "headline cold pin=0 #$i" headline SDPA_HOST_CVT_PIN=0
done 2>&1 | tee $O/cli_headline.log | cut -c1-260
for w in 200 400; do for i in 1 2; do one "headline cold pin=1 warm_ms=$w #$i" headline SDPA_PREPARE_WARM_MS=$w; done; done 2>&1 | tee $O/cli_headline_warm.log | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -q -k "numa" 2>&1 | tail -2
for rep in 1 2 3; do for pin in 1 0; do
  SDPA_HOST_CVT_PIN=$pin SDPA_HOST_CVT_TRACE=1 timeout 200 python tools/gpu_hostlevel.py config5:bf16 headline 2> $O/err_$pin.log | sed "s/^/pin=$pin /" >> $O/warm.log
  grep "hostcvt trace" $O/err_$pin.log | tail -1 | cut -c20-360
done; done
python - <<'P'
import json
for l in open('gpurun_out/r05_26/warm.log'):
    a, js = l.split(' ', 1); j = json.loads(js)
    print(a, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'launch', j['kernel_ms'])
P
