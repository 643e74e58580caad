#!/bin/bash
# round 5, call 11 (the acq1 library = the .inc with one wave per workgroup acquiring, an experiment whose code was not kept): streamed calls back to back on DIFFERENT inputs (stale L2 / L1 lines of the previous call's images?), on the
# shipped library and on the one-wave-acquire build (-DSDPA_STREAM_ACQ_ONE_WAVE=1); then the boundary A/B of the two builds
O=gpurun_out/r05_11; mkdir -p $O
export TMPDIR=/tmp
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
T=tests/test_gpu_host_pipeline.py
SDPA_STREAM_TIMEOUT_MS=1500 timeout 600 python -m pytest $T -m gpu -q -k "back_to_back or streamed" > $O/b2b_base.log 2>&1; echo "base rc=$? $(tail -1 $O/b2b_base.log | cut -c1-120)"
SDPA_HIP_LIB=$PWD/$PKG/lib/variants/libsdpa_hip_acq1.so SDPA_STREAM_TIMEOUT_MS=1500 timeout 600 python -m pytest $T -m gpu -q -k "back_to_back or streamed" > $O/b2b_acq1.log 2>&1; echo "acq1 rc=$? $(tail -1 $O/b2b_acq1.log | cut -c1-120)"
grep -an "^FAILED\|^E  " $O/b2b_base.log $O/b2b_acq1.log | head -20 | cut -c1-300
for rep in 1 2 3; do
  for tag in base acq1; do
    lib=$PWD/$PKG/lib/variants/libsdpa_hip_$tag.so; [ $tag = base ] && lib=$PWD/$PKG/lib/libsdpa_hip.so
    for sh in headline config2; do
      SDPA_HIP_LIB=$lib timeout 200 python tools/gpu_hostlevel.py $sh 2>/dev/null | sed "s/^/$tag /" >> $O/acq_ab.log
    done
  done
done
python - <<'P'
import json
for l in open('gpurun_out/r05_11/acq_ab.log'):
    tag, js = l.split(' ', 1); j = json.loads(js)
    print(tag, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'streamed', j['streamed'])
P
