#!/bin/bash
# round 5, call 15 ($SDPA_KV_TAIL_CHUNK was an experiment that was NOT kept -- profiles/r05/config5_bf16_feed_analysis.log): (a) config 5 in bf16 at the boundary with the last 2048 keys as a chunk of their own ($SDPA_KV_TAIL_CHUNK) against
# the 8192-key last chunk, interleaved; (b) tools/probes/d2h_pattern on the HIP runtime the Python processes load (PyTorch's bundled
# ROCm 7.0.2 libamdhip64) -- the standalone probe (ROCm 7.2) had every device->host copy on the copy engine
O=gpurun_out/r05_15; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
  for tail in 0 2048 4096; do
    SDPA_KV_TAIL_CHUNK=$tail timeout 200 python tools/gpu_hostlevel.py config5:bf16 2>/dev/null | sed "s/^/tail=$tail /" >> $O/tail_ab.log
  done
done
python - <<'P'
import json
for l in open('gpurun_out/r05_15/tail_ab.log'):
    tag, js = l.split(' ', 1); j = json.loads(js)
    print(tag, j['shape'], 'total', j['total_ms'], 'head', j['head_ms'], 'kvstage', j['kv_stage_ms'], 'tail', j['tail_ms'], 'kernel', j['kernel_ms'], 'launches', j['fused_launches'], 'chunks', j['kv_chunks'])
P
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for pat in 0 1 3 5; do
  bytes=16777216
  (cd /tmp && LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH timeout 100 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/t_${pat} -o t -- $R/tools/probes/d2h_pattern $pat $bytes > $R/$O/p_${pat}.log 2>&1)
  k=$(grep -h copyBuffer $O/t_${pat}/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"')
  c=$(grep -h MEMORY_COPY $O/t_${pat}/*memory_copy_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"' | tr '\n' ' ')
  echo "torch-runtime pattern $pat | $(grep -h '^pattern' $O/p_${pat}.log) | shader: ${k:-none} | engine: ${c:-none}" | tee -a $O/patterns_torch_runtime.log
  rm -rf $O/t_${pat}
done
LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH ldd $R/tools/probes/d2h_pattern | grep -i "hip64\|hsa" >> $O/patterns_torch_runtime.log; tail -2 $O/patterns_torch_runtime.log
# and the same process image the tools use: which libamdhip64 does python map?
python - <<'P' | tee -a gpurun_out/r05_15/patterns_torch_runtime.log
import importlib
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd"); pkg.load()
print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l][::8])
P
