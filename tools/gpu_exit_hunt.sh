#!/bin/bash
# Many short-lived processes: does anything crash at start-up or at interpreter exit?
mkdir -p gpurun_out/flaky
fails=0
for i in $(seq 1 ${RUNS:-40}); do
  python -X faulthandler - > gpurun_out/flaky/exit_$i.log 2>&1 <<'PY'
import importlib, sys, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
import torch
rng = np.random.default_rng(0)
Q, K, V = rng.standard_normal((300, 64)), rng.standard_normal((500, 64)), rng.standard_normal((500, 64))
r = pkg.attention(Q, K, V)
be = pkg.HipBackend("cuda:0")
sa = pkg.ShardedAttention(be); sa.load_kv_from_root(K, V, 500, 64, 64)
c, lm, ls = sa.batch_partial(sa.convert_q(torch.from_numpy(Q).cuda()))
torch.cuda.synchronize()
print("ok", float(np.abs(r).max()))
PY
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; head -30 gpurun_out/flaky/exit_$i.log; fi
done
echo "failures: $fails of ${RUNS:-40}"
